// Stress update of a LINEAR static analysis on the device: fstr_UpdateNewton's element loop
// (fistr1/src/analysis/static/fstr_Update.f90:73-264) for TYPE=361 elements with isotropic ELASTIC materials, small strain:
//   ELEMOPT361 IC    UpdateST_C3D8IC  static_LIB_3dIC.f90:220-455
//              BBAR  Update_C3D8Bbar  static_LIB_C3D8.f90:203-547 (nlgeom_flag INFINITE)
//              FI    UPDATE_C3        static_LIB_3d.f90:516-837   (nlgeom_flag INFINITE)
// strain / stress at the 8 quadrature points of every element and the internal force QFORCE from the total displacement.
//
// Work decomposition: 8 lanes per element, lane g = quadrature point g (its Jacobian, strain and stress live in that lane).
// The incompatible-mode element needs the element's 9 internal dofs alpha = -Kaa^-1 (Kad u): the reference builds the whole 33x33
// matrix [Kdd Kda; Kad Kaa] for that; here Kad u = sum_g wg Ba_g^T D (B_g u) and Kaa = sum_g wg Ba_g^T D Ba_g are summed over the 8
// lanes of the element with an xor butterfly (54 values), every lane solves the 9x9 system (Cholesky, as the assembly kernel),
// strain_g = B_g u + Ba_g alpha, and the internal force [Kdd Kda][u; alpha] = sum_g wg B_g^T sigma_g is reduced the same way; lane a
// adds node a's three entries to QFORCE (hardware fp64 atomics: up to 8 elements share a node).  Same result as the
// reference's matrix products to rounding (the parity tests hold 1e-11 of the largest entry).  Included once by fistr_hip.hip.
#pragma once

#define FXU_BS 256
#define FXU_EPB (FXU_BS / 8)

__device__ __forceinline__ double sum8(double v) {  // over the 8 lanes of an element (lanes 8k .. 8k+7 of the wave)
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  return v;
}

template <int ELEMOPT>
__global__ __launch_bounds__(FXU_BS) void k_update_c3d8_linear(int32_t n_elem, const double *__restrict__ coord,
                                                              const int32_t *__restrict__ conn, double D11, double D12, double D44,
                                                              const int32_t *__restrict__ elem_mat,
                                                              const double *__restrict__ mat_tab, const double *__restrict__ disp,
                                                              double *__restrict__ strain, double *__restrict__ stress,
                                                              double *__restrict__ qforce, int32_t *__restrict__ err) {
  const int el = threadIdx.x >> 3, g = threadIdx.x & 7;
  const int64_t e_raw = (int64_t)blockIdx.x * FXU_EPB + el;
  const bool active = e_raw < n_elem;
  const int32_t elem = active ? (int32_t)e_raw : n_elem - 1;  // idle lanes shadow the last element (uniform shuffles), write nothing
  const double GP = 0.577350269189626;
  if (elem_mat) {
    const int32_t mid = elem_mat[elem] - 1;
    D11 = mat_tab[3 * mid]; D12 = mat_tab[3 * mid + 1]; D44 = mat_tab[3 * mid + 2];
  }
  int32_t nod[8];
  double ec[8][3], ue[8][3];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    nod[j] = conn[(size_t)8 * elem + j];
#pragma unroll
    for (int d = 0; d < 3; d++) {
      ec[j][d] = coord[(size_t)3 * (nod[j] - 1) + d];
      ue[j][d] = disp[(size_t)3 * (nod[j] - 1) + d];
    }
  }
  const double xi = (g & 1) ? GP : -GP, et = (g & 2) ? GP : -GP, ze = (g & 4) ? GP : -GP;
  double det, inv[3][3], gd[11][3];
  double c0[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};  // IC: det0 * inverse Jacobian at the centre (3dIC.f90:268-270)
  double bbar[8][3];                                     // B-bar: global derivatives at the centroid (C3D8.f90:271-272)
  double vol0 = 0.0;
  if (ELEMOPT == 1 || ELEMOPT == 2) {
    hex8_global_deriv(ec, 0.0, 0.0, 0.0, det, inv, gd);
    if (ELEMOPT == 1) {
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) c0[i][j] = inv[i][j] * det;
    } else {
      double tr = 0.0;
#pragma unroll
      for (int a = 0; a < 8; a++)
#pragma unroll
        for (int d = 0; d < 3; d++) { bbar[a][d] = gd[a][d]; tr += ue[a][d] * gd[a][d]; }
      vol0 = tr / 3.0;
    }
  }
  hex8_global_deriv(ec, xi, et, ze, det, inv, gd);  // this lane's quadrature point
  const double wg = det;                             // unit weights (quadrature.f90:221)
  double gu[3][3];                                   // gdispderiv = matmul(totaldisp, gderiv)
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      double s = 0.0;
#pragma unroll
      for (int a = 0; a < 8; a++) s += ue[a][i] * gd[a][j];
      gu[i][j] = s;
    }
  double eps[6];
  {
    const double dvol = (ELEMOPT == 2) ? vol0 - (gu[0][0] + gu[1][1] + gu[2][2]) / 3.0 : 0.0;
    eps[0] = gu[0][0] + dvol; eps[1] = gu[1][1] + dvol; eps[2] = gu[2][2] + dvol;
    eps[3] = gu[0][1] + gu[1][0]; eps[4] = gu[1][2] + gu[2][1]; eps[5] = gu[2][0] + gu[0][2];
  }
  auto stress_of = [&](const double *e, double *s) {
    s[0] = D11 * e[0] + D12 * e[1] + D12 * e[2];
    s[1] = D12 * e[0] + D11 * e[1] + D12 * e[2];
    s[2] = D12 * e[0] + D12 * e[1] + D11 * e[2];
    s[3] = D44 * e[3]; s[4] = D44 * e[4]; s[5] = D44 * e[5];
  };
  if (ELEMOPT == 1) {
    // incompatible modes: derivatives of mode m at this point (3dIC.f90:296-298), B of the three modes, alpha
#pragma unroll
    for (int m = 0; m < 3; m++) {
      const double x = (m == 0) ? xi : ((m == 1) ? et : ze);
#pragma unroll
      for (int d = 0; d < 3; d++) gd[8 + m][d] = -2.0 * x * c0[m][d] / det;
    }
    double sc[6];
    stress_of(eps, sc);  // D B u: the compatible part
    double f[9], Kaa[45];
    // Ba^T s for a 6-vector s: column (m, d) of Ba has entries from node_B's pattern
    auto BaT = [&](int m, const double *s, double *out) {
      const double *q = gd[8 + m];
      out[0] = q[0] * s[0] + q[1] * s[3] + q[2] * s[5];
      out[1] = q[1] * s[1] + q[0] * s[3] + q[2] * s[4];
      out[2] = q[2] * s[2] + q[1] * s[4] + q[0] * s[5];
    };
#pragma unroll
    for (int m = 0; m < 3; m++) {
      double o[3];
      BaT(m, sc, o);
#pragma unroll
      for (int d = 0; d < 3; d++) f[3 * m + d] = o[d] * wg;
    }
    // Kaa (symmetric 9x9, lower triangle row-major: (i, j <= i) at i (i + 1) / 2 + j): column j = (m, d): D * Ba e_j, then Ba^T
#pragma unroll
    for (int mj = 0; mj < 3; mj++)
#pragma unroll
      for (int dj = 0; dj < 3; dj++) {
        const double *q = gd[8 + mj];
        double e[6] = {0, 0, 0, 0, 0, 0}, s[6];
        if (dj == 0) { e[0] = q[0]; e[3] = q[1]; e[5] = q[2]; }
        if (dj == 1) { e[1] = q[1]; e[3] = q[0]; e[4] = q[2]; }
        if (dj == 2) { e[2] = q[2]; e[4] = q[1]; e[5] = q[0]; }
        stress_of(e, s);
        const int j = 3 * mj + dj;
#pragma unroll
        for (int mi = 0; mi < 3; mi++) {
          double o[3];
          BaT(mi, s, o);
#pragma unroll
          for (int di = 0; di < 3; di++) {
            const int i = 3 * mi + di;
            if (i >= j) Kaa[i * (i + 1) / 2 + j] = o[di] * wg;
          }
        }
      }
#pragma unroll
    for (int k = 0; k < 9; k++) f[k] = sum8(f[k]);
#pragma unroll
    for (int k = 0; k < 45; k++) Kaa[k] = sum8(Kaa[k]);
    // Cholesky of Kaa in place (lower), alpha = -Kaa^-1 f
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 9; k++) {
      double d = Kaa[k * (k + 1) / 2 + k];
#pragma unroll
      for (int j = 0; j < k; j++) d -= Kaa[k * (k + 1) / 2 + j] * Kaa[k * (k + 1) / 2 + j];
      if (!(d > 1.0e-35)) { bad = true; d = 1.0; }
      const double lkk = sqrt(d), il = 1.0 / lkk;
      Kaa[k * (k + 1) / 2 + k] = il;  // keep the reciprocal on the diagonal
#pragma unroll
      for (int i = k + 1; i < 9; i++) {
        double v = Kaa[i * (i + 1) / 2 + k];
#pragma unroll
        for (int j = 0; j < k; j++) v -= Kaa[i * (i + 1) / 2 + j] * Kaa[k * (k + 1) / 2 + j];
        Kaa[i * (i + 1) / 2 + k] = v * il;
      }
    }
    if (bad && active && g == 0 && err) atomicExch(err, 1);
    double al[9];
#pragma unroll
    for (int q = 0; q < 9; q++) {  // L y = -f
      double v = -f[q];
#pragma unroll
      for (int j = 0; j < q; j++) v -= Kaa[q * (q + 1) / 2 + j] * al[j];
      al[q] = v * Kaa[q * (q + 1) / 2 + q];
    }
#pragma unroll
    for (int q = 8; q >= 0; q--) {  // L^T alpha = y
      double v = al[q];
#pragma unroll
      for (int j = q + 1; j < 9; j++) v -= Kaa[j * (j + 1) / 2 + q] * al[j];
      al[q] = v * Kaa[q * (q + 1) / 2 + q];
    }
    // strain += Ba alpha (3dIC.f90:433)
#pragma unroll
    for (int m = 0; m < 3; m++) {
      const double *q = gd[8 + m], *a3 = al + 3 * m;
      eps[0] += q[0] * a3[0]; eps[1] += q[1] * a3[1]; eps[2] += q[2] * a3[2];
      eps[3] += q[1] * a3[0] + q[0] * a3[1];
      eps[4] += q[2] * a3[1] + q[1] * a3[2];
      eps[5] += q[2] * a3[0] + q[0] * a3[2];
    }
  }
  double sg[6];
  stress_of(eps, sg);
  if (active) {
    double *se = strain + (size_t)48 * elem + 6 * g, *ss = stress + (size_t)48 * elem + 6 * g;
#pragma unroll
    for (int k = 0; k < 6; k++) { se[k] = eps[k]; ss[k] = sg[k]; }
  }
  // internal force: qf_a = sum_g wg B_a^T sigma_g  (B-bar: with the dilatational correction of C3D8.f90:458-481)
  double mine[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int a = 0; a < 8; a++) {
    double h[3] = {0.0, 0.0, 0.0};
    if (ELEMOPT == 2) {
#pragma unroll
      for (int d = 0; d < 3; d++) h[d] = (bbar[a][d] - gd[a][d]) / 3.0;
    }
    const double *q = gd[a];
    const double tr = sg[0] + sg[1] + sg[2];
    double o[3];
    o[0] = q[0] * sg[0] + q[1] * sg[3] + q[2] * sg[5] + h[0] * tr;
    o[1] = q[1] * sg[1] + q[0] * sg[3] + q[2] * sg[4] + h[1] * tr;
    o[2] = q[2] * sg[2] + q[1] * sg[4] + q[0] * sg[5] + h[2] * tr;
#pragma unroll
    for (int d = 0; d < 3; d++) {
      const double s = sum8(o[d] * wg);
      if (g == a) mine[d] = s;
    }
  }
  if (active) {
    int32_t mynode = nod[0];
#pragma unroll
    for (int a = 1; a < 8; a++)
      if (g == a) mynode = nod[a];
#pragma unroll
    for (int d = 0; d < 3; d++) unsafeAtomicAdd(qforce + (size_t)3 * (mynode - 1) + d, mine[d]);
  }
}

// Pinned host staging of the results (grown on demand, kept with the context): 2 x 48 doubles per element would otherwise cross
// PCIe through pageable memory at a fraction of the link rate.
struct UpdStage {
  double *strain = nullptr, *stress = nullptr;
  size_t cap = 0;  // doubles per array
  std::thread maker;  // pinning 2.5 GB of host memory takes 0.3 s at 3.3M elements: fx_update_c3d8_linear_prepare does it beside the solve
  bool making = false;
  int make_err = 0;
  ~UpdStage() { if (maker.joinable()) maker.join(); }  // a process that ends before using what it asked for
};
static UpdStage g_upd_stage;

static void upd_stage_wait() {
  if (g_upd_stage.making) { g_upd_stage.maker.join(); g_upd_stage.making = false; }
}
static int upd_stage_make(int device, size_t doubles) {  // (re)allocates both arrays; on the calling thread
  if (hipSetDevice(device) != hipSuccess) return 1;
  if (g_upd_stage.strain) (void)hipHostFree(g_upd_stage.strain);
  if (g_upd_stage.stress) (void)hipHostFree(g_upd_stage.stress);
  g_upd_stage.strain = g_upd_stage.stress = nullptr;
  g_upd_stage.cap = 0;
  if (hipHostMalloc((void **)&g_upd_stage.strain, doubles * 8, hipHostMallocDefault) != hipSuccess) return 1;
  if (hipHostMalloc((void **)&g_upd_stage.stress, doubles * 8, hipHostMallocDefault) != hipSuccess) return 1;
  g_upd_stage.cap = doubles;
  return 0;
}

// Optional: start pinning the host staging of fx_update_c3d8_linear for a mesh of n_elem elements on a helper thread and return at
// once (a caller that knows a stress update will follow the solve -- the fistr1 binding after fstr_StiffMatrix -- hides the 0.3 s).
extern "C" int fx_update_c3d8_linear_prepare(fx_context *c, int32_t n_elem) {
  if (!c || n_elem < 1) { g_fx_error = "fx_update_c3d8_linear_prepare: bad argument"; return FX_ERROR_RUNTIME; }
  upd_stage_wait();
  if (g_upd_stage.cap >= (size_t)48 * n_elem) return 0;
  const int device = c->device;
  const size_t doubles = (size_t)48 * n_elem;
  g_upd_stage.making = true;
  g_upd_stage.make_err = 0;
  g_upd_stage.maker = std::thread([device, doubles] { g_upd_stage.make_err = upd_stage_make(device, doubles); });
  return 0;
}

// fstr_UpdateNewton of a linear static analysis (see the header of this file).  mesh: coordinates + connectivity (host); n_mat
// materials (E, nu), elem_mat 1-based per element (NULL with one material); elemopt 1 IC, 2 B-bar, 3 FI; disp = total
// displacement unode + dunode (3 * n_node, host).  Out: *strain, *stress = pinned host arrays owned by the library, valid until
// the next call ([n_elem][8][6], the reference's gausses(1:8)%strain(1:6) / %stress(1:6)); qforce (3 * n_node, host, caller's).
extern "C" int fx_update_c3d8_linear(fx_context *c, const fx_mesh_view *mesh, int32_t n_mat, const double *E, const double *nu,
                                     const int32_t *elem_mat, int elemopt, const double *disp, const double **strain,
                                     const double **stress, double *qforce, float *ms_kernel) {
  HIP_TRY(hipSetDevice(c->device));
  if (!mesh || !E || !nu || !disp || n_mat < 1) { g_fx_error = "fx_update_c3d8_linear: null argument"; return FX_ERROR_RUNTIME; }
  if (elemopt < 1 || elemopt > 3) { g_fx_error = "fx_update_c3d8_linear: elemopt must be 1 (IC), 2 (B-bar) or 3 (FI)"; return FX_ERROR_UNSUPPORTED; }
  const int32_t ne = mesh->n_elem, nn = mesh->n_node;
  if (ne < 1 || nn < 1) { g_fx_error = "fx_update_c3d8_linear: empty mesh"; return FX_ERROR_RUNTIME; }
  for (int64_t k = 0; k < (int64_t)8 * ne; k++)
    if (mesh->conn[k] < 1 || mesh->conn[k] > nn) { g_fx_error = "fx_update_c3d8_linear: node id out of range"; return FX_ERROR_RUNTIME; }
  PhaseTimer pt("update linear");
  DevScratch tmp;
  double *d_coord = nullptr, *d_disp = nullptr, *d_strain = nullptr, *d_stress = nullptr, *d_q = nullptr, *d_mtab = nullptr;
  int32_t *d_conn = nullptr, *d_emat = nullptr, *d_err = nullptr;
  if (tmp.alloc(&d_coord, (size_t)3 * nn) || tmp.alloc(&d_disp, (size_t)3 * nn) || tmp.alloc(&d_q, (size_t)3 * nn) ||
      tmp.alloc(&d_conn, (size_t)8 * ne) || tmp.alloc(&d_strain, (size_t)48 * ne) || tmp.alloc(&d_stress, (size_t)48 * ne) ||
      tmp.alloc(&d_err, 1))
    return FX_ERROR_RUNTIME;
  HIP_TRY(hipMemcpyAsync(d_coord, mesh->coord, (size_t)3 * nn * 8, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(d_conn, mesh->conn, (size_t)8 * ne * 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(d_disp, disp, (size_t)3 * nn * 8, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemsetAsync(d_q, 0, (size_t)3 * nn * 8, c->stream));
  HIP_TRY(hipMemsetAsync(d_err, 0, 4, c->stream));
  double D11 = 0.0, D12 = 0.0, D44 = 0.0;
  std::vector<double> tab((size_t)3 * n_mat);
  for (int32_t k = 0; k < n_mat; k++) elastic_constants(E[k], nu[k], tab[3 * k], tab[3 * k + 1], tab[3 * k + 2]);
  if (n_mat > 1 || elem_mat) {
    if (!elem_mat) { g_fx_error = "fx_update_c3d8_linear: several materials need elem_mat"; return FX_ERROR_RUNTIME; }
    for (int32_t e = 0; e < ne; e++)
      if (elem_mat[e] < 1 || elem_mat[e] > n_mat) { g_fx_error = "fx_update_c3d8_linear: material id out of range"; return FX_ERROR_RUNTIME; }
    if (tmp.alloc(&d_emat, (size_t)ne) || tmp.alloc(&d_mtab, tab.size())) return FX_ERROR_RUNTIME;
    HIP_TRY(hipMemcpyAsync(d_emat, elem_mat, (size_t)ne * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_mtab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, c->stream));
  } else {
    D11 = tab[0]; D12 = tab[1]; D44 = tab[2];
  }
  if (pt.on) HIP_TRY(hipStreamSynchronize(c->stream));
  pt.lap("device buffers + uploads");
  upd_stage_wait();
  if (g_upd_stage.make_err || g_upd_stage.cap < (size_t)48 * ne) {
    g_upd_stage.make_err = 0;
    if (upd_stage_make(c->device, (size_t)48 * ne)) { g_fx_error = "fx_update_c3d8_linear: cannot pin the host staging"; (void)hipGetLastError(); return FX_ERROR_RUNTIME; }
  }
  pt.lap("pinned staging");
  const dim3 grid((unsigned)((ne + FXU_EPB - 1) / FXU_EPB)), blk(FXU_BS);
  HIP_TRY(hipEventRecord(c->ev0, c->stream));
#define FXU_LAUNCH(EO)                                                                                                       \
  hipLaunchKernelGGL((k_update_c3d8_linear<EO>), grid, blk, 0, c->stream, ne, d_coord, d_conn, D11, D12, D44, d_emat, d_mtab, \
                     d_disp, d_strain, d_stress, d_q, d_err)
  if (elemopt == 1) FXU_LAUNCH(1);
  else if (elemopt == 2) FXU_LAUNCH(2);
  else FXU_LAUNCH(3);
#undef FXU_LAUNCH
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(c->ev1, c->stream));
  int32_t herr = 0;
  HIP_TRY(hipMemcpyAsync(g_upd_stage.strain, d_strain, (size_t)48 * ne * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(g_upd_stage.stress, d_stress, (size_t)48 * ne * 8, hipMemcpyDeviceToHost, c->stream));
  if (qforce) HIP_TRY(hipMemcpyAsync(qforce, d_q, (size_t)3 * nn * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(&herr, d_err, 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  pt.lap("kernel + downloads");
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  if (ms_kernel) *ms_kernel = ms;
  if (herr) { g_fx_error = "PIVOT ERROR in the incompatible-mode block of an element (UpdateST_C3D8IC, calInverse)"; return FX_ERROR_RUNTIME; }
  if (strain) *strain = g_upd_stage.strain;
  if (stress) *stress = g_upd_stage.stress;
  return 0;
}
