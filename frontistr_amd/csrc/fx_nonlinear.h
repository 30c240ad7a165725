// Nonlinear (elastoplastic, updated/total Lagrange) C3D8 B-bar path of libfistr_hip: the steps either
// side of the linear solve inside fstr_Newton (fstr_solve_NonLinear.f90:29-167).
//
//   k_nl_stiffness   fstr_StiffMatrix.f90:58-207 -> STF_C3D8Bbar static_LIB_C3D8.f90:23-200
//                    (+ GEOMAT_C3 static_LIB_3d.f90:15-37, MatlMatrix calMatMatrix.f90:28-113,
//                    calElastoPlasticMatrix Elastoplastic.f90:16-117) + hecmw_mat_ass_elem scatter
//   k_nl_update      fstr_UpdateNewton fstr_Update.f90:25-293 -> Update_C3D8Bbar static_LIB_C3D8.f90:203-547
//                    + BackwardEuler Elastoplastic.f90:351-558 (Mises, isotropic hardening)
//   k_nl_residual    fstr_Update_NDForce fstr_Residual.f90:23-71
//   fx_nl_commit     fstr_UpdateState fstr_Update.f90:296-345
//
// Work decomposition: 8 lanes per element (8 elements per wave64).  A lane first acts as quadrature
// point LX = lane: Jacobian, global derivatives, material matrix, stress update and return mapping of
// that point -- each of these is computed once per element, not once per row as a node-parallel
// kernel would.  For the tangent the lanes then turn into the 8 node rows: the per-point data
// (24 derivatives, 21 material entries, 6 stresses, weight) is broadcast point by point with
// 8-wide shuffles (no LDS, no barrier) and lane a accumulates its 3x24 row block in registers, which
// it scatters with the same binary search + hardware fp64 atomics as the linear assembly kernel.
// The internal force is reduced over the 8 points with a shuffle butterfly and scattered by node.
//
// The reference's `integer :: flag = 0` latch in MatlMatrix (implicitly SAVEd; see oracle/fstr_nl_oracle.c)
// is state of the context: NlDev::latch is set by the first stress update of an elastoplastic material and
// from then on the tangent uses the elastic matrix, exactly as the reference does.
#pragma once
#include "fx_assemble.h"

#define FXN_BLOCK 256
#define FXN_EPB (FXN_BLOCK / 8)

// getGlobalDeriv for the 8 corner nodes (element.f90:693-744, :772-818)
__device__ __forceinline__ void hex8_gderiv(const double (&ec)[8][3], double xi, double et, double ze, double &det,
                                            double (&gd)[8][3]) {
  double dN[8][3];
  hex8_shape_deriv(xi, et, ze, dN);
  double J[3][3];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      double s = 0.0;
#pragma unroll
      for (int a = 0; a < 8; a++) s += ec[a][i] * dN[a][j];
      J[i][j] = s;
    }
  det = J[0][0] * J[1][1] * J[2][2] + J[1][0] * J[2][1] * J[0][2] + J[2][0] * J[0][1] * J[1][2] -
        J[2][0] * J[1][1] * J[0][2] - J[1][0] * J[0][1] * J[2][2] - J[0][0] * J[2][1] * J[1][2];
  const double dum = 1.0 / det;
  double inv[3][3];
  inv[0][0] = dum * (J[1][1] * J[2][2] - J[2][1] * J[1][2]);
  inv[0][1] = dum * (-J[0][1] * J[2][2] + J[2][1] * J[0][2]);
  inv[0][2] = dum * (J[0][1] * J[1][2] - J[1][1] * J[0][2]);
  inv[1][0] = dum * (-J[1][0] * J[2][2] + J[2][0] * J[1][2]);
  inv[1][1] = dum * (J[0][0] * J[2][2] - J[2][0] * J[0][2]);
  inv[1][2] = dum * (-J[0][0] * J[1][2] + J[1][0] * J[0][2]);
  inv[2][0] = dum * (J[1][0] * J[2][1] - J[2][0] * J[1][1]);
  inv[2][1] = dum * (-J[0][0] * J[2][1] + J[2][0] * J[0][1]);
  inv[2][2] = dum * (J[0][0] * J[1][1] - J[1][0] * J[0][1]);
#pragma unroll
  for (int a = 0; a < 8; a++)
#pragma unroll
    for (int j = 0; j < 3; j++) gd[a][j] = dN[a][0] * inv[0][j] + dN[a][1] * inv[1][j] + dN[a][2] * inv[2][j];
}

// 1-D table lookups (GetTableData / GetTableGrad, ttable.f90:320-335, :221-235)
__device__ __forceinline__ double nl_table_value(const NlMat &m, double a) {
  const int n = m.ntab;
  const double *t = m.tab;
  if (a < t[1]) return t[0];
  if (a >= t[2 * (n - 1) + 1]) return t[2 * (n - 1)];
  for (int i = 0; i < n - 1; i++)
    if (a >= t[2 * i + 1] && a < t[2 * i + 3]) {
      const double lambda = (a - t[2 * i + 1]) / (t[2 * i + 3] - t[2 * i + 1]);
      return (1.0 - lambda) * t[2 * i] + lambda * t[2 * i + 2];
    }
  return t[2 * (n - 1)];
}
__device__ __forceinline__ double nl_table_grad(const NlMat &m, double a) {
  const int n = m.ntab;
  const double *t = m.tab;
  if (a < t[1]) return 0.0;
  if (a >= t[2 * (n - 1) + 1]) return 0.0;
  for (int i = 0; i < n - 1; i++)
    if (a >= t[2 * i + 1] && a < t[2 * i + 3]) return (t[2 * i + 2] - t[2 * i]) / (t[2 * i + 3] - t[2 * i + 1]);
  return 0.0;
}
// calCurrYield, Elastoplastic.f90:254-292
__device__ __forceinline__ double nl_curr_yield(const NlMat &m, double p) {
  switch (m.harden) {
    case 0: return m.pl[0] + m.pl[1] * p;
    case 1: return nl_table_value(m, p);
    case 2: return m.pl[1] * pow(m.pl[0] + p, m.pl[2]);
    case 3: return (p <= m.pl[0]) ? m.pl[1] : m.pl[1] * pow(p / m.pl[0], 1.0 / m.pl[2]);
  }
  return -1.0;
}
// calHardenCoeff, Elastoplastic.f90:175-220
__device__ __forceinline__ double nl_harden_coeff(const NlMat &m, double p) {
  switch (m.harden) {
    case 0: return m.pl[1];
    case 1: return nl_table_grad(m, p);
    case 2: return m.pl[1] * m.pl[2] * pow(m.pl[0] + p, m.pl[2] - 1.0);
    case 3: {
      const double ef = nl_curr_yield(m, p);
      return m.pl[1] * pow(ef / m.pl[1], 1.0 - m.pl[2]) / (m.pl[0] * m.pl[2]);
    }
  }
  return -1.0;
}

// BackwardEuler, Mises branch (Elastoplastic.f90:351-459, :557)
__device__ __forceinline__ void nl_backward_euler(const NlMat &m, double (&s)[6], double plstrain, int32_t &istat, double &fstat1) {
  const double tol = 1.0e-3;
  const double J1 = (s[0] + s[1] + s[2]) / 3.0;
  double dv[6] = {s[0] - J1, s[1] - J1, s[2] - J1, s[3], s[4], s[5]};
  const double J2 = 0.5 * (dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2]) + (dv[3] * dv[3] + dv[4] * dv[4] + dv[5] * dv[5]);
  const double yd = sqrt(3.0 * J2);
  double f = yd - nl_curr_yield(m, plstrain);
  if (fabs(f) < tol) { istat = 1; return; }
  if (f < 0.0) { istat = 0; return; }
  istat = 1;
  const double G = m.E / (2.0 * (1.0 + m.nu));
  double dlambda = 0.0;
  for (int i = 0; i < 5; i++) {
    const double H = nl_harden_coeff(m, plstrain + dlambda);
    dlambda = dlambda + f / (3.0 * G + H);
    if (dlambda < 0.0) { dlambda = 0.0; istat = 0; break; }
    f = yd - 3.0 * G * dlambda - nl_curr_yield(m, plstrain + dlambda);
    if (fabs(f) < tol * tol) break;
  }
  const double fac = 1.0 - 3.0 * dlambda * G / yd;
  s[0] = fac * dv[0] + J1; s[1] = fac * dv[1] + J1; s[2] = fac * dv[2] + J1;
  s[3] = fac * dv[3]; s[4] = fac * dv[4]; s[5] = fac * dv[5];
  fstat1 = plstrain + dlambda;
}

// symmetric 6x6 in 21 entries, row-major upper triangle: index of (i,j), i<=j
__device__ __forceinline__ constexpr int sym21(int i, int j) { return (i <= j) ? (i * (13 - i)) / 2 + (j - i) : (j * (13 - j)) / 2 + (i - j); }

// material matrix of one quadrature point, as STF_C3D8Bbar uses it (:90-101): MatlMatrix with the latch,
// minus GEOMAT_C3 for the updated-Lagrange flag.
__device__ __forceinline__ void nl_point_matrix(const NlMat &m, int latch, int flag, const double (&s)[6], int istat, double fstat1,
                                                double (&Dm)[21]) {
  const double D11 = m.E * (1.0 - m.nu) / (1.0 - 2.0 * m.nu) / (1.0 + m.nu);
  const double D12 = m.E * m.nu / (1.0 - 2.0 * m.nu) / (1.0 + m.nu);
  const double D44 = m.E / (1.0 + m.nu) * 0.5;
#pragma unroll
  for (int k = 0; k < 21; k++) Dm[k] = 0.0;
  Dm[sym21(0, 0)] = D11; Dm[sym21(1, 1)] = D11; Dm[sym21(2, 2)] = D11;
  Dm[sym21(0, 1)] = D12; Dm[sym21(0, 2)] = D12; Dm[sym21(1, 2)] = D12;
  Dm[sym21(3, 3)] = D44; Dm[sym21(4, 4)] = D44; Dm[sym21(5, 5)] = D44;
  if (m.plastic && !latch && istat != 0) {  // calElastoPlasticMatrix, Mises (:49-115)
    const double J1 = s[0] + s[1] + s[2];
    const double dv[6] = {s[0] - J1 / 3.0, s[1] - J1 / 3.0, s[2] - J1 / 3.0, s[3], s[4], s[5]};
    const double J2 = 0.5 * (dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2]) + (dv[3] * dv[3] + dv[4] * dv[4] + dv[5] * dv[5]);
    const double q = 2.0 * sqrt(J2), r3 = sqrt(3.0);
    const double a[6] = {r3 * (dv[0] / q), r3 * (dv[1] / q), r3 * (dv[2] / q), r3 * (2.0 * dv[3] / q), r3 * (2.0 * dv[4] / q), r3 * (2.0 * dv[5] / q)};
    const double da[6] = {D11 * a[0] + D12 * a[1] + D12 * a[2], D12 * a[0] + D11 * a[1] + D12 * a[2], D12 * a[0] + D12 * a[1] + D11 * a[2],
                          D44 * a[3], D44 * a[4], D44 * a[5]};
    double dum = 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++) dum += da[i] * a[i];
    dum = nl_harden_coeff(m, fstat1) + dum;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
      for (int j = i; j < 6; j++) Dm[sym21(i, j)] -= da[i] * da[j] / dum;
  }
  if (flag == 2) {  // GEOMAT_C3
    Dm[sym21(0, 0)] -= 2.0 * s[0]; Dm[sym21(0, 3)] -= s[3]; Dm[sym21(0, 5)] -= s[5];
    Dm[sym21(1, 1)] -= 2.0 * s[1]; Dm[sym21(1, 3)] -= s[3]; Dm[sym21(1, 4)] -= s[4];
    Dm[sym21(2, 2)] -= 2.0 * s[2]; Dm[sym21(2, 4)] -= s[4]; Dm[sym21(2, 5)] -= s[5];
    Dm[sym21(3, 3)] -= 0.5 * (s[0] + s[1]); Dm[sym21(3, 4)] -= 0.5 * s[5]; Dm[sym21(3, 5)] -= 0.5 * s[4];
    Dm[sym21(4, 4)] -= 0.5 * (s[2] + s[1]); Dm[sym21(4, 5)] -= 0.5 * s[3];
    Dm[sym21(5, 5)] -= 0.5 * (s[0] + s[2]);
  }
}

// strain-displacement block of one node incl. the B-bar correction and, for the total-Lagrange flag, BL1
// (static_LIB_C3D8.f90:103-158): g = global derivatives of the node, h = (Bbar - g)/3, F = gdispderiv.
template <int NLGEOM>
__device__ __forceinline__ void nl_node_B(const double *g, const double *h, const double (&F)[9], double (&B)[6][3]) {
  node_B(g, h, B);
  if (NLGEOM == 1) {
#pragma unroll
    for (int c = 0; c < 3; c++) {  // F[3*c+d] = gdispderiv(c+1, d+1)
      B[0][c] += F[3 * c + 0] * g[0];
      B[1][c] += F[3 * c + 1] * g[1];
      B[2][c] += F[3 * c + 2] * g[2];
      B[3][c] += F[3 * c + 1] * g[0] + F[3 * c + 0] * g[1];
      B[4][c] += F[3 * c + 1] * g[2] + F[3 * c + 2] * g[1];
      B[5][c] += F[3 * c + 2] * g[0] + F[3 * c + 0] * g[2];
    }
  }
}

__device__ __forceinline__ double bcast8(double v, int src) { return __shfl(v, src, 8); }

template <int NLGEOM>
__global__ __launch_bounds__(FXN_BLOCK) void k_nl_stiffness(int32_t n_elem, const double *__restrict__ coord,
                                                            const int32_t *__restrict__ conn, const double *__restrict__ unode,
                                                            const double *__restrict__ dunode, NlMat m, int latch,
                                                            const double *__restrict__ stress, const double *__restrict__ fstat,
                                                            const int32_t *__restrict__ istat, const int32_t *__restrict__ indexL,
                                                            const int32_t *__restrict__ itemL, const int32_t *__restrict__ indexU,
                                                            const int32_t *__restrict__ itemU, double *__restrict__ D,
                                                            double *__restrict__ AL, double *__restrict__ AU,
                                                            double *__restrict__ Kout, int32_t *__restrict__ err,
                                                            const int32_t *__restrict__ elem_list, int32_t e0,
                                                            const int32_t *__restrict__ pos_map, int atomic,
                                                            const NlMat *__restrict__ mats, const int32_t *__restrict__ emat) {
  // positions [e0, n_elem) of elem_list: elements of one NLGEOM group; atomic == 0: they are of one colour (no shared node),
  // scattered without atomics, see k_assemble_c3d8.  mats / emat: several sections, element e uses mats[emat[e] - 1].
  const int lane8 = threadIdx.x & 7;
  int32_t epos = e0 + blockIdx.x * FXN_EPB + (threadIdx.x >> 3);
  const bool active = epos < n_elem;
  if (!active) epos = n_elem - 1;  // keep the 8-lane group converged for the shuffles; results discarded
  const int32_t elem = elem_list ? elem_list[epos] : epos;
  if (mats) m = mats[emat[elem] - 1];
  int32_t nod[8];
  double gd[8][3], bbar[8][3], Dm[21], S[6], F[9], wg;
  {
    double ec[8][3], ut[8][3];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      nod[j] = conn[(size_t)8 * elem + j];
#pragma unroll
      for (int d = 0; d < 3; d++) {
        const size_t k = (size_t)3 * (nod[j] - 1) + d;
        ut[j][d] = unode[k] + dunode[k];
        ec[j][d] = coord[k];
        if (NLGEOM == 2) ec[j][d] += ut[j][d];
      }
    }
    double det;
    hex8_gderiv(ec, 0.0, 0.0, 0.0, det, bbar);  // dilatation at the centroid (:72-73)
    const double GP = 0.577350269189626;
    const double xi = (lane8 & 1) ? GP : -GP, et = (lane8 & 2) ? GP : -GP, ze = (lane8 & 4) ? GP : -GP;
    hex8_gderiv(ec, xi, et, ze, det, gd);
    wg = det;
#pragma unroll
    for (int i = 0; i < 6; i++) S[i] = stress[((size_t)8 * elem + lane8) * 6 + i];
    nl_point_matrix(m, latch, NLGEOM, S, m.plastic ? istat[(size_t)8 * elem + lane8] : 0, m.plastic ? fstat[(size_t)8 * elem + lane8] : 0.0, Dm);
#pragma unroll
    for (int k = 0; k < 9; k++) F[k] = 0.0;
    if (NLGEOM == 1) {  // gdispderiv = u . gderiv (:131)
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
        for (int d = 0; d < 3; d++) {
          double s = 0.0;
#pragma unroll
          for (int a = 0; a < 8; a++) s += ut[a][c] * gd[a][d];
          F[3 * c + d] = s;
        }
    }
  }
  // ---- lanes become node rows
  const int a = lane8;
  double K[8][9];
#pragma unroll
  for (int b = 0; b < 8; b++)
#pragma unroll
    for (int e = 0; e < 9; e++) K[b][e] = 0.0;
  for (int LX = 0; LX < 8; LX++) {
    double g[8][3], Dl[21], Sl[6], Fl[9];
#pragma unroll
    for (int b = 0; b < 8; b++)
#pragma unroll
      for (int d = 0; d < 3; d++) g[b][d] = bcast8(gd[b][d], LX);
#pragma unroll
    for (int k = 0; k < 21; k++) Dl[k] = bcast8(Dm[k], LX);
    if (NLGEOM != 0) {
#pragma unroll
      for (int k = 0; k < 6; k++) Sl[k] = bcast8(S[k], LX);
    }
    if (NLGEOM == 1) {
#pragma unroll
      for (int k = 0; k < 9; k++) Fl[k] = bcast8(F[k], LX);
    } else {
#pragma unroll
      for (int k = 0; k < 9; k++) Fl[k] = 0.0;
    }
    const double w = bcast8(wg, LX);
    double ga[3] = {0.0, 0.0, 0.0}, ha[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int b = 0; b < 8; b++)
      if (b == a) {
#pragma unroll
        for (int d = 0; d < 3; d++) { ga[d] = g[b][d]; ha[d] = (bbar[b][d] - g[b][d]) / 3.0; }
      }
    double Ba[6][3];
    nl_node_B<NLGEOM>(ga, ha, Fl, Ba);
    double sa[3] = {0.0, 0.0, 0.0};  // S . grad N_a  (initial stress matrix :164-195)
    if (NLGEOM != 0) {
      sa[0] = Sl[0] * ga[0] + Sl[3] * ga[1] + Sl[5] * ga[2];
      sa[1] = Sl[3] * ga[0] + Sl[1] * ga[1] + Sl[4] * ga[2];
      sa[2] = Sl[5] * ga[0] + Sl[4] * ga[1] + Sl[2] * ga[2];
    }
#pragma unroll
    for (int b = 0; b < 8; b++) {
      double hb[3] = {(bbar[b][0] - g[b][0]) / 3.0, (bbar[b][1] - g[b][1]) / 3.0, (bbar[b][2] - g[b][2]) / 3.0};
      double Bb[6][3], DB[6][3];
      nl_node_B<NLGEOM>(g[b], hb, Fl, Bb);
#pragma unroll
      for (int r = 0; r < 6; r++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
          double s = 0.0;
#pragma unroll
          for (int q = 0; q < 6; q++) s += Dl[sym21(r, q)] * Bb[q][j];
          DB[r][j] = s;
        }
      double geo = 0.0;
      if (NLGEOM != 0) geo = (sa[0] * g[b][0] + sa[1] * g[b][1] + sa[2] * g[b][2]) * w;
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
          double s = 0.0;
#pragma unroll
          for (int q = 0; q < 6; q++) s += Ba[q][i] * DB[q][j];
          K[b][3 * i + j] += s * w;
        }
      if (NLGEOM != 0) { K[b][0] += geo; K[b][4] += geo; K[b][8] += geo; }
    }
  }
  if (!active) return;
  if (Kout) {
#pragma unroll
    for (int b = 0; b < 8; b++)
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Kout[(size_t)elem * 576 + (size_t)(3 * a + i) * 24 + 3 * b + j] = K[b][3 * i + j];
    return;
  }
  int32_t inod = 0;
#pragma unroll
  for (int b = 0; b < 8; b++)
    if (b == a) inod = nod[b];
#pragma unroll
  for (int b = 0; b < 8; b++) {  // hecmw_mat_add_node, hecmw_mat_ass.f90:72-134
    const int32_t jnod = nod[b];
    double *dst;
    if (inod == jnod) dst = D + (size_t)9 * (inod - 1);
    else if (jnod < inod) {
      const int32_t k = pos_map ? pos_map[(size_t)64 * elem + 8 * a + b] : item_search(itemL, indexL[inod - 1], indexL[inod], jnod);
      if (k < 0) { if (err) atomicExch(err, 2); continue; }
      dst = AL + (size_t)9 * k;
    } else {
      const int32_t k = pos_map ? pos_map[(size_t)64 * elem + 8 * a + b] : item_search(itemU, indexU[inod - 1], indexU[inod], jnod);
      if (k < 0) { if (err) atomicExch(err, 2); continue; }
      dst = AU + (size_t)9 * k;
    }
    if (!atomic) {
#pragma unroll
      for (int e = 0; e < 9; e++) dst[e] += K[b][e];
    } else {
#pragma unroll
      for (int e = 0; e < 9; e++) unsafeAtomicAdd(dst + e, K[b][e]);
    }
  }
}

// Update_C3D8Bbar + scatter of the internal force.  qf_out (tests): per-element qf[24] instead of the scatter.
template <int NLGEOM>
__global__ __launch_bounds__(FXN_BLOCK) void k_nl_update(int32_t n_elem, const double *__restrict__ coord,
                                                         const int32_t *__restrict__ conn, const double *__restrict__ unode,
                                                         const double *__restrict__ dunode, NlMat m, double *__restrict__ stress,
                                                         double *__restrict__ strain, const double *__restrict__ stress_bak,
                                                         const double *__restrict__ strain_bak, const double *__restrict__ plstrain,
                                                         double *__restrict__ fstat, int32_t *__restrict__ istat,
                                                         double *__restrict__ qforce, double *__restrict__ qf_out,
                                                         const int32_t *__restrict__ elem_list, int32_t e0,
                                                         const NlMat *__restrict__ mats, const int32_t *__restrict__ emat) {
  // positions [e0, n_elem) of elem_list: the elements of this NLGEOM group; mats / emat: several sections
  const int LX = threadIdx.x & 7;
  int32_t epos = e0 + blockIdx.x * FXN_EPB + (threadIdx.x >> 3);
  const bool active = epos < n_elem;
  if (!active) epos = n_elem - 1;
  const int32_t elem = elem_list ? elem_list[epos] : epos;
  if (mats) m = mats[emat[elem] - 1];
  int32_t nod[8];
  double ec[8][3], td[8][3], e1[8][3];  // integration configuration, displacement driving the strain, end configuration
#pragma unroll
  for (int j = 0; j < 8; j++) {
    nod[j] = conn[(size_t)8 * elem + j];
#pragma unroll
    for (int d = 0; d < 3; d++) {
      const size_t k = (size_t)3 * (nod[j] - 1) + d;
      const double u = unode[k], du = dunode[k], x = coord[k];
      if (NLGEOM == 2) {  // :255-260
        ec[j][d] = (0.5 * du + u) + x;
        e1[j][d] = (du + u) + x;
        td[j][d] = du;
      } else {
        ec[j][d] = x;
        e1[j][d] = x;
        td[j][d] = u + du;
      }
    }
  }
  double det, gd[8][3], bbar[8][3];
  hex8_gderiv(ec, 0.0, 0.0, 0.0, det, bbar);
  double vol0 = 0.0;
  {
    double dd[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      double s = 0.0;
#pragma unroll
      for (int a = 0; a < 8; a++) s += td[a][i] * bbar[a][i];
      dd[i] = s;
    }
    vol0 = (dd[0] + dd[1] + dd[2]) / 3.0;
  }
  const double GP = 0.577350269189626;
  const double xi = (LX & 1) ? GP : -GP, et = (LX & 2) ? GP : -GP, ze = (LX & 4) ? GP : -GP;
  hex8_gderiv(ec, xi, et, ze, det, gd);
  double g[3][3];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      double s = 0.0;
#pragma unroll
      for (int a = 0; a < 8; a++) s += td[a][i] * gd[a][j];
      g[i][j] = s;
    }
  const double dvol = vol0 - (g[0][0] + g[1][1] + g[2][2]) / 3.0;
  const double D11 = m.E * (1.0 - m.nu) / (1.0 - 2.0 * m.nu) / (1.0 + m.nu);
  const double D12 = m.E * m.nu / (1.0 - 2.0 * m.nu) / (1.0 + m.nu);
  const double D44 = m.E / (1.0 + m.nu) * 0.5;
  double de[6] = {g[0][0] + dvol, g[1][1] + dvol, g[2][2] + dvol, g[0][1] + g[1][0], g[1][2] + g[2][1], g[2][0] + g[0][2]};
  if (NLGEOM == 1) {  // Green-Lagrange strain :378-388
#pragma unroll
    for (int c = 0; c < 3; c++) de[c] += 0.5 * (g[0][c] * g[0][c] + g[1][c] * g[1][c] + g[2][c] * g[2][c]);
    de[3] += g[0][0] * g[0][1] + g[1][0] * g[1][1] + g[2][0] * g[2][1];
    de[4] += g[0][1] * g[0][2] + g[1][1] * g[1][2] + g[2][1] * g[2][2];
    de[5] += g[0][0] * g[0][2] + g[1][0] * g[1][2] + g[2][0] * g[2][2];
  }
  const double ds[6] = {D11 * de[0] + D12 * de[1] + D12 * de[2], D12 * de[0] + D11 * de[1] + D12 * de[2],
                        D12 * de[0] + D12 * de[1] + D11 * de[2], D44 * de[3], D44 * de[4], D44 * de[5]};
  const size_t gp = (size_t)8 * elem + LX;
  double sg[6], eg[6];
  if (NLGEOM == 2) {  // :407-432
    double sb[6], eb[6];
#pragma unroll
    for (int i = 0; i < 6; i++) { sb[i] = stress_bak[gp * 6 + i]; eb[i] = strain_bak[gp * 6 + i]; }
    const double r01 = 0.5 * (g[0][1] - g[1][0]), r12 = 0.5 * (g[1][2] - g[2][1]), r02 = 0.5 * (g[0][2] - g[2][0]);
    const double rot[3][3] = {{0.0, r01, r02}, {-r01, 0.0, r12}, {-r02, -r12, 0.0}};
    const double Sb[3][3] = {{sb[0], sb[3], sb[5]}, {sb[3], sb[1], sb[4]}, {sb[5], sb[4], sb[2]}};
    double dum[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) {
        double p = 0.0, q = 0.0;
#pragma unroll
        for (int k = 0; k < 3; k++) { p += rot[i][k] * Sb[k][j]; q += Sb[i][k] * rot[k][j]; }
        dum[i][j] = p - q;
      }
    const double t3 = 3.0 * vol0;
    sg[0] = sb[0] + ds[0] + dum[0][0] - sb[0] * t3;
    sg[1] = sb[1] + ds[1] + dum[1][1] - sb[1] * t3;
    sg[2] = sb[2] + ds[2] + dum[2][2] - sb[2] * t3;
    sg[3] = sb[3] + ds[3] + dum[0][1] - sb[3] * t3;
    sg[4] = sb[4] + ds[4] + dum[1][2] - sb[4] * t3;
    sg[5] = sb[5] + ds[5] + dum[2][0] - sb[5] * t3;
#pragma unroll
    for (int i = 0; i < 6; i++) eg[i] = eb[i] + de[i];
  } else {
#pragma unroll
    for (int i = 0; i < 6; i++) { sg[i] = ds[i]; eg[i] = de[i]; }
  }
  if (m.plastic) {
    int32_t ist = istat[gp];
    double fs = fstat[gp];
    nl_backward_euler(m, sg, plstrain[gp], ist, fs);
    if (active) { istat[gp] = ist; fstat[gp] = fs; }
  }
  if (active) {
#pragma unroll
    for (int i = 0; i < 6; i++) { stress[gp * 6 + i] = sg[i]; strain[gp * 6 + i] = eg[i]; }
  }
  // ---- internal force of this point (:451-545)
  double F[9];
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int d = 0; d < 3; d++) F[3 * c + d] = g[c][d];
  if (NLGEOM == 2) {
    hex8_gderiv(e1, 0.0, 0.0, 0.0, det, bbar);  // Bbar2 at the end configuration
    hex8_gderiv(e1, xi, et, ze, det, gd);
  }
  const double wg = det;
  double qf[24];
#pragma unroll
  for (int b = 0; b < 8; b++) {
    const double hb[3] = {(bbar[b][0] - gd[b][0]) / 3.0, (bbar[b][1] - gd[b][1]) / 3.0, (bbar[b][2] - gd[b][2]) / 3.0};
    double Bb[6][3];
    nl_node_B<NLGEOM>(gd[b], hb, F, Bb);
#pragma unroll
    for (int j = 0; j < 3; j++) {
      double s = 0.0;
#pragma unroll
      for (int q = 0; q < 6; q++) s += sg[q] * Bb[q][j];
      qf[3 * b + j] = s * wg;
    }
  }
#pragma unroll
  for (int k = 0; k < 24; k++) {  // sum over the 8 quadrature points
    double v = qf[k];
    v += __shfl_xor(v, 1, 8);
    v += __shfl_xor(v, 2, 8);
    v += __shfl_xor(v, 4, 8);
    qf[k] = v;
  }
  if (!active) return;
  double mine[3] = {0.0, 0.0, 0.0};
  int32_t inod = 0;
#pragma unroll
  for (int b = 0; b < 8; b++)
    if (b == LX) { mine[0] = qf[3 * b]; mine[1] = qf[3 * b + 1]; mine[2] = qf[3 * b + 2]; inod = nod[b]; }
  if (qf_out) {
#pragma unroll
    for (int i = 0; i < 3; i++) qf_out[(size_t)elem * 24 + 3 * LX + i] = mine[i];
    return;
  }
#pragma unroll
  for (int i = 0; i < 3; i++) unsafeAtomicAdd(qforce + (size_t)3 * (inod - 1) + i, mine[i]);
}

// fstr_Update_NDForce: B = GL - QFORCE, prescribed dofs cleared (fstr_Residual.f90:45-47, :100-133)
__global__ void k_nl_residual(int64_t n3, const double *__restrict__ GL, const double *__restrict__ Q,
                              const uint8_t *__restrict__ flag, double *__restrict__ B) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n3; i += (int64_t)gridDim.x * blockDim.x)
    B[i] = (flag && flag[i]) ? 0.0 : GL[i] - Q[i];
}

// fstr_UpdateState: plstrain = fstatus(1) (updateEPState), strain_bak/stress_bak = strain/stress
__global__ void k_nl_commit(int64_t npt, int plastic, const double *__restrict__ fstat, double *__restrict__ plstrain,
                            const double *__restrict__ stress, const double *__restrict__ strain, double *__restrict__ stress_bak,
                            double *__restrict__ strain_bak, const NlMat *__restrict__ mats, const int32_t *__restrict__ emat) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < 6 * npt; i += (int64_t)gridDim.x * blockDim.x) {
    stress_bak[i] = stress[i];
    strain_bak[i] = strain[i];
    if (i < npt) {  // isElastoplastic(pMaterial%mtype) of the point's element (fstr_Update.f90:323-326)
      const int pl = mats ? mats[emat[i >> 3] - 1].plastic : plastic;
      if (pl) plstrain[i] = fstat[i];
    }
  }
}
