// Block sizes other than 3x3: hecmw_solve / hecmw_matvec for NDOF = 1, 2, 4, 5, 6 on the device.
//
//   hecmw_matvec_nn_inner      hecmw1/src/solver/las/hecmw_solver_las_nn.f90:135-310  (+ the unrolled las_11/22/44/66 copies)
//   hecmw_precond_DIAG_nn_*    hecmw1/src/solver/precond/nn/hecmw_precond_DIAG_nn.f90:27-137   (+ precond/11/22/44/66)
//   hecmw_precond_SSOR_nn_*    hecmw1/src/solver/precond/nn/hecmw_precond_SSOR_nn.f90:55-420   (RCM + multicolour ordering)
//   hecmw_precond_nn_apply     hecmw1/src/solver/precond/nn/hecmw_precond_nn.f90 (additive Schwarz loop over iterPREmax)
//   hecmw_solve_CG / BiCGSTAB  hecmw1/src/solver/iterative/hecmw_solver_CG.f90:19-312, hecmw_solver_BiCGSTAB.f90:16-297
//   hecmw_solve_GMRES / GPBiCG hecmw_solver_GMRES.f90:17-458, hecmw_solver_GPBiCG.f90:17-505 (fx_krylov2_host.h through OpsNN)
//   hecmw_solve_iterative      hecmw1/src/solver/iterative/hecmw_solver_Iterative.f90:13-210 (checks, flags, final residual)
//
// Layout: the same sliced block-ELL idea as the 3x3 path -- one thread per block row, 64 rows per slice, slice width =
// longest row of the slice -- with the NDOF*NDOF entries of a block stored entry-major across the 64 lanes
// (val[((slice_base + k) * NDOF^2 + e) * 64 + lane]), so every load of the inner loop is one coalesced 512-byte line
// Inside that line-per-value frame the values are paired into 16-byte words per lane (nn_pos; NDOF = 1 pairs entries: nn1_pos).
// Vectors stay in the caller's numbering (NDOF * NP doubles); the SSOR sweeps address rows through a slot -> row map in
// colour order.  CG and BiCGSTAB keep their scalars on the device (the k_scalar<OP> state machine of the 3x3 path); GMRES and
// GPBiCG are host-driven (fx_krylov2_host.h).  Measured in DESIGN.md §7.  No CPU arithmetic on vectors or matrices.
#pragma once

struct NnBell {
  int32_t nslots = 0, nslices = 0;
  int64_t nblocks_padded = 0;
  int64_t *slice_ptr = nullptr;  // device, nslices + 1, in block-columns
  int32_t *slot_row = nullptr;   // device, nslots: row id (0-based) or -1 (padding)
  int32_t *col = nullptr;        // device, nblocks_padded * 64
  double *val = nullptr;         // device, nblocks_padded * nd2 * 64
};

struct NnDev {
  int ndof = 0;
  int32_t N = 0, NP = 0, NPL = 0, NPU = 0, nn_internal = 0;
  bool have_matrix = false, precond_valid = false;
  int precond_kind = 0;  // 1 SSOR, 3 DIAG
  double sigma = 1.0;
  NnBell M, L, U;
  std::vector<int32_t> color_slice;  // first slice of each colour (+ end)
  int ncolor = 0;
  double *D = nullptr, *alu = nullptr, *B = nullptr, *X = nullptr;
  double *W[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  double *partials = nullptr, *scal = nullptr;
  double *extra = nullptr;  // GMRES basis / GPBiCG work vectors (count * NDOF * NP)
  double *scale = nullptr;  // SCALING=YES: 1 / sqrt(|d_ii|), NDOF * NP
  int extra_n = 0, iterpremax = 1;
  // halo tables in the caller's numbering (0-based)
  int32_t n_neighbor = 0, n_export = 0, n_import = 0;
  std::vector<int32_t> neighbor, export_index, import_index;
  int32_t *export_item = nullptr, *import_item = nullptr;
  double *sendbuf = nullptr, *recvbuf = nullptr, *h_send = nullptr, *h_recv = nullptr;
  // host copies of the profile for the SSOR set-up
  std::vector<int32_t> h_indexL, h_itemL, h_indexU, h_itemU;
  const double *cur_AL = nullptr, *cur_AU = nullptr;  // the caller's off-diagonal blocks, valid during the current fx_solve only
};

template <typename F>
static void nn_each(int64_t n, F f) {  // f(i) for i in [0, n) on the host threads
  parallel_for(n, [&](int64_t a, int64_t b) { for (int64_t i = a; i < b; i++) f(i); });
}

static NnDev *nn_of(fx_context *c) {
  if (!c->nn) c->nn = new NnDev();
  return (NnDev *)c->nn;
}

static void nn_bell_free(NnBell &b) {
  dev_free(b.slice_ptr); dev_free(b.slot_row); dev_free(b.col); dev_free(b.val);
  b = NnBell();
}

static void nn_free(fx_context *c) {
  if (!c->nn) return;
  NnDev *n = (NnDev *)c->nn;
  nn_bell_free(n->M); nn_bell_free(n->L); nn_bell_free(n->U);
  dev_free(n->D); dev_free(n->alu); dev_free(n->B); dev_free(n->X);
  for (auto &w : n->W) dev_free(w);
  dev_free(n->partials); dev_free(n->scal); dev_free(n->extra); dev_free(n->scale);
  dev_free(n->export_item); dev_free(n->import_item); dev_free(n->sendbuf); dev_free(n->recvbuf);
  if (n->h_send) (void)hipHostFree(n->h_send);
  if (n->h_recv) (void)hipHostFree(n->h_recv);
  delete n;
  c->nn = nullptr;
}

// ---------------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------------
// LU without pivoting with reciprocal pivots (DIAG_nn.f90:80-91 == SSOR_nn.f90:186-197), one thread per block row
template <int ND>
__global__ void k_nn_lu(int32_t N, const double *__restrict__ D, double sigma, double *__restrict__ alu) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  double a[ND * ND];
#pragma unroll
  for (int e = 0; e < ND * ND; e++) a[e] = D[(size_t)ND * ND * i + e];
#pragma unroll
  for (int d = 0; d < ND; d++) a[d * ND + d] = a[d * ND + d] * sigma;
#pragma unroll
  for (int k = 0; k < ND; k++) {
    a[k * ND + k] = 1.0 / a[k * ND + k];
#pragma unroll
    for (int r = k + 1; r < ND; r++) {
      a[r * ND + k] = a[r * ND + k] * a[k * ND + k];
#pragma unroll
      for (int j = k + 1; j < ND; j++) a[r * ND + j] = a[r * ND + j] - a[r * ND + k] * a[k * ND + j];
    }
  }
#pragma unroll
  for (int e = 0; e < ND * ND; e++) alu[(size_t)ND * ND * i + e] = a[e];
}

// forward / back substitution (DIAG_nn.f90:111-121 == SSOR_nn.f90:321-331).  QUIRK66 reproduces the reference's hand-unrolled
// NDOF = 6 SSOR, which reads entry (6,1) in place of entry (4,3) (precond/66/hecmw_precond_SSOR_66.f90:420, :504).
template <int ND, bool QUIRK66>
__device__ __forceinline__ void nn_lusolve(const double *__restrict__ a, double *X) {
#pragma unroll
  for (int j = 1; j < ND; j++)
#pragma unroll
    for (int k = 0; k < j; k++) X[j] = X[j] - ((QUIRK66 && ND == 6 && j == 3 && k == 2) ? a[ND * 5] : a[ND * j + k]) * X[k];
#pragma unroll
  for (int j = ND - 1; j >= 0; j--) {
#pragma unroll
    for (int k = ND - 1; k > j; k--) X[j] = X[j] - a[ND * j + k] * X[k];
    X[j] = a[(ND + 1) * j] * X[j];
  }
}

template <int ND>
__global__ void k_nn_diag_apply(int32_t N, const double *__restrict__ alu, double *__restrict__ z) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  double X[ND];
#pragma unroll
  for (int d = 0; d < ND; d++) X[d] = z[(size_t)ND * i + d];
  nn_lusolve<ND, false>(alu + (size_t)ND * ND * i, X);
#pragma unroll
  for (int d = 0; d < ND; d++) z[(size_t)ND * i + d] = X[d];
}

// Scalar systems (NDOF = 1) store the entries of a slice PAIR-PACKED -- entries k, k+1 of lane l in one 16-byte word, their
// two column ids in one 8-byte word, an odd last entry alone -- exactly as the 3x3 path packs its blocks: an 8-byte load per
// lane runs at 0.54-0.70x the rate of a 16-byte one on MI355X (MI355X_MICROARCH.md), and at one value per entry the value
// stream of the entry-major layout was nothing but 8-byte loads.  Wider blocks keep the entry-major layout.
__host__ __device__ __forceinline__ size_t nn1_pos(int64_t base, int w, int k, int lane) {  // position of entry k of `lane` (values and column ids alike)
  const int wp = w & ~1;
  return k < wp ? (size_t)(base + (k & ~1)) * 64 + (size_t)lane * 2 + (k & 1) : (size_t)(base + k) * 64 + lane;
}
// Wider blocks pair the values INSIDE an entry: values q, q+1 of lane l share a 16-byte word, an odd last value (NDOF = 5) alone.
__host__ __device__ __forceinline__ size_t nn_pos(int nd2, int64_t entry, int q, int lane) {
  const int qp = nd2 & ~1;
  return q < qp ? ((size_t)entry * nd2 + (q & ~1)) * 64 + (size_t)lane * 2 + (q & 1) : ((size_t)entry * nd2 + q) * 64 + lane;
}

// One thread per block row of a slice.  MODE 0: y = A x; 1: y = b - A x; 2: forward SSOR sweep  z_i <- LU^-1 (z_i - sum L z);
// 3: backward sweep  z_i <- z_i - LU^-1 (sum U z).  Slices [s0, s1) of one launch are mutually independent.
template <int ND, int MODE>
__global__ __launch_bounds__(256) void k_nn_rows(int32_t s0, int32_t s1, const int64_t *__restrict__ slice_ptr,
                                                 const int32_t *__restrict__ slot_row, const int32_t *__restrict__ col,
                                                 const double *__restrict__ val, const double *x, const double *__restrict__ b,
                                                 double *y, const double *__restrict__ alu, const int32_t *__restrict__ gate,
                                                 int32_t gate_val) {
  if (gate && *gate != gate_val) return;  // device-resident Krylov state: the kernel only runs while the state asks for it
  const int lane = threadIdx.x & 63;
  const int32_t s = s0 + blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= s1) return;
  const int32_t row = slot_row[(size_t)s * 64 + lane];
  const int64_t base = slice_ptr[s];
  const int w = (int)(slice_ptr[s + 1] - base);
  double acc[ND];
#pragma unroll
  for (int d = 0; d < ND; d++) acc[d] = 0.0;
  if (ND == 1) {  // pair-packed scalar entries: one 16-byte value word + one 8-byte id word per two entries, streamed past the caches
    const int np = w >> 1;
    const fx_d2 *v2 = (const fx_d2 *)(val + (size_t)base * 64) + lane;
    const fx_i2 *c2 = (const fx_i2 *)(col + (size_t)base * 64) + lane;
    double a0 = 0.0;
    int j = 0;
    for (; j + 1 < np; j += 2) {  // two pairs in flight
      const fx_i2 ca = __builtin_nontemporal_load(c2 + (size_t)j * 64), cb = __builtin_nontemporal_load(c2 + (size_t)(j + 1) * 64);
      const fx_d2 va = __builtin_nontemporal_load(v2 + (size_t)j * 64), vb = __builtin_nontemporal_load(v2 + (size_t)(j + 1) * 64);
      const double xa = x[ca.x], xb = x[ca.y], xc = x[cb.x], xd = x[cb.y];
      a0 = a0 + va.x * xa; a0 = a0 + va.y * xb;
      a0 = a0 + vb.x * xc; a0 = a0 + vb.y * xd;
    }
    if (j < np) {
      const fx_i2 ca = __builtin_nontemporal_load(c2 + (size_t)j * 64);
      const fx_d2 va = __builtin_nontemporal_load(v2 + (size_t)j * 64);
      a0 = a0 + va.x * x[ca.x]; a0 = a0 + va.y * x[ca.y];
    }
    if (w & 1) {
      const size_t o = (size_t)(base + w - 1) * 64 + lane;
      a0 = a0 + val[o] * x[col[o]];
    }
    acc[0] = a0;
  } else
  for (int k = 0; k < w; k++) {
    const int32_t cidx = col[(base + k) * 64 + lane];
    const double *v = val + (size_t)(base + k) * (ND * ND) * 64;
    double a[ND * ND], xv[ND];
#pragma unroll
    for (int j = 0; j < (ND * ND) / 2; j++) {  // value pairs as 16-byte words, streamed past the caches
      const fx_d2 wd = __builtin_nontemporal_load((const fx_d2 *)(v + (size_t)2 * j * 64) + lane);
      a[2 * j] = wd.x;
      a[2 * j + 1] = wd.y;
    }
    if ((ND * ND) & 1) a[ND * ND - 1] = __builtin_nontemporal_load(v + (size_t)(ND * ND - 1) * 64 + lane);
#pragma unroll
    for (int e = 0; e < ND; e++) xv[e] = x[(size_t)ND * cidx + e];
#pragma unroll
    for (int d = 0; d < ND; d++)
#pragma unroll
      for (int e = 0; e < ND; e++) acc[d] = acc[d] + a[d * ND + e] * xv[e];
  }
  if (row < 0) return;
  if (MODE == 0) {
#pragma unroll
    for (int d = 0; d < ND; d++) y[(size_t)ND * row + d] = acc[d];
  } else if (MODE == 1) {
#pragma unroll
    for (int d = 0; d < ND; d++) y[(size_t)ND * row + d] = b[(size_t)ND * row + d] - acc[d];
  } else if (MODE == 2) {
    double X[ND];
#pragma unroll
    for (int d = 0; d < ND; d++) X[d] = y[(size_t)ND * row + d] - acc[d];
    nn_lusolve<ND, true>(alu + (size_t)ND * ND * row, X);
#pragma unroll
    for (int d = 0; d < ND; d++) y[(size_t)ND * row + d] = X[d];
  } else {
    nn_lusolve<ND, true>(alu + (size_t)ND * ND * row, acc);
#pragma unroll
    for (int d = 0; d < ND; d++) y[(size_t)ND * row + d] = y[(size_t)ND * row + d] - acc[d];
  }
}

__global__ void k_nn_dot(int64_t n, const double *__restrict__ x, const double *__restrict__ y, double *__restrict__ partials) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += x[i] * y[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = sh[0];
}
__global__ void k_nn_reduce(int np, const double *__restrict__ partials, double *__restrict__ out) {  // fixed order: deterministic
  __shared__ double sh[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < np; i += 256) s += partials[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sh[0];
}
// out = a*x + b*y + c*z (z may be null)
__global__ void k_nn_lin(int64_t n, double *out, double a, const double *x, double b, const double *y, double cc, const double *z) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double v = a * x[i] + b * y[i];
    if (z) v += cc * z[i];
    out[i] = v;
  }
}
// p = r + beta * (p - omega * v)   (hecmw_solver_BiCGSTAB.f90:160-165)
__global__ void k_nn_bicg_p(int64_t n, double beta, double omega, const double *__restrict__ r, const double *__restrict__ v,
                            double *__restrict__ p) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    p[i] = r[i] + beta * (p[i] - omega * v[i]);
}
__global__ void k_nn_halo_pack(int32_t n, int nd, const int32_t *__restrict__ item, const double *__restrict__ x,
                               double *__restrict__ buf) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n * nd) return;
  buf[i] = x[(size_t)nd * item[i / nd] + i % nd];
}
__global__ void k_nn_halo_unpack(int32_t n, int nd, const int32_t *__restrict__ item, const double *__restrict__ buf,
                                 double *__restrict__ x) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n * nd) return;
  x[(size_t)nd * item[i / nd] + i % nd] = buf[i];
}
__global__ void k_nn_check_zero_diag(int32_t N, int nd, const double *__restrict__ D, int32_t *flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)N * nd) return;
  if (fabs(D[(size_t)nd * nd * (i / nd) + (size_t)(nd + 1) * (i % nd)]) == 0.0) *flag = 1;
}

// SCALING=YES (hecmw_solver_scaling_fw_nn / _bk_nn, las/hecmw_solver_scaling_nn.f90:20-100, :102-180): symmetric diagonal scaling
__global__ void k_nn_scale_vec(int32_t N, int nd, const double *__restrict__ D, double *__restrict__ scale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)N * nd) return;
  scale[i] = 1.0 / sqrt(fabs(D[(size_t)nd * nd * (i / nd) + (size_t)(nd + 1) * (i % nd)]));
}
__global__ void k_nn_scale_diag(int32_t NP, int nd, double *__restrict__ D, const double *__restrict__ scale, int back) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)NP * nd * nd) return;
  const int64_t i = t / (nd * nd);
  const int r = (int)(t % (nd * nd)) / nd, q = (int)(t % nd);
  const double si = scale[(size_t)nd * i + r], sj = scale[(size_t)nd * i + q];
  D[t] = back ? D[t] / (si * sj) : (D[t] * si) * sj;   // evaluation order of the reference (:62-70, :157-165)
}
template <int ND>
__global__ void k_nn_scale_bell(int32_t nslices, const int64_t *__restrict__ slice_ptr, const int32_t *__restrict__ slot_row,
                                const int32_t *__restrict__ col, double *__restrict__ val, const double *__restrict__ scale,
                                int back) {
  const int lane = threadIdx.x & 63;
  const int32_t s = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= nslices) return;
  const int32_t row = slot_row[(size_t)s * 64 + lane];
  if (row < 0) return;
  const int64_t base = slice_ptr[s];
  const int w = (int)(slice_ptr[s + 1] - base);
  for (int k = 0; k < w; k++) {
    if (ND == 1) {  // pair-packed layout
      const size_t o = nn1_pos(base, w, k, lane);
      const double si = scale[row], sj = scale[col[o]];
      val[o] = back ? val[o] / (si * sj) : (val[o] * si) * sj;
      continue;
    }
    const int32_t cidx = col[(base + k) * 64 + lane];
#pragma unroll
    for (int d = 0; d < ND; d++)
#pragma unroll
      for (int e = 0; e < ND; e++) {
        const double si = scale[(size_t)ND * row + d], sj = scale[(size_t)ND * cidx + e];
        double &x = val[nn_pos(ND * ND, base + k, d * ND + e, lane)];
        x = back ? x / (si * sj) : (x * si) * sj;
      }
  }
}
__global__ void k_nn_scale_rhs(int64_t n, double *__restrict__ B, double *__restrict__ X, const double *__restrict__ scale, int back) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (back) { X[i] = X[i] * scale[i]; B[i] = B[i] / scale[i]; }
    else B[i] = B[i] * scale[i];
  }
}

#define NN_DISPATCH(nd, ...)                                                                                    \
  switch (nd) {                                                                                                 \
    case 1: { constexpr int ND = 1; __VA_ARGS__; } break;                                                       \
    case 2: { constexpr int ND = 2; __VA_ARGS__; } break;                                                       \
    case 4: { constexpr int ND = 4; __VA_ARGS__; } break;                                                       \
    case 5: { constexpr int ND = 5; __VA_ARGS__; } break;                                                       \
    case 6: { constexpr int ND = 6; __VA_ARGS__; } break;                                                       \
    default: g_fx_error = "NDOF must be 1, 2, 3, 4, 5 or 6"; return FX_ERROR_UNSUPPORTED;                         \
  }

// ---------------------------------------------------------------------------------------------------------------
// host: layout
// ---------------------------------------------------------------------------------------------------------------
// rows[slot] = row id or -1; for each row a list of (column, pointer to its nd2 values).  Slices of 64 slots.
struct NnRowEntry { int32_t col; const double *src; };
// fill(slot, out): the (column, source block) list of that slot's row, in the order the kernel is to visit it.  Called
// twice per slot (widths, then values) from the host threads with a per-call scratch vector -- no per-row allocation.
template <class Fill>
static int nn_bell_build(fx_context *c, NnBell &b, int nd, const std::vector<int32_t> &rows, Fill fill) {
  nn_bell_free(b);
  const int nd2 = nd * nd;
  b.nslots = (int32_t)rows.size();
  b.nslices = b.nslots / 64;
  std::vector<int64_t> sp((size_t)b.nslices + 1, 0);
  parallel_for(b.nslices, [&](int64_t s0, int64_t s1) {
    std::vector<NnRowEntry> tmp;
    for (int64_t s = s0; s < s1; s++) {
      size_t w = 0;
      for (int l = 0; l < 64; l++) {
        tmp.clear();
        if (rows[(size_t)s * 64 + l] >= 0) fill((int64_t)s * 64 + l, tmp);
        w = std::max(w, tmp.size());
      }
      sp[s + 1] = (int64_t)w;
    }
  });
  for (int32_t s = 0; s < b.nslices; s++) sp[s + 1] += sp[s];
  b.nblocks_padded = sp[b.nslices];
  std::vector<int32_t> col((size_t)std::max<int64_t>(b.nblocks_padded, 1) * 64, 0);
  std::vector<double> val((size_t)std::max<int64_t>(b.nblocks_padded, 1) * nd2 * 64, 0.0);
  parallel_for(b.nslices, [&](int64_t s0, int64_t s1) {
    std::vector<NnRowEntry> tmp;
    for (int64_t s = s0; s < s1; s++)
      for (int l = 0; l < 64; l++) {
        tmp.clear();
        if (rows[(size_t)s * 64 + l] >= 0) fill((int64_t)s * 64 + l, tmp);
        const int w = (int)(sp[s + 1] - sp[s]);
        for (size_t k = 0; k < tmp.size(); k++) {
          if (nd == 1) {  // pair-packed scalar entries (nn1_pos)
            const size_t o = nn1_pos(sp[s], w, (int)k, l);
            col[o] = tmp[k].col;
            val[o] = tmp[k].src[0];
            continue;
          }
          col[(size_t)(sp[s] + k) * 64 + l] = tmp[k].col;
          for (int q = 0; q < nd2; q++) val[nn_pos(nd2, sp[s] + (int64_t)k, q, l)] = tmp[k].src[q];
        }
      }
  });
  if (dev_alloc(&b.slice_ptr, sp.size()) || dev_alloc(&b.slot_row, std::max<size_t>(rows.size(), 1)) ||
      dev_alloc(&b.col, col.size()) || dev_alloc(&b.val, val.size()))
    return FX_ERROR_RUNTIME;
  HIP_TRY(hipMemcpy(b.slice_ptr, sp.data(), sp.size() * 8, hipMemcpyHostToDevice));
  if (!rows.empty()) HIP_TRY(hipMemcpy(b.slot_row, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(b.col, col.data(), col.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(b.val, val.data(), val.size() * 8, hipMemcpyHostToDevice));
  return 0;
}

static int nn_upload(fx_context *c, const fx_matrix_view *m, const fx_comm_view *cm, bool values_changed) {
  NnDev *n = nn_of(c);
  const int nd = m->NDOF, nd2 = nd * nd;
  const bool shape = !n->have_matrix || n->ndof != nd || n->N != m->N || n->NP != m->NP || n->NPL != m->NPL || n->NPU != m->NPU;
  if (shape) {
    nn_free(c);
    n = nn_of(c);
    n->ndof = nd; n->N = m->N; n->NP = m->NP; n->NPL = m->NPL; n->NPU = m->NPU;
    n->nn_internal = (cm && cm->nn_internal > 0) ? cm->nn_internal : m->N;
    const size_t len = (size_t)nd * std::max(m->NP, 1);
    if (dev_alloc(&n->D, (size_t)nd2 * std::max(m->NP, 1)) || dev_alloc(&n->alu, (size_t)nd2 * std::max(m->NP, 1)) ||
        dev_alloc(&n->B, len) || dev_alloc(&n->X, len) || dev_alloc(&n->partials, 512) || dev_alloc(&n->scal, 8))
      return FX_ERROR_RUNTIME;
    for (auto &w : n->W) {
      if (dev_alloc(&w, len)) return FX_ERROR_RUNTIME;
      HIP_TRY(hipMemset(w, 0, len * 8));
    }
    if (cm && cm->n_neighbor_pe > 0) {
      n->n_neighbor = cm->n_neighbor_pe;
      n->neighbor.assign(cm->neighbor_pe, cm->neighbor_pe + n->n_neighbor);
      n->import_index.assign(cm->import_index, cm->import_index + n->n_neighbor + 1);
      n->export_index.assign(cm->export_index, cm->export_index + n->n_neighbor + 1);
      n->n_import = n->import_index.back();
      n->n_export = n->export_index.back();
      std::vector<int32_t> ex(cm->export_item, cm->export_item + n->n_export), im(cm->import_item, cm->import_item + n->n_import);
      for (auto &v : ex) { v -= 1; if (v < 0 || v >= m->NP) { g_fx_error = "export_item out of range"; return FX_ERROR_RUNTIME; } }
      for (auto &v : im) { v -= 1; if (v < 0 || v >= m->NP) { g_fx_error = "import_item out of range"; return FX_ERROR_RUNTIME; } }
      if (dev_alloc(&n->export_item, std::max<size_t>(ex.size(), 1)) || dev_alloc(&n->import_item, std::max<size_t>(im.size(), 1)) ||
          dev_alloc(&n->sendbuf, (size_t)nd * std::max(n->n_export, 1)) || dev_alloc(&n->recvbuf, (size_t)nd * std::max(n->n_import, 1)))
        return FX_ERROR_RUNTIME;
      if (!ex.empty()) HIP_TRY(hipMemcpy(n->export_item, ex.data(), ex.size() * 4, hipMemcpyHostToDevice));
      if (!im.empty()) HIP_TRY(hipMemcpy(n->import_item, im.data(), im.size() * 4, hipMemcpyHostToDevice));
    }
  }
  if (shape || values_changed) {
    // every column must be a valid row id: the kernels gather x[col] unconditionally
    for (int32_t j = 0; j < m->NPL; j++)
      if (m->itemL[j] < 1 || m->itemL[j] > m->NP) { g_fx_error = "itemL out of range"; return FX_ERROR_RUNTIME; }
    for (int32_t j = 0; j < m->NPU; j++)
      if (m->itemU[j] < 1 || m->itemU[j] > m->NP) { g_fx_error = "itemU out of range"; return FX_ERROR_RUNTIME; }
    n->h_indexL.assign(m->indexL, m->indexL + m->NP + 1);
    n->h_indexU.assign(m->indexU, m->indexU + m->NP + 1);
    n->h_itemL.assign(m->itemL, m->itemL + m->NPL);
    n->h_itemU.assign(m->itemU, m->itemU + m->NPU);
    HIP_TRY(hipMemcpy(n->D, m->D, (size_t)nd2 * m->NP * 8, hipMemcpyHostToDevice));
    // SpMV rows 1..N in the caller's order: D, then the lower blocks, then the upper blocks (las_nn.f90:274-307)
    const int32_t nslots = (m->N + 63) / 64 * 64;
    std::vector<int32_t> rows((size_t)nslots, -1);
    for (int32_t i = 0; i < m->N; i++) rows[i] = i;
    auto fillM = [&](int64_t i, std::vector<NnRowEntry> &e) {
      e.push_back({(int32_t)i, m->D + (size_t)nd2 * i});
      for (int32_t j = m->indexL[i]; j < m->indexL[i + 1]; j++) e.push_back({m->itemL[j] - 1, m->AL + (size_t)nd2 * j});
      for (int32_t j = m->indexU[i]; j < m->indexU[i + 1]; j++) e.push_back({m->itemU[j] - 1, m->AU + (size_t)nd2 * j});
    };
    if (nn_bell_build(c, n->M, nd, rows, fillM)) return FX_ERROR_RUNTIME;
    n->have_matrix = true;
    if (shape) n->precond_valid = false;  // new values alone: the flags / recycle policy of the solve decide (as the 3x3 path)
  }
  if (m->B) HIP_TRY(hipMemcpy(n->B, m->B, (size_t)nd * m->NP * 8, hipMemcpyHostToDevice));
  if (m->X) HIP_TRY(hipMemcpy(n->X, m->X, (size_t)nd * m->NP * 8, hipMemcpyHostToDevice));
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// host: building blocks
// ---------------------------------------------------------------------------------------------------------------
static int nn_halo(fx_context *c, double *x) {  // hecmw_update_m_R
  NnDev *n = nn_of(c);
  if (n->n_neighbor <= 0 || (c->nranks <= 1 && !c->nccl && !c->cb_halo)) return 0;
  if (!c->nccl && !c->cb_halo) { g_fx_error = "halo exchange requested but no communicator (fx_comm_init) was set"; return FX_ERROR_RUNTIME; }
  const int nd = n->ndof;
  if (n->n_export > 0)
    hipLaunchKernelGGL(k_nn_halo_pack, dim3(((int64_t)n->n_export * nd + 255) / 256), dim3(256), 0, c->stream, n->n_export, nd,
                       n->export_item, x, n->sendbuf);
  if (!c->nccl) {
    if (!n->h_send) {
      HIP_TRY(hipHostMalloc((void **)&n->h_send, (size_t)nd * std::max(n->n_export, 1) * 8, hipHostMallocDefault));
      HIP_TRY(hipHostMalloc((void **)&n->h_recv, (size_t)nd * std::max(n->n_import, 1) * 8, hipHostMallocDefault));
    }
    HIP_TRY(hipMemcpyAsync(n->h_send, n->sendbuf, (size_t)nd * n->n_export * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->cb_halo(n->h_send, n->h_recv, c->cb_user);
    HIP_TRY(hipMemcpyAsync(n->recvbuf, n->h_recv, (size_t)nd * n->n_import * 8, hipMemcpyHostToDevice, c->stream));
  } else {
    NCCL_TRY(g_rccl.GroupStart());
    for (int k = 0; k < n->n_neighbor; k++) {
      const int32_t ns = n->export_index[k + 1] - n->export_index[k], nr = n->import_index[k + 1] - n->import_index[k];
      if (ns > 0)
        NCCL_TRY(g_rccl.Send(n->sendbuf + (size_t)nd * n->export_index[k], (size_t)nd * ns, ncclDouble, n->neighbor[k],
                             (ncclComm_t)c->nccl, c->stream));
      if (nr > 0)
        NCCL_TRY(g_rccl.Recv(n->recvbuf + (size_t)nd * n->import_index[k], (size_t)nd * nr, ncclDouble, n->neighbor[k],
                             (ncclComm_t)c->nccl, c->stream));
    }
    NCCL_TRY(g_rccl.GroupEnd());
  }
  if (n->n_import > 0)
    hipLaunchKernelGGL(k_nn_halo_unpack, dim3(((int64_t)n->n_import * nd + 255) / 256), dim3(256), 0, c->stream, n->n_import, nd,
                       n->import_item, n->recvbuf, x);
  HIP_TRY(hipGetLastError());
  return 0;
}

template <int ND, int MODE>
static void nn_rows_launch(fx_context *c, const NnBell &b, int32_t s0, int32_t s1, const double *x, const double *rhs, double *y,
                           const double *alu, const int32_t *gate = nullptr, int32_t gate_val = 0) {
  if (s1 <= s0) return;
  hipLaunchKernelGGL((k_nn_rows<ND, MODE>), dim3((s1 - s0 + 3) / 4), dim3(256), 0, c->stream, s0, s1, b.slice_ptr, b.slot_row, b.col,
                     b.val, x, rhs, y, alu, gate, gate_val);
}

// y = A x (mode 0) or y = b - A x (mode 1); x gets its halo first (las_nn.f90:247)
static int nn_spmv(fx_context *c, int mode, double *x, const double *b, double *y, const int32_t *gate = nullptr, int32_t gate_val = 0) {
  NnDev *n = nn_of(c);
  if (nn_halo(c, x)) return FX_ERROR_RUNTIME;  // collective: every rank exchanges whether or not its state lets the product run
  if (mode == 0) { NN_DISPATCH(n->ndof, nn_rows_launch<ND, 0>(c, n->M, 0, n->M.nslices, x, nullptr, y, nullptr, gate, gate_val)) }
  else { NN_DISPATCH(n->ndof, nn_rows_launch<ND, 1>(c, n->M, 0, n->M.nslices, x, b, y, nullptr, gate, gate_val)) }
  HIP_TRY(hipGetLastError());
  return 0;
}

static int nn_dot(fx_context *c, const double *x, const double *y, double *out) {  // hecmw_InnerProduct_R over NDOF * nn_internal
  NnDev *n = nn_of(c);
  const int64_t len = (int64_t)n->ndof * n->nn_internal;
  const int np = (int)std::max<int64_t>(1, std::min<int64_t>(512, (len + 2047) / 2048));
  hipLaunchKernelGGL(k_nn_dot, dim3(np), dim3(256), 0, c->stream, len, x, y, n->partials);
  hipLaunchKernelGGL(k_nn_reduce, dim3(1), dim3(256), 0, c->stream, np, n->partials, n->scal);
  HIP_TRY(hipGetLastError());
  if (multi_rank(c) && allreduce_dev(c, n->scal, 1)) return FX_ERROR_RUNTIME;
  double *h = (double *)(c->st_host + 3);  // pinned staging word
  HIP_TRY(hipMemcpyAsync(h, n->scal, 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  *out = *h;
  return 0;
}

static inline int nn_vgrid(int64_t len) { return (int)std::max<int64_t>(1, std::min<int64_t>(2048, (len + 255) / 256)); }
#define NN_LIN(out, a, x, b, y, cc, z) \
  hipLaunchKernelGGL(k_nn_lin, dim3(nn_vgrid(nlen)), dim3(256), 0, c->stream, nlen, out, a, x, b, y, cc, z)

// SSOR set-up: the reference's RCM + multicolour ordering (fx_order.cpp, same as the 3x3 path), lower / upper parts in the
// new numbering with halo columns dropped (hecmw_matrix_reorder.f90:50), rows of a colour padded to whole slices.
static int nn_ssor_setup(fx_context *c, int ncolor_in) {
  NnDev *n = nn_of(c);
  const int nd = n->ndof, nd2 = nd * nd;
  const int32_t N = n->N;
  std::vector<int32_t> perm, cidx;
  {  // the ordering of the 3x3 path (ssor_ordering: device walks from 100 k block rows on); the profile goes up for the walk only
    DevScratch tmp;
    DevCSR A;
    A.N = N; A.NP = n->NP; A.NPL = (int32_t)n->h_itemL.size(); A.NPU = (int32_t)n->h_itemU.size();
    if (N >= std::min(c->bfs_device_min, c->mc_device_min)) {
      if (tmp.alloc(&A.indexL, n->h_indexL.size()) || tmp.alloc(&A.indexU, n->h_indexU.size()) || tmp.alloc(&A.itemL, n->h_itemL.size()) ||
          tmp.alloc(&A.itemU, n->h_itemU.size()))
        return FX_ERROR_RUNTIME;
      HIP_TRY(hipMemcpyAsync(A.indexL, n->h_indexL.data(), n->h_indexL.size() * 4, hipMemcpyHostToDevice, c->stream));
      HIP_TRY(hipMemcpyAsync(A.indexU, n->h_indexU.data(), n->h_indexU.size() * 4, hipMemcpyHostToDevice, c->stream));
      HIP_TRY(hipMemcpyAsync(A.itemL, n->h_itemL.data(), n->h_itemL.size() * 4, hipMemcpyHostToDevice, c->stream));
      HIP_TRY(hipMemcpyAsync(A.itemU, n->h_itemU.data(), n->h_itemU.size() * 4, hipMemcpyHostToDevice, c->stream));
    }
    std::vector<int32_t> deg;
    PhaseTimer pt("nn ssor symbolic");
    if (ssor_ordering(c, A, n->h_indexL.data(), n->h_itemL.data(), n->h_indexU.data(), n->h_itemU.data(), std::max(ncolor_in, 1), deg, perm,
                      cidx, pt))
      return FX_ERROR_RUNTIME;
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  std::vector<int32_t> iperm((size_t)N);
  for (int32_t i = 0; i < N; i++) iperm[perm[i]] = i;
  n->ncolor = (int)cidx.size() - 1;
  std::vector<int32_t> rows;
  n->color_slice.assign(1, 0);
  std::vector<int32_t> slot_new;  // slot -> new index or -1
  for (int k = 0; k < n->ncolor; k++) {
    // rows of one colour are mutually independent: visiting them in ascending row id changes no result and keeps the gathers
    // of neighbouring lanes close in memory (the reference's order inside a colour is the RCM visiting order)
    std::vector<std::pair<int32_t, int32_t>> byrow;
    for (int32_t q = cidx[k]; q < cidx[k + 1]; q++) byrow.push_back({perm[q], q});
    std::sort(byrow.begin(), byrow.end());
    for (auto &pr : byrow) { rows.push_back(pr.first); slot_new.push_back(pr.second); }
    while (rows.size() % 64) { rows.push_back(-1); slot_new.push_back(-1); }
    n->color_slice.push_back((int32_t)(rows.size() / 64));
  }
  auto fillLU = [&](int64_t s, std::vector<NnRowEntry> &out, bool lower) {
    const int32_t iold = rows[s], inew = slot_new[s];
    std::pair<int32_t, NnRowEntry> buf[128];  // keyed by the NEW index of the column
    std::vector<std::pair<int32_t, NnRowEntry>> big;
    size_t cnt = 0;
    auto add = [&](int32_t kold, const double *src) {
      if (kold >= N) return;  // halo column: localized preconditioner
      const int32_t knew = iperm[kold];
      if ((knew < inew) != lower) return;
      if (cnt < 128) buf[cnt++] = {knew, {kold, src}};
      else { if (big.empty()) big.assign(buf, buf + cnt); big.push_back({knew, {kold, src}}); cnt++; }
    };
    for (int32_t j = n->h_indexL[iold]; j < n->h_indexL[iold + 1]; j++) add(n->h_itemL[j] - 1, n->cur_AL + (size_t)nd2 * j);
    for (int32_t j = n->h_indexU[iold]; j < n->h_indexU[iold + 1]; j++) add(n->h_itemU[j] - 1, n->cur_AU + (size_t)nd2 * j);
    std::pair<int32_t, NnRowEntry> *p = big.empty() ? buf : big.data();
    if (lower) std::sort(p, p + cnt, [](const auto &a, const auto &b) { return a.first < b.first; });  // forward: ascending (:300)
    else std::sort(p, p + cnt, [](const auto &a, const auto &b) { return a.first > b.first; });       // backward: descending (:352)
    for (size_t k = 0; k < cnt; k++) out.push_back(p[k].second);
  };
  if (nn_bell_build(c, n->L, nd, rows, [&](int64_t s, std::vector<NnRowEntry> &o) { fillLU(s, o, true); }) ||
      nn_bell_build(c, n->U, nd, rows, [&](int64_t s, std::vector<NnRowEntry> &o) { fillLU(s, o, false); }))
    return FX_ERROR_RUNTIME;
  return 0;
}

static int nn_precond_setup(fx_context *c, int precond, double sigma, int ncolor_in) {
  NnDev *n = nn_of(c);
  if (precond == 1 || precond == 2) {
    if (nn_ssor_setup(c, ncolor_in)) return FX_ERROR_RUNTIME;
    n->precond_kind = 1;
  } else if (precond == 3) {
    n->precond_kind = 3;
  } else {
    g_fx_error = "NDOF != 3: PRECOND must be 1/2 (SSOR) or 3 (DIAG) on the GPU path";
    return FX_ERROR_INCONS_PC;
  }
  n->sigma = sigma;
  HIP_TRY(hipMemsetAsync(n->alu, 0, (size_t)n->ndof * n->ndof * std::max(n->NP, 1) * 8, c->stream));
  if (n->N > 0) { NN_DISPATCH(n->ndof, hipLaunchKernelGGL((k_nn_lu<ND>), dim3((n->N + 127) / 128), dim3(128), 0, c->stream, n->N, n->D, sigma, n->alu)) }
  HIP_TRY(hipGetLastError());
  n->precond_valid = true;
  return 0;
}

// hecmw_precond_nn_apply: ZP = R (internal rows, halo rows 0), Z = 0, iterPREmax x { ZP <- M^-1 ZP; Z += ZP; ZP = R - A Z }
static int nn_precond_apply(fx_context *c, int iterpremax, const double *r, double *z) {
  NnDev *n = nn_of(c);
  const int nd = n->ndof;
  const int64_t nlen = (int64_t)nd * n->N;
  double *zp = n->W[8];
  if (iterpremax <= 0) {
    HIP_TRY(hipMemcpyAsync(z, r, (size_t)nlen * 8, hipMemcpyDeviceToDevice, c->stream));
    return 0;
  }
  HIP_TRY(hipMemsetAsync(zp, 0, (size_t)nd * n->NP * 8, c->stream));
  HIP_TRY(hipMemcpyAsync(zp, r, (size_t)nlen * 8, hipMemcpyDeviceToDevice, c->stream));
  HIP_TRY(hipMemsetAsync(z, 0, (size_t)nd * n->NP * 8, c->stream));
  for (int it = 1; it <= iterpremax; it++) {
    if (n->precond_kind == 3) {
      if (n->N > 0) { NN_DISPATCH(nd, hipLaunchKernelGGL((k_nn_diag_apply<ND>), dim3((n->N + 127) / 128), dim3(128), 0, c->stream, n->N, n->alu, zp)) }
    } else {
      for (int k = 0; k < n->ncolor; k++) {
        NN_DISPATCH(nd, nn_rows_launch<ND, 2>(c, n->L, n->color_slice[k], n->color_slice[k + 1], zp, nullptr, zp, n->alu))
      }
      for (int k = n->ncolor - 1; k >= 0; k--) {
        NN_DISPATCH(nd, nn_rows_launch<ND, 3>(c, n->U, n->color_slice[k], n->color_slice[k + 1], zp, nullptr, zp, n->alu))
      }
    }
    NN_LIN(z, 1.0, z, 1.0, zp, 0.0, (const double *)nullptr);
    if (it == iterpremax) break;
    if (nn_spmv(c, 1, z, r, zp)) return FX_ERROR_RUNTIME;
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Krylov loops: CG and BiCGSTAB with the scalars RESIDENT ON THE DEVICE, exactly as the 3x3 path runs them (KrylovState,
// k_scalar<OP>: fixed-order reduction of the partial sums + the reference's scalar logic in one single-block kernel; every
// update kernel early-outs unless the state is RUNNING).  The host enqueues whole iterations and polls one word every few
// iterations -- no host round trip per dot product (the first version synchronised 3-6 times per iteration, and so does
// the reference: hecmw_InnerProduct_R, hecmw_solver_misc.f90:46-70).  Multi-rank: reduce -> in-stream all-reduce -> logic.
// ---------------------------------------------------------------------------------------------------------------
struct NnResult { int iter = 0, error = 0; double resid = 0.0; std::vector<double> hist; };

__global__ __launch_bounds__(FX_BLOCK) void k_nn_cg_p(int64_t n, const KrylovState *__restrict__ st, const double *__restrict__ z,
                                                      double *__restrict__ p) {  // hecmw_solver_CG.f90:188-197
  if (st->status != 0) return;
  const bool first = (st->iter == 1);
  const double beta = st->beta;
  for (int64_t i = (int64_t)blockIdx.x * FX_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * FX_BLOCK)
    p[i] = first ? z[i] : z[i] + beta * p[i];
}
template <bool UPDATE_R>
__global__ __launch_bounds__(FX_BLOCK) void k_nn_cg_xr(int64_t n, const KrylovState *__restrict__ st, const double *__restrict__ p,
                                                       const double *__restrict__ q, double *__restrict__ x, double *__restrict__ r,
                                                       double *__restrict__ partials) {  // :227-240
  if (st->status != 0) return;
  const double alpha = st->alpha;
  double d[1] = {0.0};
  for (int64_t i = (int64_t)blockIdx.x * FX_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * FX_BLOCK) {
    x[i] = x[i] + alpha * p[i];
    if (UPDATE_R) {
      const double rv = r[i] - alpha * q[i];
      r[i] = rv;
      d[0] += rv * rv;
    }
  }
  if (UPDATE_R) block_sum_store<1>(d, partials, 0);
}

static int nn_partials(fx_context *c) {  // the partial-sum buffer of the scalar stages (shared with the 3x3 path)
  if (c->max_partials < 4096 + 8) {
    dev_free(c->partials);
    if (dev_alloc(&c->partials, (size_t)(4096 + 8) * 3)) return FX_ERROR_RUNTIME;
    c->max_partials = 4096 + 8;
  }
  return 0;
}
static int nn_dot_parts(fx_context *c, const double *x, const double *y, const int32_t *gate, int32_t gate_val, int *np) {
  NnDev *n = nn_of(c);
  const int64_t len = (int64_t)n->ndof * n->nn_internal;  // hecmw_InnerProduct_R: NDOF * nn_internal entries
  const int g = grid_for(len, FX_BLOCK, 2048);
  hipLaunchKernelGGL(k_dot, dim3(g), dim3(FX_BLOCK), 0, c->stream, len, x, y, c->partials, gate, gate_val);
  HIP_TRY(hipGetLastError());
  *np = g;
  return 0;
}

static int nn_cg_iteration(fx_context *c, int it, int iterpremax) {  // hecmw_solver_CG.f90:153-271
  NnDev *n = nn_of(c);
  const int64_t nlen = (int64_t)n->ndof * n->N, dlen = (int64_t)n->ndof * n->nn_internal;
  double *X = n->X, *B = n->B, *R = n->W[0], *Z = n->W[1], *Q = n->W[1], *P = n->W[2];
  const int RECOMPUTE = 50, vgrid = grid_for(dlen, FX_BLOCK, 2048);
  int np;
  if (nn_precond_apply(c, iterpremax, R, Z) || nn_dot_parts(c, R, Z, gate_status(c), 0, &np)) return FX_ERROR_RUNTIME;
  if (scalar_stage<OP_CG_RHO>(c, np, 0, RECOMPUTE)) return FX_ERROR_RUNTIME;
  hipLaunchKernelGGL(k_nn_cg_p, dim3(nn_vgrid(nlen)), dim3(FX_BLOCK), 0, c->stream, nlen, c->st, Z, P);
  if (nn_spmv(c, 0, P, nullptr, Q) || nn_dot_parts(c, P, Q, gate_status(c), 0, &np)) return FX_ERROR_RUNTIME;
  if (scalar_stage<OP_CG_C1>(c, np, 0, RECOMPUTE)) return FX_ERROR_RUNTIME;
  if (it % RECOMPUTE == 0) {
    hipLaunchKernelGGL((k_nn_cg_xr<false>), dim3(vgrid), dim3(FX_BLOCK), 0, c->stream, dlen, c->st, P, Q, X, R, c->partials);
    if (nn_spmv(c, 1, X, B, R, gate_status(c), 0) || nn_dot_parts(c, R, R, gate_status(c), 0, &np)) return FX_ERROR_RUNTIME;
  } else {
    hipLaunchKernelGGL((k_nn_cg_xr<true>), dim3(vgrid), dim3(FX_BLOCK), 0, c->stream, dlen, c->st, P, Q, X, R, c->partials);
    np = vgrid;
  }
  if (scalar_stage<OP_RESID>(c, np, 0, RECOMPUTE)) return FX_ERROR_RUNTIME;
  if (it % RECOMPUTE != 0) {  // converged by the recurrence: true residual, re-test (:259-266) -- runs only when the device asks
    if (nn_spmv(c, 1, X, B, R, gate_verify(c), 1) || nn_dot_parts(c, R, R, gate_verify(c), 1, &np)) return FX_ERROR_RUNTIME;
    if (scalar_stage<OP_VERIFY>(c, np, 0, RECOMPUTE)) return FX_ERROR_RUNTIME;
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

static int nn_bicgstab_iteration(fx_context *c, int it, int iterpremax) {  // hecmw_solver_BiCGSTAB.f90:146-264
  NnDev *n = nn_of(c);
  const int64_t dlen = (int64_t)n->ndof * n->nn_internal;
  double *X = n->X, *B = n->B, *R = n->W[0], *RT = n->W[1], *P = n->W[2], *PT = n->W[3], *S = n->W[4], *ST = n->W[0], *T = n->W[5],
         *V = n->W[6];
  const int RECOMPUTE = 100, vgrid = grid_for(dlen, FX_BLOCK, 2048);
  int np;
  if (nn_dot_parts(c, R, RT, gate_status(c), 0, &np) || scalar_stage<OP_BI_RHO>(c, np, 0, RECOMPUTE)) return FX_ERROR_RUNTIME;
  hipLaunchKernelGGL(k_bi_update_p, dim3(vgrid), dim3(FX_BLOCK), 0, c->stream, dlen, c->st, R, V, P);
  if (nn_precond_apply(c, iterpremax, P, PT) || nn_spmv(c, 0, PT, nullptr, V)) return FX_ERROR_RUNTIME;
  if (nn_dot_parts(c, RT, V, gate_status(c), 0, &np) || scalar_stage<OP_BI_C2>(c, np, 0, RECOMPUTE)) return FX_ERROR_RUNTIME;
  hipLaunchKernelGGL(k_bi_update_s, dim3(vgrid), dim3(FX_BLOCK), 0, c->stream, dlen, c->st, R, V, S);
  if (nn_precond_apply(c, iterpremax, S, ST) || nn_spmv(c, 0, ST, nullptr, T)) return FX_ERROR_RUNTIME;  // ST aliases R (:50)
  hipLaunchKernelGGL(k_dot2, dim3(vgrid), dim3(FX_BLOCK), 0, c->stream, dlen, T, S, c->partials, c->max_partials, gate_status(c));
  if (scalar_stage<OP_BI_OMEGA>(c, vgrid, c->max_partials, RECOMPUTE)) return FX_ERROR_RUNTIME;
  if (it % RECOMPUTE == 0) {
    hipLaunchKernelGGL((k_bi_update_xr<false>), dim3(vgrid), dim3(FX_BLOCK), 0, c->stream, dlen, c->st, PT, ST, S, T, X, R, c->partials);
    if (nn_spmv(c, 1, X, B, R, gate_status(c), 0) || nn_dot_parts(c, R, R, gate_status(c), 0, &np)) return FX_ERROR_RUNTIME;
  } else {
    hipLaunchKernelGGL((k_bi_update_xr<true>), dim3(vgrid), dim3(FX_BLOCK), 0, c->stream, dlen, c->st, PT, ST, S, T, X, R, c->partials);
    np = vgrid;
  }
  if (scalar_stage<OP_RESID>(c, np, 0, RECOMPUTE)) return FX_ERROR_RUNTIME;
  if (it % RECOMPUTE != 0) {
    if (nn_spmv(c, 1, X, B, R, gate_verify(c), 1) || nn_dot_parts(c, R, R, gate_verify(c), 1, &np)) return FX_ERROR_RUNTIME;
    if (scalar_stage<OP_VERIFY>(c, np, 0, RECOMPUTE)) return FX_ERROR_RUNTIME;
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// begin (r0, ||b||), then iterations enqueued in chunks with one poll of the device state per chunk
static int nn_krylov(fx_context *c, int method, int MAXIT, double TOL, int iterpremax, NnResult *o) {
  NnDev *n = nn_of(c);
  const int64_t nlen = (int64_t)n->ndof * n->N;
  const size_t vbytes = (size_t)n->ndof * n->NP * 8;
  double *X = n->X, *B = n->B, *R = n->W[0];
  int np;
  if (nn_partials(c) || krylov_init_state(c, MAXIT, TOL)) return FX_ERROR_RUNTIME;
  HIP_TRY(hipMemsetAsync(n->W[2], 0, vbytes, c->stream));               // P (beta = 0 on the first iteration; keep it finite)
  if (method == 2) HIP_TRY(hipMemsetAsync(n->W[6], 0, vbytes, c->stream));  // V
  if (nn_spmv(c, 1, X, B, R)) return FX_ERROR_RUNTIME;                   // CG :120 / BiCGSTAB :107
  if (method == 2) HIP_TRY(hipMemcpyAsync(n->W[1], R, (size_t)nlen * 8, hipMemcpyDeviceToDevice, c->stream));  // r~ = r0
  if (nn_dot_parts(c, B, B, nullptr, 0, &np) || scalar_stage<OP_BNRM2>(c, np, 0, 50)) return FX_ERROR_RUNTIME;
  const int chunk = method == 1 ? 16 : 8;
  KrylovState s;
  memset(&s, 0, sizeof s);
  if (MAXIT <= 0) { if (poll_state(c, &s)) return FX_ERROR_RUNTIME; }
  for (int it = 1; it <= MAXIT; it++) {
    if (method == 1 ? nn_cg_iteration(c, it, iterpremax) : nn_bicgstab_iteration(c, it, iterpremax)) return FX_ERROR_RUNTIME;
    if (it % chunk == 0 || it == MAXIT) {
      if (poll_state(c, &s)) return FX_ERROR_RUNTIME;
      if (s.status != 0) break;
    }
  }
  if (s.status == 0 && poll_state(c, &s)) return FX_ERROR_RUNTIME;
  o->iter = s.iter;
  o->resid = s.resid;
  o->error = s.status > 1 ? s.status : 0;
  o->hist.assign((size_t)std::max(0, s.n_hist), 0.0);
  if (s.n_hist > 0) HIP_TRY(hipMemcpy(o->hist.data(), c->hist, (size_t)s.n_hist * 8, hipMemcpyDeviceToHost));
  return 0;
}
static int nn_cg(fx_context *c, int MAXIT, double TOL, int iterpremax, NnResult *o) { return nn_krylov(c, 1, MAXIT, TOL, iterpremax, o); }
static int nn_bicgstab(fx_context *c, int MAXIT, double TOL, int iterpremax, NnResult *o) { return nn_krylov(c, 2, MAXIT, TOL, iterpremax, o); }

static int nn_scale_bell(fx_context *c, NnBell &b, int back) {
  NnDev *n = nn_of(c);
  if (b.nslices <= 0) return 0;
  NN_DISPATCH(n->ndof, hipLaunchKernelGGL((k_nn_scale_bell<ND>), dim3((b.nslices + 3) / 4), dim3(256), 0, c->stream, b.nslices,
                                          b.slice_ptr, b.slot_row, b.col, b.val, n->scale, back))
  HIP_TRY(hipGetLastError());
  return 0;
}
// forward: scale vector (+ its halo), matrix, right-hand side; back: x, right-hand side and matrix restored
static int nn_scaling(fx_context *c, int back) {
  NnDev *n = nn_of(c);
  const int nd = n->ndof;
  const size_t len = (size_t)nd * std::max(n->NP, 1);
  if (!n->scale && dev_alloc(&n->scale, len)) return FX_ERROR_RUNTIME;
  if (n->N <= 0) return 0;
  if (!back) {
    HIP_TRY(hipMemsetAsync(n->scale, 0, len * 8, c->stream));
    hipLaunchKernelGGL(k_nn_scale_vec, dim3(((int64_t)n->N * nd + 255) / 256), dim3(256), 0, c->stream, n->N, nd, n->D, n->scale);
    if (nn_halo(c, n->scale)) return FX_ERROR_RUNTIME;
  } else {
    hipLaunchKernelGGL(k_nn_scale_rhs, dim3(nn_vgrid((int64_t)nd * n->N)), dim3(256), 0, c->stream, (int64_t)nd * n->N, n->B, n->X,
                       n->scale, 1);
  }
  hipLaunchKernelGGL(k_nn_scale_diag, dim3(((int64_t)n->NP * nd * nd + 255) / 256), dim3(256), 0, c->stream, n->NP, nd, n->D, n->scale, back);
  if (nn_scale_bell(c, n->M, back)) return FX_ERROR_RUNTIME;
  if (!back)
    hipLaunchKernelGGL(k_nn_scale_rhs, dim3(nn_vgrid((int64_t)nd * n->N)), dim3(256), 0, c->stream, (int64_t)nd * n->N, n->B, n->X,
                       n->scale, 0);
  HIP_TRY(hipGetLastError());
  return 0;
}

// GMRES(m) and GPBiCG of fx_krylov2_host.h on the generic-block system
struct OpsNN {
  fx_context *c;
  NnDev *n() const { return nn_of(c); }
  double *X() const { return n()->X; }
  double *B() const { return n()->B; }
  double *W(int k) const { return n()->W[k]; }
  double *extra(int k) const { return n()->extra + (size_t)k * n()->ndof * n()->NP; }
  int64_t veclen() const { return (int64_t)n()->ndof * n()->N; }
  size_t wbytes() const { return (size_t)n()->ndof * n()->NP * 8; }
  int64_t ndof_np() const { return (int64_t)n()->ndof * n()->NP; }
  int spmv(int mode, double *x, const double *b, double *y) const { return nn_spmv(c, mode, x, b, y); }
  int precond(const double *r, double *z) const { return nn_precond_apply(c, n()->iterpremax, r, z); }
  int dot(const double *x, const double *y, double *out) const { return nn_dot(c, x, y, out); }
  int prepare(int, double, int count) const {
    NnDev *d = n();
    const size_t len = (size_t)d->ndof * d->NP;
    for (int k = 0; k < 8; k++) HIP_TRY(hipMemsetAsync(d->W[k], 0, len * 8, c->stream));
    if (d->extra_n < count) {
      dev_free(d->extra);
      if (dev_alloc(&d->extra, (size_t)count * len)) return FX_ERROR_RUNTIME;
      d->extra_n = count;
    }
    HIP_TRY(hipMemsetAsync(d->extra, 0, (size_t)count * len * 8, c->stream));
    return 0;
  }
};

// hecmw_solve for NDOF != 3: host arrays in, host X out (fx_solve forwards here)
static int nn_solve(fx_context *c, const fx_matrix_view *m, const fx_comm_view *cm, int32_t *Iarray, double *Rarray,
                    fx_solve_info *info, double *hist, int32_t hist_len) {
  HIP_TRY(hipSetDevice(c->device));
  if (m->NDOF < 1 || m->NDOF > 6) { g_fx_error = "NDOF must be 1..6"; return FX_ERROR_UNSUPPORTED; }
  c->view_petot = cm ? std::max(1, (int)cm->PETOT) : 1;
  if (require_transport(c, "fx_solve")) return FX_ERROR_RUNTIME;
  const int maxit = Iarray[0], precond = Iarray[2], method2 = Iarray[7], iterpremax = Iarray[4];
  int method = Iarray[1];
  const bool scaling = Iarray[6] != 0;  // SCALING=YES
  const double t0 = now_s();
  NnDev *n0 = nn_of(c);
  const bool values_changed = Iarray[97] >= 1 || Iarray[96] >= 1 || !n0->have_matrix ||
                              m->D != c->host_D || m->AL != c->host_AL || m->AU != c->host_AU;  // see fx_solve
  if (nn_upload(c, m, cm, values_changed)) return FX_ERROR_RUNTIME;
  if (values_changed) { c->host_D = m->D; c->host_AL = m->AL; c->host_AU = m->AU; }
  NnDev *n = nn_of(c);
  n->cur_AL = m->AL; n->cur_AU = m->AU;  // the sweep layouts of a preconditioner (re)built in this call are gathered from these
  const int nd = n->ndof;
  int ret = 0;
  double rhs2 = 0.0;
  if (nn_dot(c, n->B, n->B, &rhs2)) return FX_ERROR_RUNTIME;  // hecmw_solve_check_zerorhs (:242-278)
  if (rhs2 == 0.0) { ret = FX_ERROR_ZERO_RHS; HIP_TRY(hipMemsetAsync(n->X, 0, (size_t)nd * n->NP * 8, c->stream)); }
  {  // hecmw_solve_check_zerodiag (:212-240)
    int32_t *flag = (int32_t *)(n->scal + 4), hflag = 0;
    HIP_TRY(hipMemsetAsync(flag, 0, 4, c->stream));
    if (n->N > 0)
      hipLaunchKernelGGL(k_nn_check_zero_diag, dim3(((int64_t)n->N * nd + 255) / 256), dim3(256), 0, c->stream, n->N, nd, n->D, flag);
    HIP_TRY(hipMemcpyAsync(&hflag, flag, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (multi_rank(c)) {
      double f = hflag ? 1.0 : 0.0;
      HIP_TRY(hipMemcpyAsync(n->scal + 2, &f, 8, hipMemcpyHostToDevice, c->stream));
      if (allreduce_dev(c, n->scal + 2, 1)) return FX_ERROR_RUNTIME;
      HIP_TRY(hipMemcpyAsync(&f, n->scal + 2, 8, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      hflag = f > 0.0;
    }
    if (hflag && precond < 10 && iterpremax > 0) return FX_ERROR_ZERO_DIAG;
  }
  // hecmw_mat_recycle_precond_setting (hecmw_matrix_misc.f90:678-697)
  if (Iarray[97] >= 1) { Iarray[96] = 1; Iarray[95] = 0; }
  else if (Iarray[96] > 1) { Iarray[95] = 0; Iarray[96] = 1; }
  else if (Iarray[96] == 1) {
    if (Iarray[95] < Iarray[34]) { Iarray[96] = 0; Iarray[95]++; }
    else Iarray[95] = 0;
  }
  double sigma = Rarray[1] < 0.0 ? 1.0 : Rarray[1];
  if (iterpremax > 0 && (!n->precond_valid || Iarray[97] == 1 || Iarray[96] == 1 || n->sigma != sigma ||
                         n->precond_kind != ((precond == 3) ? 3 : 1))) {
    if (int e = nn_precond_setup(c, precond, sigma, Iarray[33])) return e;
  }
  Iarray[97] = 0; Iarray[96] = 0;
  const double t_setup = now_s() - t0, t1 = now_s();
  NnResult res;
  c->attempts.clear();
  for (;;) {
    Iarray[80] = 0; Iarray[81] = 0;
    res = NnResult();
    int e;
    n->iterpremax = iterpremax;
    if (scaling) {  // scale, then build the preconditioner of the scaled matrix (hecmw_solver_CG.f90:104-112)
      if (nn_scaling(c, 0)) return FX_ERROR_RUNTIME;
      if (iterpremax > 0) {
        if (int pe = nn_precond_setup(c, precond, sigma, Iarray[33])) return pe;
        if (n->precond_kind == 1 && (nn_scale_bell(c, n->L, 0) || nn_scale_bell(c, n->U, 0))) return FX_ERROR_RUNTIME;  // built from the caller's values
      }
    }
    if (method == 1) e = nn_cg(c, maxit, Rarray[0], iterpremax, &res);
    else if (method == 2) e = nn_bicgstab(c, maxit, Rarray[0], iterpremax, &res);
    else if (method == 3 || method == 4) {
      HostKrylov hk;
      e = (method == 3) ? gmres_solve_t(OpsNN{c}, maxit, Rarray[0], Iarray[5], &hk) : gpbicg_solve_t(OpsNN{c}, maxit, Rarray[0], &hk);
      res.iter = hk.iter; res.resid = hk.resid; res.error = hk.error; res.hist = hk.hist;
    } else { g_fx_error = "METHOD must be 1 (CG), 2 (BiCGSTAB), 3 (GMRES) or 4 (GPBiCG)"; return FX_ERROR_INCONS_PC; }
    if (e) return e;
    c->attempts.emplace_back();
    c->attempts.back().method = method;
    c->attempts.back().sigma_diag = sigma;
    c->attempts.back().hist = res.hist;
    if (scaling) {  // x and b back, matrix restored (hecmw_solver_CG.f90:277); the preconditioner belonged to the scaled matrix
      if (nn_scaling(c, 1)) return FX_ERROR_RUNTIME;
      n->precond_valid = false;
    }
    if (res.error == FX_ERROR_DIVERGE_PC || res.error == FX_ERROR_DIVERGE_MAT) {  // Iterative.f90:145-156
      Iarray[81] = 1;
      if (method == 1 && method2 > 1) { method = method2; continue; }
    }
    break;
  }
  if (nn_halo(c, n->X)) return FX_ERROR_RUNTIME;  // hecmw_update_m_R (hecmw_solver_CG.f90:280)
  if (res.error > 1) ret = res.error;
  // hecmw_rel_resid_L2 (hecmw_solver_las.f90:129-158)
  double r2 = 0.0, b2 = rhs2 == 0.0 ? 1.0 : rhs2;
  if (nn_spmv(c, 1, n->X, n->B, n->W[7]) || nn_dot(c, n->W[7], n->W[7], &r2)) return FX_ERROR_RUNTIME;
  const double resid2 = sqrt(r2 / b2);
  if (resid2 < Rarray[0]) Iarray[80] = 1;
  HIP_TRY(hipMemcpy(m->X, n->X, (size_t)nd * n->NP * 8, hipMemcpyDeviceToHost));
  const int nh = std::max(0, std::min((int)hist_len, (int)res.hist.size()));
  if (info) {
    memset(info, 0, sizeof *info);
    info->iterations = res.iter; info->method = method; info->precond = precond;
    info->ncolor = n->precond_kind == 1 ? n->ncolor : 0;
    info->resid = res.resid; info->rel_resid = resid2;
    info->time_setup = t_setup; info->time_sol = now_s() - t1;
    info->n_hist = hist ? nh : 0;
  }
  if (hist && nh > 0) memcpy(hist, res.hist.data(), (size_t)nh * 8);
  return ret;
}

// hecmw_matvec for NDOF != 3: Y(1:NDOF*N) = A X, the halo part of X is updated
static int nn_matvec(fx_context *c, const fx_matrix_view *m, const fx_comm_view *cm, double *x, double *y, double *commtime) {
  HIP_TRY(hipSetDevice(c->device));
  NnDev *n0 = nn_of(c);
  fx_matrix_view mv = *m;
  mv.B = nullptr; mv.X = nullptr;
  if (!m->D) {  // "use the resident values" (see fx_matvec)
    if (!n0->have_matrix || n0->ndof != m->NDOF || n0->N != m->N || n0->NP != m->NP || n0->NPL != m->NPL || n0->NPU != m->NPU) {
      g_fx_error = "fx_matvec: mat->D is NULL but no matrix values are resident";
      return FX_ERROR_RUNTIME;
    }
  } else {
    if (nn_upload(c, &mv, cm, true)) return FX_ERROR_RUNTIME;  // no change flag on this entry: values uploaded every call
    c->host_D = m->D; c->host_AL = m->AL; c->host_AU = m->AU;   // whose values are resident now (see fx_solve)
  }
  NnDev *n = nn_of(c);
  const size_t len = (size_t)n->ndof * n->NP * 8;
  HIP_TRY(hipMemcpyAsync(n->W[6], x, len, hipMemcpyHostToDevice, c->stream));
  const double t0 = now_s();
  if (nn_halo(c, n->W[6])) return FX_ERROR_RUNTIME;
  if (commtime) { HIP_TRY(hipStreamSynchronize(c->stream)); *commtime += now_s() - t0; }
  if (nn_spmv(c, 0, n->W[6], nullptr, n->W[7])) return FX_ERROR_RUNTIME;
  HIP_TRY(hipMemcpyAsync(y, n->W[7], (size_t)n->ndof * n->N * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(x, n->W[6], len, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// timed resident SpMV of the NDOF != 3 layout (bench / profiles)
static int nn_matvec_resident(fx_context *c, int nrepeat, float *ms_per_call) {
  NnDev *n = nn_of(c);
  if (!n->have_matrix) { g_fx_error = "no NDOF != 3 matrix resident"; return FX_ERROR_RUNTIME; }
  if (nn_spmv(c, 0, n->W[6], nullptr, n->W[7])) return FX_ERROR_RUNTIME;
  HIP_TRY(hipEventRecord(c->ev0, c->stream));
  for (int r = 0; r < nrepeat; r++)
    if (nn_spmv(c, 0, n->W[6], nullptr, n->W[7])) return FX_ERROR_RUNTIME;
  HIP_TRY(hipEventRecord(c->ev1, c->stream));
  HIP_TRY(hipEventSynchronize(c->ev1));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  if (ms_per_call) *ms_per_call = ms / std::max(nrepeat, 1);
  return 0;
}
