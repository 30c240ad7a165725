// Device kernels of libfistr_hip (gfx950 / CDNA4, wave64).  Included once by fistr_hip.hip.
//
// All sweep kernels stream the BELL-64 layout (fx_internal.h): one thread per
// block row, every wave-level load a full 1 KiB / 512 B coalesced segment, no
// cross-lane reduction in the row loop.  Everything on this path is bound by HBM
// bandwidth (SpMV arithmetic intensity ~0.23 flop/B), so
// MFMA is deliberately not used: a 3x3 block per 72 B offers no dense tile.
#pragma once
#include "fx_internal.h"

#define FX_BLOCK 256

// XCD-aware block order (cdna_hip_programming.md T1): hardware deals workgroups round-robin
// over the 8 XCDs, each with a private 4 MiB L2.  Remapping so that every XCD walks one
// CONTIGUOUS eighth of the slices lets the x / z gathers of neighbouring rows hit the same
// L2 (bijective for any grid size).  Pure speed: results do not depend on placement.
__device__ __forceinline__ int xcd_block(int b, int nb) {
  const int q = nb >> 3, r = nb & 7, x = b & 7, k = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

// ------------------------------------------------------------------------
// reductions: wave shuffle (64 lanes) then LDS across the 4 waves, fixed order
// => bitwise reproducible run to run.
// ------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

template <int NV, int BS = FX_BLOCK>
__device__ __forceinline__ void block_sum_store(double (&v)[NV], double *out, int stride, int slot = -1) {
  if (slot < 0) slot = blockIdx.x;
  __shared__ double sm[NV][BS / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; i++) {
    double s = wave_sum(v[i]);
    if (lane == 0) sm[i][w] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NV; i++) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < BS / 64; k++) s += sm[i][k];
      out[(size_t)i * stride + slot] = s;
    }
  }
}

// ------------------------------------------------------------------------
// BELL fill: gather 3x3 blocks of the reference arrays into the sliced layout.
// src code: 3*idx + which (0 = D, 1 = AL, 2 = AU), idx 0-based block; -1 = padding.
// Run once per numeric refresh (Iarray(97)); reads 72 B segments, writes coalesced.
// ------------------------------------------------------------------------
__global__ void k_bell_fill(int32_t nslices, const int32_t *__restrict__ half_ptr, const int32_t *__restrict__ src,
                            const double *__restrict__ D, const double *__restrict__ AL,
                            const double *__restrict__ AU, double *__restrict__ val) {
  const int slice = blockIdx.x * (FX_BLOCK / 64) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (slice >= nslices) return;
  const int h0 = half_ptr[slice], h1 = half_ptr[slice + 1];
  const int npair2 = ((h1 - h0) >> 1) << 1;  // blocks stored as 16-byte pairs; an odd last block is stored alone
  for (int k = 0; k < h1 - h0; k++) {
    const bool paired = k < npair2;
    const size_t hb = (size_t)(h0 + (paired ? (k & ~1) : k));
    const int sc = paired ? src[hb * 64 + lane * 2 + (k & 1)] : src[hb * 64 + lane];
    double a[9];
#pragma unroll
    for (int e = 0; e < 9; e++) a[e] = 0.0;
    if (sc >= 0) {
      const int w = sc % 3;
      const double *base = (w == 0 ? D : (w == 1 ? AL : AU)) + (size_t)9 * (sc / 3);
#pragma unroll
      for (int e = 0; e < 9; e++) a[e] = base[e];
    }
#pragma unroll
    for (int e = 0; e < 9; e++) {
      if (paired) val[hb * 576 + (size_t)(e * 64 + lane) * 2 + (k & 1)] = a[e];
      else val[hb * 576 + (size_t)e * 64 + lane] = a[e];
    }
  }
}

// ------------------------------------------------------------------------
// The row loop shared by SpMV and the SSOR sweeps:  s += sum_k A_k * x[col_k] over the
// block pairs [p0,p1) of this lane's row.  PIPE = false: plain loop (two dependent memory
// latencies per pair: column ids, then the gather).  PIPE = true: 2-deep software
// pipeline -- while pair i is being multiplied, the 9 value words and the 6 gathered
// vector entries of pair i+1 and the column ids of pair i+2 are already in flight, so a
// wave exposes at most one latency per pair and keeps ~2x the bytes in flight.
// ------------------------------------------------------------------------
// The matrix stream (values + column ids) is read exactly once per sweep: load it non-temporally so
// that it does not evict the gathered vector entries from L2 / Infinity Cache.  Measured on MI355X,
// 10.1M DOF, same process pair: SpMV 1.323 -> 1.246 ms, SSOR apply 1.956 -> 1.81 ms, CG+SSOR 285 -> 301 it/s.
typedef double fx_d2 __attribute__((ext_vector_type(2)));
typedef int fx_i2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ld_stream(const double2 *p) {
  const fx_d2 v = __builtin_nontemporal_load((const fx_d2 *)p);
  return make_double2(v.x, v.y);
}
__device__ __forceinline__ int2 ld_stream(const int2 *p) {
  const fx_i2 v = __builtin_nontemporal_load((const fx_i2 *)p);
  return make_int2(v.x, v.y);
}

__device__ __forceinline__ double ld_stream(const double *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ int ld_stream(const int *p) { return __builtin_nontemporal_load(p); }

// An odd last block of the slice is stored alone (8-byte words) instead of padding the pair:
// rows of a hex mesh have 27 blocks, so pair padding alone would cost 1/27 = 3.7 % of the stream.
__device__ __forceinline__ void bell_tail_block(const double *__restrict__ vt, const int *__restrict__ ct,
                                                const double *__restrict__ x, double &s0, double &s1, double &s2) {
  const int cc = ld_stream(ct);
  double a[9];
#pragma unroll
  for (int e = 0; e < 9; e++) a[e] = ld_stream(vt + e * 64);
  const double *xa = x + (size_t)3 * cc;
  const double xa0 = xa[0], xa1 = xa[1], xa2 = xa[2];
  s0 += a[0] * xa0 + a[1] * xa1 + a[2] * xa2;
  s1 += a[3] * xa0 + a[4] * xa1 + a[5] * xa2;
  s2 += a[6] * xa0 + a[7] * xa1 + a[8] * xa2;
}

template <bool PIPE>
__device__ __forceinline__ void bell_row_sweep(int h0, int h1, const double *__restrict__ val,
                                               const int *__restrict__ col, int lane,
                                               const double *__restrict__ x, double &s0, double &s1, double &s2) {
  const int np = (h1 - h0) >> 1;
  if ((h1 - h0) & 1)
    bell_tail_block(val + (size_t)(h0 + 2 * np) * 576 + lane, col + (size_t)(h0 + 2 * np) * 64 + lane, x, s0, s1, s2);
  if (np <= 0) return;
  const double2 *v = (const double2 *)(val + (size_t)h0 * 576) + lane;
  const int2 *c = (const int2 *)(col + (size_t)h0 * 64) + lane;
  if (!PIPE) {
    for (int i = 0; i < np; i++, v += 576, c += 64) {
      const int2 cc = ld_stream(c);
      double2 a[9];
#pragma unroll
      for (int e = 0; e < 9; e++) a[e] = ld_stream(v + e * 64);
      const double *xa = x + (size_t)3 * cc.x, *xb = x + (size_t)3 * cc.y;
      const double xa0 = xa[0], xa1 = xa[1], xa2 = xa[2];
      const double xb0 = xb[0], xb1 = xb[1], xb2 = xb[2];
      s0 += a[0].x * xa0 + a[1].x * xa1 + a[2].x * xa2;
      s1 += a[3].x * xa0 + a[4].x * xa1 + a[5].x * xa2;
      s2 += a[6].x * xa0 + a[7].x * xa1 + a[8].x * xa2;
      s0 += a[0].y * xb0 + a[1].y * xb1 + a[2].y * xb2;
      s1 += a[3].y * xb0 + a[4].y * xb1 + a[5].y * xb2;
      s2 += a[6].y * xb0 + a[7].y * xb1 + a[8].y * xb2;
    }
    return;
  }
  int2 cc1 = (np > 1) ? ld_stream(c + 64) : ld_stream(c);
  double2 a[9];
  double xa0, xa1, xa2, xb0, xb1, xb2;
  {
    const int2 cc0 = ld_stream(c);
#pragma unroll
    for (int e = 0; e < 9; e++) a[e] = ld_stream(v + e * 64);
    const double *xa = x + (size_t)3 * cc0.x, *xb = x + (size_t)3 * cc0.y;
    xa0 = xa[0]; xa1 = xa[1]; xa2 = xa[2];
    xb0 = xb[0]; xb1 = xb[1]; xb2 = xb[2];
  }
  for (int i = 0; i < np; i++) {
    const bool more = (i + 1 < np);  // wave-uniform: p0/p1 belong to the slice
    double2 an[9];
    double xan0 = 0, xan1 = 0, xan2 = 0, xbn0 = 0, xbn1 = 0, xbn2 = 0;
    int2 cc2 = cc1;
    if (more) {
      v += 576; c += 64;
#pragma unroll
      for (int e = 0; e < 9; e++) an[e] = ld_stream(v + e * 64);
      const double *xa = x + (size_t)3 * cc1.x, *xb = x + (size_t)3 * cc1.y;
      xan0 = xa[0]; xan1 = xa[1]; xan2 = xa[2];
      xbn0 = xb[0]; xbn1 = xb[1]; xbn2 = xb[2];
      if (i + 2 < np) cc2 = ld_stream(c + 64);
    }
    s0 += a[0].x * xa0 + a[1].x * xa1 + a[2].x * xa2;
    s1 += a[3].x * xa0 + a[4].x * xa1 + a[5].x * xa2;
    s2 += a[6].x * xa0 + a[7].x * xa1 + a[8].x * xa2;
    s0 += a[0].y * xb0 + a[1].y * xb1 + a[2].y * xb2;
    s1 += a[3].y * xb0 + a[4].y * xb1 + a[5].y * xb2;
    s2 += a[6].y * xb0 + a[7].y * xb1 + a[8].y * xb2;
    if (more) {
#pragma unroll
      for (int e = 0; e < 9; e++) a[e] = an[e];
      xa0 = xan0; xa1 = xan1; xa2 = xan2; xb0 = xbn0; xb1 = xbn1; xb2 = xbn2;
      cc1 = cc2;
    }
  }
}

// ------------------------------------------------------------------------
// K1/K2: y = A x  (hecmw_matvec_33_inner, hecmw_solver_las_33.f90:263-300)
// MODE 0: y = A x            MODE 1: y = b - A x (hecmw_matresid_33 :371-379)
// DOT  0: none  1: partial of x.y (p.q in CG)  2: partial of y.y (||r||^2 after matresid)
// `gate` (may be null): device status word; the kernel is a no-op unless *gate == gate_val.
// ------------------------------------------------------------------------
template <int MODE, int DOT, bool PIPE, int BS>
__global__ __launch_bounds__(BS) void k_spmv(int32_t nslices, int32_t nrows,
                                                   const int32_t *__restrict__ pair_ptr,
                                                   const double *__restrict__ val2,
                                                   const int *__restrict__ col2,
                                                   const double *__restrict__ x, const double *__restrict__ b,
                                                   double *__restrict__ y, double *__restrict__ partials,
                                                   const int32_t *__restrict__ gate, int32_t gate_val,
                                                   const int32_t *__restrict__ slice_order,
                                                   const int32_t *__restrict__ wg_list) {
  if (gate && *gate != gate_val) return;
  // wg_list (may be null): the virtual workgroups of THIS launch -- the interior / boundary halves of a domain-decomposed
  // product (spmv(), fistr_hip.hip).  A virtual workgroup always covers the same four slices and owns the same partial-sum
  // slot, so the two halves together are bit-identical to the one launch over all workgroups.
  int vb = xcd_block(blockIdx.x, gridDim.x);
  if (wg_list) vb = wg_list[vb];
  int slice = vb * (BS / 64) + (threadIdx.x >> 6);
  const bool live = slice < nslices;
  if (live && slice_order) slice = slice_order[slice];
  if (!live) slice = nslices;
  const int lane = threadIdx.x & 63;
  double y0 = 0.0, y1 = 0.0, y2 = 0.0;
  const int row = slice * 64 + lane;
  if (slice < nslices)
    bell_row_sweep<PIPE>(pair_ptr[slice], pair_ptr[slice + 1], val2, col2, lane, x, y0, y1, y2);
  double d[1] = {0.0};
  if (slice < nslices) {
    if (MODE == 1) {
      y0 = b[(size_t)3 * row] - y0; y1 = b[(size_t)3 * row + 1] - y1; y2 = b[(size_t)3 * row + 2] - y2;
    }
    y[(size_t)3 * row] = y0; y[(size_t)3 * row + 1] = y1; y[(size_t)3 * row + 2] = y2;
    if (DOT == 1) d[0] = x[(size_t)3 * row] * y0 + x[(size_t)3 * row + 1] * y1 + x[(size_t)3 * row + 2] * y2;
    if (DOT == 2) d[0] = y0 * y0 + y1 * y1 + y2 * y2;
  }
  if (DOT != 0) block_sum_store<1, BS>(d, partials, 0, vb);
}

// ------------------------------------------------------------------------
// 3x3 LU helpers (hecmw_precond_DIAG_33.f90:96-105 and :140-144; identical code
// in hecmw_precond_SSOR_33.f90:190-201, :341-345)
// ------------------------------------------------------------------------
__device__ __forceinline__ void lu33_dev(double *a) {
#pragma unroll
  for (int k = 0; k < 3; k++) {
    a[4 * k] = 1.0 / a[4 * k];
#pragma unroll
    for (int i = k + 1; i < 3; i++) {
      a[3 * i + k] = a[3 * i + k] * a[4 * k];
#pragma unroll
      for (int j = k + 1; j < 3; j++) a[3 * i + j] = a[3 * i + j] - a[3 * i + k] * a[3 * k + j];
    }
  }
}

__device__ __forceinline__ void lusolve33_dev(const double *u, double &x1, double &x2, double &x3) {
  x2 = x2 - u[3] * x1;
  x3 = x3 - u[6] * x1 - u[7] * x2;
  x3 = u[8] * x3;
  x2 = u[4] * (x2 - u[5] * x3);
  x1 = u[0] * (x1 - u[2] * x3 - u[1] * x2);
}

// D~ x from the in-place LU factors of lu33_dev (unit lower L, upper U with the RECIPROCALS of its pivots on the diagonal):
// y = U x, then L y.  Also returns the diagonal of D~ (what SIGMA_DIAG scaled).  Lets Eisenstat's form stream the factors only --
// the unfactored diagonal blocks need no array of their own.
__device__ __forceinline__ void lumul33_dev(const double *u, double x1, double x2, double x3, double &y1, double &y2, double &y3,
                                            double &d1, double &d2, double &d3) {
  const double p1 = 1.0 / u[0], p2 = 1.0 / u[4], p3 = 1.0 / u[8];
  const double t1 = p1 * x1 + u[1] * x2 + u[2] * x3;
  const double t2 = p2 * x2 + u[5] * x3;
  const double t3 = p3 * x3;
  y1 = t1;
  y2 = u[3] * t1 + t2;
  y3 = u[6] * t1 + u[7] * t2 + t3;
  d1 = p1;
  d2 = u[3] * u[1] + p2;
  d3 = u[6] * u[2] + u[7] * u[5] + p3;
}

// ALU setup: slot_row (may be null = identity) maps slot -> 0-based node; D in reference layout.
// Output layout [slice][e][lane].
__global__ void k_alu_setup(int32_t nslots, int32_t nrows, const int32_t *__restrict__ slot_row,
                            const double *__restrict__ D, double sigma_diag, double *__restrict__ alu) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= nslots) return;
  const int row = slot_row ? slot_row[slot] : slot;
  double a[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (row >= 0 && row < nrows) {
#pragma unroll
    for (int e = 0; e < 9; e++) a[e] = D[(size_t)9 * row + e];
    a[0] *= sigma_diag; a[4] *= sigma_diag; a[8] *= sigma_diag;
    lu33_dev(a);
  }
  const size_t base = (size_t)(slot >> 6) * 576 + (slot & 63);
#pragma unroll
  for (int e = 0; e < 9; e++) alu[base + (size_t)e * 64] = a[e];
}

// K6: block-Jacobi apply z = D~^-1 r (hecmw_precond_DIAG_33.f90:125-152), fused with the
// prologue/epilogue of hecmw_precond_apply (Z = 0 + ZP) and the partial of r.z.
__global__ __launch_bounds__(FX_BLOCK) void k_diag_apply(int32_t nrows, const double *__restrict__ alu,
                                                         const double *__restrict__ r, double *__restrict__ z,
                                                         double *__restrict__ partials,
                                                         const int32_t *__restrict__ gate) {
  if (gate && *gate != 0) return;
  const int row = blockIdx.x * FX_BLOCK + threadIdx.x;
  double d[1] = {0.0};
  if (row < nrows) {
    double u[9];
    const size_t base = (size_t)(row >> 6) * 576 + (row & 63);
#pragma unroll
    for (int e = 0; e < 9; e++) u[e] = alu[base + (size_t)e * 64];
    const double r1 = r[(size_t)3 * row], r2 = r[(size_t)3 * row + 1], r3 = r[(size_t)3 * row + 2];
    double x1 = r1, x2 = r2, x3 = r3;
    lusolve33_dev(u, x1, x2, x3);
    z[(size_t)3 * row] = x1; z[(size_t)3 * row + 1] = x2; z[(size_t)3 * row + 2] = x3;
    d[0] = r1 * x1 + r2 * x2 + r3 * x3;
  }
  if (partials) block_sum_store<1>(d, partials, 0);
}

// K7: one colour of the multicolour block SSOR sweep (hecmw_precond_SSOR_33.f90:300-352
// forward, :355-410 backward).  Rows of one colour are independent.
// The sweep works on a private vector zs in COLOUR-MAJOR slot numbering (each colour a
// contiguous range, natural node order inside it), so the 13+13 neighbour gathers of
// adjacent lanes hit adjacent entries and use full cache lines; the Krylov vectors r, z stay
// in the natural numbering (where the SpMV gathers are best) and are touched only through
// the row's own 24 bytes: r_i is read from r[node], the final z_i is written to z[node] by
// the backward sweep.  (The reference keeps ZP in the old numbering and indexes it through
// perm -- same arithmetic, different addresses.)
//   FWD: zs_i = D~_i^-1 ( r_i - sum_{j in L(i)} L_ij zs_j )
//   BWD: zs_i = zs_i - D~_i^-1 sum_{j in U(i)} U_ij zs_j ;  z[node_i] = zs_i  (+ partial of r.z)
template <bool FWD, bool PIPE, int BS>
__global__ __launch_bounds__(BS) void k_ssor_color(int32_t slice0, int32_t slice1,
                                                         const int32_t *__restrict__ pair_ptr,
                                                         const double *__restrict__ val2,
                                                         const int *__restrict__ col2,
                                                         const int32_t *__restrict__ slot_node,
                                                         const double *__restrict__ alu,
                                                         const double *__restrict__ r, double *__restrict__ zs,
                                                         double *__restrict__ z, double *__restrict__ partials,
                                                         const int32_t *__restrict__ gate, int32_t spw) {
  if (gate && *gate != 0) return;
  // spw consecutive slices per wave (1 by default; FX_SSOR_SPW): a wave that walks several slices pays its start-up once
  const int vb = xcd_block(blockIdx.x, gridDim.x);
  const int lane = threadIdx.x & 63;
  double d[1] = {0.0};
  const int first = slice0 + (vb * (BS / 64) + (threadIdx.x >> 6)) * spw;
  for (int slice = first; slice < first + spw && slice < slice1; slice++) {
    const int slot = slice * 64 + lane;
    const int node = slot_node ? slot_node[slot] : slot;  // null: r and z already live in slot numbering
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    bell_row_sweep<PIPE>(pair_ptr[slice], pair_ptr[slice + 1], val2, col2, lane, zs, s0, s1, s2);
    if (node >= 0) {
      double u[9];
      const size_t base = (size_t)slice * 576 + lane;
#pragma unroll
      for (int e = 0; e < 9; e++) u[e] = alu[base + (size_t)e * 64];
      double *zi = zs + (size_t)3 * slot;
      const double *ri = r + (size_t)3 * node;
      if (FWD) {
        double x1 = ri[0] - s0, x2 = ri[1] - s1, x3 = ri[2] - s2;
        lusolve33_dev(u, x1, x2, x3);
        zi[0] = x1; zi[1] = x2; zi[2] = x3;
      } else {
        lusolve33_dev(u, s0, s1, s2);
        const double x1 = zi[0] - s0, x2 = zi[1] - s1, x3 = zi[2] - s2;
        zi[0] = x1; zi[1] = x2; zi[2] = x3;
        if (z) {
          double *zn = z + (size_t)3 * node;
          zn[0] = x1; zn[1] = x2; zn[2] = x3;
        }
        if (partials) d[0] += ri[0] * x1 + ri[1] * x2 + ri[2] * x3;
      }
    }
  }
  if (!FWD && partials) block_sum_store<1, BS>(d, partials, 0, vb);
}

// Latency-bound colours and ILU(0) levels (a few dozen to a few thousand slices: less than two waves per SIMD, so a
// wave's chain of dependent loads -- column ids -> gathered z -> next pair -- is the whole kernel time, 9.3 us per
// level at 10M DOF).  WPS waves share one slice: wave w takes the block pairs w, w+WPS, ... (7 pairs for a hex-mesh
// lower part => at most two per wave at WPS=4), issues every load of its pairs before the first use, and the partial
// sums meet in LDS in a fixed order.  Same layout, same coalescing, 1/WPS of the dependent depth.
// The contraction is written out (which product is fused into which sum) so that every kernel that shares these helpers
// rounds identically: k_ssor_color_split and k_tri_dataflow are bit-identical by construction, not by the compiler's mood.
__device__ __forceinline__ double bell_dot3(double a0, double a1, double a2, double x0, double x1, double x2) {
  return fma(a2, x2, fma(a1, x1, a0 * x0));
}
__device__ __forceinline__ void bell_pair_fma(const double2 (&a)[9], const double (&xv)[6], double &s0, double &s1, double &s2) {
  s0 += bell_dot3(a[0].x, a[1].x, a[2].x, xv[0], xv[1], xv[2]);
  s1 += bell_dot3(a[3].x, a[4].x, a[5].x, xv[0], xv[1], xv[2]);
  s2 += bell_dot3(a[6].x, a[7].x, a[8].x, xv[0], xv[1], xv[2]);
  s0 += bell_dot3(a[0].y, a[1].y, a[2].y, xv[3], xv[4], xv[5]);
  s1 += bell_dot3(a[3].y, a[4].y, a[5].y, xv[3], xv[4], xv[5]);
  s2 += bell_dot3(a[6].y, a[7].y, a[8].y, xv[3], xv[4], xv[5]);
}
__device__ __forceinline__ void bell_single_fma(const double (&a)[9], const double (&xv)[3], double &s0, double &s1, double &s2) {
  s0 += bell_dot3(a[0], a[1], a[2], xv[0], xv[1], xv[2]);
  s1 += bell_dot3(a[3], a[4], a[5], xv[0], xv[1], xv[2]);
  s2 += bell_dot3(a[6], a[7], a[8], xv[0], xv[1], xv[2]);
}

template <bool FWD, int WPS>
__global__ __launch_bounds__(64 * WPS) void k_ssor_color_split(int32_t slice0, int32_t slice1,
                                                               const int32_t *__restrict__ pair_ptr,
                                                               const double *__restrict__ val2,
                                                               const int *__restrict__ col2,
                                                               const int32_t *__restrict__ slot_node,
                                                               const double *__restrict__ alu,
                                                               const double *__restrict__ r, double *__restrict__ zs,
                                                               double *__restrict__ z, double *__restrict__ partials,
                                                               const int32_t *__restrict__ gate) {
  __shared__ double part[WPS][3][64];
  const int slice = slice0 + blockIdx.x;  // grid = slice1 - slice0
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int32_t gv = gate ? *gate : 0;  // one scalar round trip for the gate AND the slice's row pointers (the early return used to serialise them)
  const int h0 = pair_ptr[slice], h1 = pair_ptr[slice + 1];
  if (gv != 0) return;
  const int np = (h1 - h0) >> 1;
  // the finishing wave fetches its diagonal factor and right-hand side up front: they do not depend on the sweep,
  // so their latency overlaps the block pairs instead of following the LDS exchange
  const int slot = slice * 64 + lane;
  int node = -1;
  double u[9], ri0 = 0.0, ri1 = 0.0, ri2 = 0.0, zo0 = 0.0, zo1 = 0.0, zo2 = 0.0;
  if (w == 0) {
    node = slot_node ? slot_node[slot] : slot;
    const size_t base = (size_t)slice * 576 + lane;
#pragma unroll
    for (int e = 0; e < 9; e++) u[e] = alu[base + (size_t)e * 64];
    if (node >= 0) {
      if (FWD || partials) { ri0 = r[(size_t)3 * node]; ri1 = r[(size_t)3 * node + 1]; ri2 = r[(size_t)3 * node + 2]; }
      if (!FWD) { zo0 = zs[(size_t)3 * slot]; zo1 = zs[(size_t)3 * slot + 1]; zo2 = zs[(size_t)3 * slot + 2]; }
    }
  }
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  const double2 *vbase = (const double2 *)(val2 + (size_t)h0 * 576) + lane;
  const int2 *cbase = (const int2 *)(col2 + (size_t)h0 * 64) + lane;
  int i = w;
  for (; i + WPS < np; i += 2 * WPS) {  // two of this wave's pairs in flight together
    const double2 *va = vbase + (size_t)i * 576, *vb = vbase + (size_t)(i + WPS) * 576;
    const int2 ca = ld_stream(cbase + (size_t)i * 64), cb = ld_stream(cbase + (size_t)(i + WPS) * 64);
    double2 a[9], b[9];
#pragma unroll
    for (int e = 0; e < 9; e++) a[e] = ld_stream(va + e * 64);
#pragma unroll
    for (int e = 0; e < 9; e++) b[e] = ld_stream(vb + e * 64);
    const double *xa = zs + (size_t)3 * ca.x, *xb = zs + (size_t)3 * ca.y, *xc = zs + (size_t)3 * cb.x, *xd = zs + (size_t)3 * cb.y;
    const double xva[6] = {xa[0], xa[1], xa[2], xb[0], xb[1], xb[2]};
    const double xvb[6] = {xc[0], xc[1], xc[2], xd[0], xd[1], xd[2]};
    bell_pair_fma(a, xva, s0, s1, s2);
    bell_pair_fma(b, xvb, s0, s1, s2);
  }
  if (i < np) {
    const double2 *va = vbase + (size_t)i * 576;
    const int2 ca = ld_stream(cbase + (size_t)i * 64);
    double2 a[9];
#pragma unroll
    for (int e = 0; e < 9; e++) a[e] = ld_stream(va + e * 64);
    const double *xa = zs + (size_t)3 * ca.x, *xb = zs + (size_t)3 * ca.y;
    const double xva[6] = {xa[0], xa[1], xa[2], xb[0], xb[1], xb[2]};
    bell_pair_fma(a, xva, s0, s1, s2);
  }
  if (((h1 - h0) & 1) && w == (np % WPS)) {  // an odd last block of the slice is stored alone
    const double *vt = val2 + (size_t)(h0 + 2 * np) * 576 + lane;
    const int cc = ld_stream(col2 + (size_t)(h0 + 2 * np) * 64 + lane);
    double a[9];
#pragma unroll
    for (int e = 0; e < 9; e++) a[e] = ld_stream(vt + e * 64);
    const double *xa = zs + (size_t)3 * cc;
    const double x[3] = {xa[0], xa[1], xa[2]};
    bell_single_fma(a, x, s0, s1, s2);
  }
  part[w][0][lane] = s0; part[w][1][lane] = s1; part[w][2][lane] = s2;
  __syncthreads();
  double d[1] = {0.0};
  if (w == 0) {
    s0 = part[0][0][lane]; s1 = part[0][1][lane]; s2 = part[0][2][lane];
#pragma unroll
    for (int k = 1; k < WPS; k++) { s0 += part[k][0][lane]; s1 += part[k][1][lane]; s2 += part[k][2][lane]; }
    if (node >= 0) {
      double *zi = zs + (size_t)3 * slot;
      if (FWD) {
        double x1 = ri0 - s0, x2 = ri1 - s1, x3 = ri2 - s2;
        lusolve33_dev(u, x1, x2, x3);
        zi[0] = x1; zi[1] = x2; zi[2] = x3;
      } else {
        lusolve33_dev(u, s0, s1, s2);
        const double x1 = zo0 - s0, x2 = zo1 - s1, x3 = zo2 - s2;
        zi[0] = x1; zi[1] = x2; zi[2] = x3;
        if (z) {
          double *zn = z + (size_t)3 * node;
          zn[0] = x1; zn[1] = x2; zn[2] = x3;
        }
        if (partials) d[0] = ri0 * x1 + ri1 * x2 + ri2 * x3;
      }
    }
  }
  if (!FWD && partials) block_sum_store<1, 64 * WPS>(d, partials, 0, blockIdx.x);
}

// ------------------------------------------------------------------------
// Eisenstat's form of CG + multicolour SSOR (opt-in; fx_context::eisenstat).  Per iteration, with ph = (D~+U) p carried
// instead of p:   ph = D~ t + beta ph                      (k_cg_update_p on (dt, ph))
//                 p  = (D~+U)^-1 ph                        (backward sweep = k_ssor_color<FWD> on the U layout, colours descending)
//                 v  = (D~+L)^-1 (ph + (D - 2D~) p),  w = p + v = (D~+L)^-1 A p,  q = ph + L p + (D - D~) p = A p
//                                                           (k_eis_forward: ONE pass over the L layout, two gathers per block)
//                 alpha = rho / (ph . w)                    (p.q = p.(D~+L) w = ((D~+U) p).w for a symmetric matrix)
//                 x += alpha p; r -= alpha q; t -= alpha w; dt = D~ t; rho' = t.dt  (= r.M^-1 r)      (k_eis_update)
// L / U = strictly lower / upper part in the colour ordering, halo columns dropped (single rank: none exist).
// ------------------------------------------------------------------------
// Row loop with TWO gathered vectors per block (v and p share the matrix stream): 2-deep software pipeline as bell_row_sweep.
__device__ __forceinline__ void bell_row_sweep_dual(int h0, int h1, const double *__restrict__ val, const int *__restrict__ col,
                                                    int lane, const double *__restrict__ xv, const double *__restrict__ xp,
                                                    double (&sv)[3], double (&sp)[3]) {
  const int np = (h1 - h0) >> 1;
  if ((h1 - h0) & 1) {  // odd last block of the slice, stored alone
    const double *vt = val + (size_t)(h0 + 2 * np) * 576 + lane;
    const int cc = ld_stream(col + (size_t)(h0 + 2 * np) * 64 + lane);
    double a[9];
#pragma unroll
    for (int e = 0; e < 9; e++) a[e] = ld_stream(vt + e * 64);
    const double av[3] = {xv[(size_t)3 * cc], xv[(size_t)3 * cc + 1], xv[(size_t)3 * cc + 2]};
    const double ap[3] = {xp[(size_t)3 * cc], xp[(size_t)3 * cc + 1], xp[(size_t)3 * cc + 2]};
    bell_single_fma(a, av, sv[0], sv[1], sv[2]);
    bell_single_fma(a, ap, sp[0], sp[1], sp[2]);
  }
  if (np <= 0) return;
  const double2 *v = (const double2 *)(val + (size_t)h0 * 576) + lane;
  const int2 *c = (const int2 *)(col + (size_t)h0 * 64) + lane;
  int2 cc1 = (np > 1) ? ld_stream(c + 64) : ld_stream(c);
  double2 a[9];
  double gv[6], gp[6];
  {
    const int2 cc0 = ld_stream(c);
#pragma unroll
    for (int e = 0; e < 9; e++) a[e] = ld_stream(v + e * 64);
#pragma unroll
    for (int k = 0; k < 3; k++) {
      gv[k] = xv[(size_t)3 * cc0.x + k]; gv[3 + k] = xv[(size_t)3 * cc0.y + k];
      gp[k] = xp[(size_t)3 * cc0.x + k]; gp[3 + k] = xp[(size_t)3 * cc0.y + k];
    }
  }
  for (int i = 0; i < np; i++) {
    const bool more = (i + 1 < np);
    double2 an[9];
    double nv[6] = {0, 0, 0, 0, 0, 0}, npv[6] = {0, 0, 0, 0, 0, 0};
    int2 cc2 = cc1;
    if (more) {
      v += 576; c += 64;
#pragma unroll
      for (int e = 0; e < 9; e++) an[e] = ld_stream(v + e * 64);
#pragma unroll
      for (int k = 0; k < 3; k++) {
        nv[k] = xv[(size_t)3 * cc1.x + k]; nv[3 + k] = xv[(size_t)3 * cc1.y + k];
        npv[k] = xp[(size_t)3 * cc1.x + k]; npv[3 + k] = xp[(size_t)3 * cc1.y + k];
      }
      if (i + 2 < np) cc2 = ld_stream(c + 64);
    }
    bell_pair_fma(a, gv, sv[0], sv[1], sv[2]);
    bell_pair_fma(a, gp, sp[0], sp[1], sp[2]);
    if (more) {
#pragma unroll
      for (int e = 0; e < 9; e++) a[e] = an[e];
#pragma unroll
      for (int k = 0; k < 6; k++) { gv[k] = nv[k]; gp[k] = npv[k]; }
      cc1 = cc2;
    }
  }
}

// the row's own part of the forward Eisenstat sweep, from the two block sums sv = (L v)_i, sp = (L p)_i; returns ph_i . w_i
// esc = (SIGMA_DIAG - 1) / SIGMA_DIAG: (D~ - D) p = esc * diag(D~) p (SIGMA_DIAG scales the three scalar diagonal entries only).
__device__ __forceinline__ double eis_forward_finish(int slice, int lane, const double (&sv)[3], const double (&sp)[3],
                                                     const double *__restrict__ alu, double esc,
                                                     const double *__restrict__ ph, const double *__restrict__ p,
                                                     double *__restrict__ v, double *__restrict__ w, double *__restrict__ q,
                                                     const double *__restrict__ hp) {
  const int slot = slice * 64 + lane;
  double u[9];
  const size_t base = (size_t)slice * 576 + lane;
#pragma unroll
  for (int e = 0; e < 9; e++) u[e] = alu[base + (size_t)e * 64];
  const double p0 = p[(size_t)3 * slot], p1 = p[(size_t)3 * slot + 1], p2 = p[(size_t)3 * slot + 2];
  double h0v = ph[(size_t)3 * slot], h1v = ph[(size_t)3 * slot + 1], h2v = ph[(size_t)3 * slot + 2];
  const double pd = h0v, pe = h1v, pf = h2v;  // ph itself (the dot product p.q = ph.w uses it without the halo term)
  if (hp) {  // subdomain: A = (D~+L) + (D~+U) + (D - 2D~) + H with H the halo columns; H p enters g and q alike
    h0v += hp[(size_t)3 * slot]; h1v += hp[(size_t)3 * slot + 1]; h2v += hp[(size_t)3 * slot + 2];
  }
  double Tp0, Tp1, Tp2, d0, d1, d2;   // D~ p and diag(D~) from the factors
  lumul33_dev(u, p0, p1, p2, Tp0, Tp1, Tp2, d0, d1, d2);
  const double e0 = esc * d0 * p0, e1 = esc * d1 * p1, e2 = esc * d2 * p2;  // (D~ - D) p
  // g = ph + H p + (D - 2 D~) p = ph + H p - D~ p - (D~ - D) p ;  v = D~^-1 (g - L v)
  double x1 = h0v - Tp0 - e0 - sv[0], x2 = h1v - Tp1 - e1 - sv[1], x3 = h2v - Tp2 - e2 - sv[2];
  lusolve33_dev(u, x1, x2, x3);
  v[(size_t)3 * slot] = x1; v[(size_t)3 * slot + 1] = x2; v[(size_t)3 * slot + 2] = x3;
  const double w0 = p0 + x1, w1 = p1 + x2, w2 = p2 + x3;
  w[(size_t)3 * slot] = w0; w[(size_t)3 * slot + 1] = w1; w[(size_t)3 * slot + 2] = w2;
  // q = A p = ph + H p + L p + (D - D~) p
  q[(size_t)3 * slot] = h0v + sp[0] - e0; q[(size_t)3 * slot + 1] = h1v + sp[1] - e1; q[(size_t)3 * slot + 2] = h2v + sp[2] - e2;
  return pd * w0 + pe * w1 + pf * w2;
}

template <int BS>
__global__ __launch_bounds__(BS) void k_eis_forward(int32_t slice0, int32_t slice1, const int32_t *__restrict__ pair_ptr,
                                                    const double *__restrict__ val2, const int *__restrict__ col2,
                                                    const double *__restrict__ alu, double esc,
                                                    const double *__restrict__ ph, const double *__restrict__ p,
                                                    double *__restrict__ v, double *__restrict__ w, double *__restrict__ q,
                                                    double *__restrict__ partials, int32_t part0, const int32_t *__restrict__ gate,
                                                    const double *__restrict__ hp) {
  if (gate && *gate != 0) return;
  const int vb = xcd_block(blockIdx.x, gridDim.x);
  const int slice = slice0 + vb * (BS / 64) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  double d[1] = {0.0};
  if (slice < slice1) {
    double sv[3] = {0.0, 0.0, 0.0}, sp[3] = {0.0, 0.0, 0.0};
    bell_row_sweep_dual(pair_ptr[slice], pair_ptr[slice + 1], val2, col2, lane, v, p, sv, sp);
    d[0] = eis_forward_finish(slice, lane, sv, sp, alu, esc, ph, p, v, w, q, hp);
  }
  block_sum_store<1, BS>(d, partials, 0, part0 + vb);
}

// the same sweep for latency-bound colours: WPS waves share a slice's block pairs (as k_ssor_color_split)
template <int WPS>
__global__ __launch_bounds__(64 * WPS) void k_eis_forward_split(int32_t slice0, int32_t slice1, const int32_t *__restrict__ pair_ptr,
                                                                const double *__restrict__ val2, const int *__restrict__ col2,
                                                                const double *__restrict__ alu,
                                                                double esc, const double *__restrict__ ph, const double *__restrict__ p,
                                                                double *__restrict__ v, double *__restrict__ w, double *__restrict__ q,
                                                                double *__restrict__ partials, int32_t part0,
                                                                const int32_t *__restrict__ gate, const double *__restrict__ hp) {
  // grid <= slices of the colour: workgroup b takes slices b, b + grid, ... (FX_EIS_GRID; one partial per workgroup); a second
  // barrier per slice only when the workgroup goes on to another one (the LDS exchange is single-buffered: occupancy)
  __shared__ double part[WPS][6][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (gate && *gate != 0) return;
  double dsum = 0.0;
  for (int slice = slice0 + blockIdx.x; slice < slice1; slice += gridDim.x) {
    const int h0 = pair_ptr[slice], h1 = pair_ptr[slice + 1];
    const int np = (h1 - h0) >> 1;
    double sv[3] = {0.0, 0.0, 0.0}, sp[3] = {0.0, 0.0, 0.0};
    const double2 *vbase = (const double2 *)(val2 + (size_t)h0 * 576) + lane;
    const int2 *cbase = (const int2 *)(col2 + (size_t)h0 * 64) + lane;
    for (int i = wv; i < np; i += WPS) {
      const int2 cc = ld_stream(cbase + (size_t)i * 64);
      double2 a[9];
#pragma unroll
      for (int e = 0; e < 9; e++) a[e] = ld_stream(vbase + (size_t)i * 576 + e * 64);
      double gv[6], gp[6];
#pragma unroll
      for (int k = 0; k < 3; k++) {
        gv[k] = v[(size_t)3 * cc.x + k]; gv[3 + k] = v[(size_t)3 * cc.y + k];
        gp[k] = p[(size_t)3 * cc.x + k]; gp[3 + k] = p[(size_t)3 * cc.y + k];
      }
      bell_pair_fma(a, gv, sv[0], sv[1], sv[2]);
      bell_pair_fma(a, gp, sp[0], sp[1], sp[2]);
    }
    if (((h1 - h0) & 1) && wv == (np % WPS)) {
      const double *vt = val2 + (size_t)(h0 + 2 * np) * 576 + lane;
      const int cc = ld_stream(col2 + (size_t)(h0 + 2 * np) * 64 + lane);
      double a[9];
#pragma unroll
      for (int e = 0; e < 9; e++) a[e] = ld_stream(vt + e * 64);
      const double av[3] = {v[(size_t)3 * cc], v[(size_t)3 * cc + 1], v[(size_t)3 * cc + 2]};
      const double ap[3] = {p[(size_t)3 * cc], p[(size_t)3 * cc + 1], p[(size_t)3 * cc + 2]};
      bell_single_fma(a, av, sv[0], sv[1], sv[2]);
      bell_single_fma(a, ap, sp[0], sp[1], sp[2]);
    }
#pragma unroll
    for (int k = 0; k < 3; k++) { part[wv][k][lane] = sv[k]; part[wv][3 + k][lane] = sp[k]; }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
      for (int k = 0; k < 3; k++) { sv[k] = part[0][k][lane]; sp[k] = part[0][3 + k][lane]; }
#pragma unroll
      for (int j = 1; j < WPS; j++)
#pragma unroll
        for (int k = 0; k < 3; k++) { sv[k] += part[j][k][lane]; sp[k] += part[j][3 + k][lane]; }
      dsum += eis_forward_finish(slice, lane, sv, sp, alu, esc, ph, p, v, w, q, hp);
    }
    if (slice + (int)gridDim.x < slice1) __syncthreads();
  }
  if (wv == 0) {
    dsum = wave_sum(dsum);
    if (lane == 0) partials[part0 + blockIdx.x] = dsum;
  }
}

// Backward half of the Eisenstat iteration with the direction update fused in: ph_i = dt_i + beta ph_i (hecmw_solver_CG.f90:188-197
// in the transformed variables; beta = 0 on the first iteration), then p_i = D~_i^-1 (ph_i - sum_{j in U(i)} U_ij p_j).  Replaces
// k_cg_update_p + k_ssor_color<true> on the U layout: ph is read and written by the row's own lane, one pass and one launch less.
__device__ __forceinline__ void eis_backward_finish(int slice, int lane, double s0, double s1, double s2, const double *__restrict__ alu,
                                                    const KrylovState *__restrict__ st, const double *__restrict__ dt,
                                                    double *__restrict__ ph, double *__restrict__ p) {
  const size_t o = (size_t)3 * (slice * 64 + lane);
  double u[9];
  const size_t base = (size_t)slice * 576 + lane;
#pragma unroll
  for (int e = 0; e < 9; e++) u[e] = alu[base + (size_t)e * 64];
  double h0 = dt[o], h1 = dt[o + 1], h2 = dt[o + 2];
  if (st->iter != 1) {
    const double beta = st->beta;
    h0 = h0 + beta * ph[o]; h1 = h1 + beta * ph[o + 1]; h2 = h2 + beta * ph[o + 2];
  }
  ph[o] = h0; ph[o + 1] = h1; ph[o + 2] = h2;
  double x1 = h0 - s0, x2 = h1 - s1, x3 = h2 - s2;
  lusolve33_dev(u, x1, x2, x3);
  p[o] = x1; p[o + 1] = x2; p[o + 2] = x3;
}

template <int BS>
__global__ __launch_bounds__(BS) void k_eis_backward(int32_t slice0, int32_t slice1, const int32_t *__restrict__ pair_ptr,
                                                     const double *__restrict__ val2, const int *__restrict__ col2,
                                                     const double *__restrict__ alu, const KrylovState *__restrict__ st,
                                                     const double *__restrict__ dt, double *__restrict__ ph, double *__restrict__ p,
                                                     const int32_t *__restrict__ gate) {
  if (gate && *gate != 0) return;
  const int vb = xcd_block(blockIdx.x, gridDim.x);
  const int slice = slice0 + vb * (BS / 64) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (slice >= slice1) return;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  bell_row_sweep<true>(pair_ptr[slice], pair_ptr[slice + 1], val2, col2, lane, p, s0, s1, s2);
  eis_backward_finish(slice, lane, s0, s1, s2, alu, st, dt, ph, p);
}

template <int WPS>
__global__ __launch_bounds__(64 * WPS) void k_eis_backward_split(int32_t slice0, int32_t slice1, const int32_t *__restrict__ pair_ptr,
                                                                 const double *__restrict__ val2, const int *__restrict__ col2,
                                                                 const double *__restrict__ alu, const KrylovState *__restrict__ st,
                                                                 const double *__restrict__ dt, double *__restrict__ ph,
                                                                 double *__restrict__ p, const int32_t *__restrict__ gate) {
  __shared__ double part[WPS][3][64];  // as k_eis_forward_split: workgroup b takes slices b, b + grid, ...
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (gate && *gate != 0) return;
  for (int slice = slice0 + blockIdx.x; slice < slice1; slice += gridDim.x) {
    const int h0 = pair_ptr[slice], h1 = pair_ptr[slice + 1];
    const int np = (h1 - h0) >> 1;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    // the finishing wave's own operands (diagonal factor, D~t, ph) do not depend on the sweep: in flight before its block pairs, so that
    // their latency overlaps the pairs' instead of following the LDS exchange (round 4, same-process pairs: 1.788 -> 1.774 ms per
    // iteration; the same change in the forward kernel, which carries six more gathered entries per pair, lost 2 %: 106 -> 130 VGPRs)
    double fu[9], fd[3], fp[3];
    if (w == 0) {
      const size_t o = (size_t)3 * (slice * 64 + lane), base = (size_t)slice * 576 + lane;
#pragma unroll
      for (int e = 0; e < 9; e++) fu[e] = alu[base + (size_t)e * 64];
#pragma unroll
      for (int k = 0; k < 3; k++) { fd[k] = dt[o + k]; fp[k] = ph[o + k]; }
    }
    const double2 *vbase = (const double2 *)(val2 + (size_t)h0 * 576) + lane;
    const int2 *cbase = (const int2 *)(col2 + (size_t)h0 * 64) + lane;
    for (int i = w; i < np; i += WPS) {
      const int2 cc = ld_stream(cbase + (size_t)i * 64);
      double2 a[9];
#pragma unroll
      for (int e = 0; e < 9; e++) a[e] = ld_stream(vbase + (size_t)i * 576 + e * 64);
      const double *xa = p + (size_t)3 * cc.x, *xb = p + (size_t)3 * cc.y;
      const double xv[6] = {xa[0], xa[1], xa[2], xb[0], xb[1], xb[2]};
      bell_pair_fma(a, xv, s0, s1, s2);
    }
    if (((h1 - h0) & 1) && w == (np % WPS)) {
      const double *vt = val2 + (size_t)(h0 + 2 * np) * 576 + lane;
      const int cc = ld_stream(col2 + (size_t)(h0 + 2 * np) * 64 + lane);
      double a[9];
#pragma unroll
      for (int e = 0; e < 9; e++) a[e] = ld_stream(vt + e * 64);
      const double x[3] = {p[(size_t)3 * cc], p[(size_t)3 * cc + 1], p[(size_t)3 * cc + 2]};
      bell_single_fma(a, x, s0, s1, s2);
    }
    part[w][0][lane] = s0; part[w][1][lane] = s1; part[w][2][lane] = s2;
    __syncthreads();
    if (w == 0) {
      s0 = part[0][0][lane]; s1 = part[0][1][lane]; s2 = part[0][2][lane];
#pragma unroll
      for (int k = 1; k < WPS; k++) { s0 += part[k][0][lane]; s1 += part[k][1][lane]; s2 += part[k][2][lane]; }
      {
        const size_t o = (size_t)3 * (slice * 64 + lane);
        double g0 = fd[0], g1 = fd[1], g2 = fd[2];
        if (st->iter != 1) {
          const double beta = st->beta;
          g0 = g0 + beta * fp[0]; g1 = g1 + beta * fp[1]; g2 = g2 + beta * fp[2];
        }
        ph[o] = g0; ph[o + 1] = g1; ph[o + 2] = g2;
        double x1 = g0 - s0, x2 = g1 - s1, x3 = g2 - s2;
        lusolve33_dev(fu, x1, x2, x3);
        p[o] = x1; p[o + 1] = x2; p[o + 2] = x3;
      }
    }
    if (slice + (int)gridDim.x < slice1) __syncthreads();
  }
}

// hp = H p: the halo-column blocks of every row against the freshly exchanged halo part of p (rows without halo blocks get 0)
__global__ __launch_bounds__(256) void k_eis_halo(int32_t nslices, const int32_t *__restrict__ pair_ptr, const double *__restrict__ val2,
                                                  const int *__restrict__ col2, const double *__restrict__ p, double *__restrict__ hp,
                                                  const int32_t *__restrict__ gate) {
  if (gate && *gate != 0) return;
  const int slice = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (slice >= nslices) return;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  bell_row_sweep<false>(pair_ptr[slice], pair_ptr[slice + 1], val2, col2, lane, p, s0, s1, s2);
  const size_t o = (size_t)3 * (slice * 64 + lane);
  hp[o] = s0; hp[o + 1] = s1; hp[o + 2] = s2;
}

// dt = D~ t (from the LU factors: lumul33_dev) and the partial of rho = t.dt; MODE 1 also x += alpha p, r -= alpha q (+ partial ||r||^2), t -= alpha w first;
// MODE 2: x += alpha p only (the iterations that recompute r = b - A x).  One thread per slot, blocks in [slice][e][lane] layout.
template <int MODE>
__global__ __launch_bounds__(FX_BLOCK) void k_eis_update(int32_t nslots, const KrylovState *__restrict__ st,
                                                         const double *__restrict__ alu, const double *__restrict__ p,
                                                         const double *__restrict__ q, const double *__restrict__ w,
                                                         double *__restrict__ x, double *__restrict__ r, double *__restrict__ t,
                                                         double *__restrict__ dt, double *__restrict__ part_rr,
                                                         double *__restrict__ part_rho, const int32_t *__restrict__ gate) {
  if (gate && *gate != 0) return;
  const int slot = blockIdx.x * FX_BLOCK + threadIdx.x;
  double d[2] = {0.0, 0.0};
  if (slot < nslots) {
    const size_t o = (size_t)3 * slot;
    if (MODE == 2) {
      const double alpha = st->alpha;
      x[o] += alpha * p[o]; x[o + 1] += alpha * p[o + 1]; x[o + 2] += alpha * p[o + 2];
    } else {
      double t0 = t[o], t1 = t[o + 1], t2 = t[o + 2];
      if (MODE == 1) {
        const double alpha = st->alpha;
        x[o] += alpha * p[o]; x[o + 1] += alpha * p[o + 1]; x[o + 2] += alpha * p[o + 2];
        const double r0 = r[o] - alpha * q[o], r1 = r[o + 1] - alpha * q[o + 1], r2 = r[o + 2] - alpha * q[o + 2];
        r[o] = r0; r[o + 1] = r1; r[o + 2] = r2;
        d[0] = r0 * r0 + r1 * r1 + r2 * r2;
        t0 -= alpha * w[o]; t1 -= alpha * w[o + 1]; t2 -= alpha * w[o + 2];
        t[o] = t0; t[o + 1] = t1; t[o + 2] = t2;
      }
      double u[9];
      const size_t base = (size_t)(slot >> 6) * 576 + (slot & 63);
#pragma unroll
      for (int e = 0; e < 9; e++) u[e] = alu[base + (size_t)e * 64];
      double y0, y1, y2, dd0, dd1, dd2;
      lumul33_dev(u, t0, t1, t2, y0, y1, y2, dd0, dd1, dd2);   // D~ t from the factors
      dt[o] = y0; dt[o + 1] = y1; dt[o + 2] = y2;
      d[1] = t0 * y0 + t1 * y1 + t2 * y2;
    }
  }
  if (MODE == 2) return;
  __shared__ double sm[2][FX_BLOCK / 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const double s0 = wave_sum(d[0]), s1 = wave_sum(d[1]);
  if (lane == 0) { sm[0][wv] = s0; sm[1][wv] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int k = 0; k < FX_BLOCK / 64; k++) { a += sm[0][k]; b += sm[1][k]; }
    if (MODE == 1) part_rr[blockIdx.x] = a;
    part_rho[blockIdx.x] = b;
  }
}

// ------------------------------------------------------------------------
// Dataflow triangular sweeps: ONE persistent launch per preconditioner apply instead of one launch per colour / per
// ILU(0) dependency level (1,044 levels x 2 half sweeps at 10M DOF; each level is ~50 slices, i.e. a few microseconds
// of dependent latency and almost no bandwidth).  There is no grid barrier either: the DATA is the flag.  Both sweep
// vectors are filled with a sentinel bit pattern before the launch; a row publishes its three entries with 8-byte
// agent-scope (write-through, sc1) stores -- one naturally aligned store each, so never torn -- and a consumer re-reads
// the entries it gathers with agent-scope loads until none is the sentinel (MI355X_MICROARCH.md, inter-workgroup
// visibility: the self-tagged 8-byte granule hand-off, 0.8-1.5 us per hop; a kernel boundary costs 1.5-1.9 us plus the
// dependent index -> block -> gather chain after it).  The matrix stream of a slice (blocks, column ids, diagonal factor,
// right-hand side) does not depend on the sweep and is in flight before the slice starts to poll, so only the gather of
// z, the 3x3 substitution and the store sit on the critical path.
//   forward  (ascending slices):  zf_i = D~_i^-1 (r_i - sum_{j in L(i)} L_ij zf_j)
//   backward (descending slices): zb_i = zf_i - D~_i^-1 sum_{j in U(i)} U_ij zb_j ;  z[node_i] = zb_i (+ partial of r.z)
// Same operands in the same order as k_ssor_color_split with the same WPS: results are bit-identical to it.
// Progress: workgroup w owns slices w, w + G, ...; it walks them upwards in the forward sweep and downwards in the
// backward sweep, and a slice only waits for slices earlier in its sweep's order, so the first unfinished slice of a
// sweep can always run -- provided all G workgroups are resident (G <= CUs here).
// Every spin is bounded by wall-clock time; a timeout raises *err (the host turns it into a runtime failure).
// ------------------------------------------------------------------------
#define FX_DF_SENTINEL (-1LL)          // 0xFFFFFFFFFFFFFFFF: a NaN pattern no arithmetic instruction produces
#define FX_DF_TIMEOUT_TICKS 200000000ull  // 2 s of the 100 MHz constant clock

__device__ __forceinline__ double df_load(const double *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void df_store(double *p, double v) {
  if (__double_as_longlong(v) == FX_DF_SENTINEL) v = __longlong_as_double(0x7FF8000000000000LL);  // keep the tag unique
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Tag fill of the sweep vectors before a launch, with the same agent-scope write-through stores the hand-off itself uses
// (a plain memset's lines can survive, stale, in another XCD's L2: per-XCD L2s are only kept coherent for sc1 traffic).
typedef unsigned int fx_u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_df_fill(int64_t n16, fx_u4 *__restrict__ a, fx_u4 *__restrict__ b) {
  const fx_u4 tag = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(a + i), "v"(tag) : "memory");
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(b + i), "v"(tag) : "memory");
  }
}

// Gather the two vector entries (3 doubles each) of one block pair, waiting until their producers have published them.
// A padding block points at the lane's own slot (value 0): no producer, contributes 0.
// POLL selects how a wave waits (A/B-measured in one process, scripts/experiments/ab_dataflow3.sh):
//   0  every pass re-reads all 3 * NB entries;  1  later passes re-read only the entries still unpublished.
// Layout of the sweep vectors.  SOA = false: entry k of slot s at 3 s + k (the Krylov vectors' layout: needed when the backward
// sweep writes the Krylov vector itself).  SOA = true (private sweep vectors: ILU(0), natural-order SSOR): [slice][k][lane] -- a
// wave publishes a component of its 64 rows as ONE 512-byte segment (four whole 128-byte lines per store instruction instead of 64
// scattered 8-byte fabric writes), and a gather of 64 consecutive slots reads one such segment per component.
template <bool SOA> __device__ __forceinline__ size_t df_ix(int slot) { return SOA ? ((size_t)(slot >> 6) * 192 + (slot & 63)) : (size_t)3 * slot; }
#define DF_ST(SOA) ((SOA) ? 64 : 1)

template <int NB, int POLL, bool SOA>
__device__ __forceinline__ void df_gather(const double *__restrict__ zs, const int (&col)[NB], int self, double (&x)[3 * NB],
                                          int32_t *__restrict__ err, bool &dead, int nsleep) {
  bool miss[3 * NB];
  bool any = false;
#pragma unroll
  for (int b = 0; b < NB; b++) {
    const double *xa = zs + df_ix<SOA>(col[b]);
    const bool need = col[b] != self;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      x[3 * b + k] = df_load(xa + k * DF_ST(SOA));
      miss[3 * b + k] = need && __double_as_longlong(x[3 * b + k]) == FX_DF_SENTINEL;
      any |= miss[3 * b + k];
    }
    if (!need) { x[3 * b] = 0.0; x[3 * b + 1] = 0.0; x[3 * b + 2] = 0.0; }
  }
  if (!__any(any) || dead) return;  // a wave that has given up takes what is there: the launch drains without waiting
  unsigned long long t0 = 0;
  for (unsigned spins = 1;; spins++) {
    for (int q = 0; q < nsleep; q++) __builtin_amdgcn_s_sleep(1);  // 64 clocks each
    any = false;
    // all re-reads of a pass in flight together -- an entry that is not re-read loads slot 0 instead (3.140 -> 3.115 ms against the row's own slot), so that there is no
    // branch (and no wait) between the loads: one round trip per pass, not one per missing entry
    double v[3 * NB];
#pragma unroll
    for (int e = 0; e < 3 * NB; e++) {
      const bool need = col[e / 3] != self;
      const bool rd = POLL == 0 ? need : miss[e];
      v[e] = df_load(zs + df_ix<SOA>(rd ? col[e / 3] : 0) + (e % 3) * DF_ST(SOA));  // not re-read: slot 0, one address for all such lanes
    }
#pragma unroll
    for (int e = 0; e < 3 * NB; e++) {
      const bool need = col[e / 3] != self;
      if (POLL == 0 ? need : miss[e]) {
        x[e] = v[e];
        miss[e] = __double_as_longlong(v[e]) == FX_DF_SENTINEL;
        any |= miss[e];
      }
    }
    if (!__any(any)) return;
    if ((spins & 255u) == 0u) {  // bounded spin: give up after FX_DF_TIMEOUT_TICKS, or as soon as somebody else has
      const unsigned long long now = __builtin_amdgcn_s_memrealtime();
      if (t0 == 0) t0 = now;
      const int e = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (e != 0 || now - t0 > FX_DF_TIMEOUT_TICKS) {
        if (e == 0) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        dead = true;
        return;
      }
    }
  }
}

template <bool FWD, int WPS, int POLL, bool SOA>
__device__ __forceinline__ void df_slice(int slice, const int32_t *__restrict__ pair_ptr, const double *__restrict__ val2,
                                         const int *__restrict__ col2, const int32_t *__restrict__ slot_node,
                                         const double *__restrict__ alu, const double *__restrict__ r,
                                         double *__restrict__ zf, double *__restrict__ zb, double *__restrict__ z,
                                         double *__restrict__ partials, int32_t *__restrict__ err, double (*part)[3][64],
                                         bool &dead, int nsleep) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int h0 = pair_ptr[slice], h1 = pair_ptr[slice + 1];
  const int np = (h1 - h0) >> 1;
  const int slot = slice * 64 + lane;
  const double *zsrc = FWD ? zf : zb;  // the vector whose entries this sweep produces and gathers
  // the finishing wave's own operands: independent of the sweep, in flight before the first poll
  int node = -1;
  double u[9], ri0 = 0.0, ri1 = 0.0, ri2 = 0.0, zo0 = 0.0, zo1 = 0.0, zo2 = 0.0;
  if (w == 0) {
    node = slot_node ? slot_node[slot] : slot;
    const size_t base = (size_t)slice * 576 + lane;
#pragma unroll
    for (int e = 0; e < 9; e++) u[e] = alu[base + (size_t)e * 64];
    if (node >= 0) {
      if (FWD || partials) { ri0 = r[(size_t)3 * node]; ri1 = r[(size_t)3 * node + 1]; ri2 = r[(size_t)3 * node + 2]; }
      if (!FWD) {  // this row's forward value: written by this very thread earlier in the launch (a workgroup keeps its slices)
        const double *zq = zf + df_ix<SOA>(slot);
        zo0 = df_load(zq); zo1 = df_load(zq + DF_ST(SOA)); zo2 = df_load(zq + 2 * DF_ST(SOA));
      }
    }
  }
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  const double2 *vbase = (const double2 *)(val2 + (size_t)h0 * 576) + lane;
  const int2 *cbase = (const int2 *)(col2 + (size_t)h0 * 64) + lane;
  int i = w;
  for (; i + WPS < np; i += 2 * WPS) {  // two of this wave's pairs in flight together
    const double2 *va = vbase + (size_t)i * 576, *vb = vbase + (size_t)(i + WPS) * 576;
    const int2 ca = ld_stream(cbase + (size_t)i * 64), cb = ld_stream(cbase + (size_t)(i + WPS) * 64);
    double2 a[9], b[9];
#pragma unroll
    for (int e = 0; e < 9; e++) a[e] = ld_stream(va + e * 64);
#pragma unroll
    for (int e = 0; e < 9; e++) b[e] = ld_stream(vb + e * 64);
    const int cols[4] = {ca.x, ca.y, cb.x, cb.y};
    double x[12];
    df_gather<4, POLL, SOA>(zsrc, cols, slot, x, err, dead, nsleep);
    const double xva[6] = {x[0], x[1], x[2], x[3], x[4], x[5]};
    const double xvb[6] = {x[6], x[7], x[8], x[9], x[10], x[11]};
    bell_pair_fma(a, xva, s0, s1, s2);
    bell_pair_fma(b, xvb, s0, s1, s2);
  }
  if (i < np) {
    const double2 *va = vbase + (size_t)i * 576;
    const int2 ca = ld_stream(cbase + (size_t)i * 64);
    double2 a[9];
#pragma unroll
    for (int e = 0; e < 9; e++) a[e] = ld_stream(va + e * 64);
    const int cols[2] = {ca.x, ca.y};
    double x[6];
    df_gather<2, POLL, SOA>(zsrc, cols, slot, x, err, dead, nsleep);
    bell_pair_fma(a, x, s0, s1, s2);
  }
  if (((h1 - h0) & 1) && w == (np % WPS)) {  // an odd last block of the slice is stored alone
    const double *vt = val2 + (size_t)(h0 + 2 * np) * 576 + lane;
    const int cols[1] = {ld_stream(col2 + (size_t)(h0 + 2 * np) * 64 + lane)};
    double a[9], x[3];
#pragma unroll
    for (int e = 0; e < 9; e++) a[e] = ld_stream(vt + e * 64);
    df_gather<1, POLL, SOA>(zsrc, cols, slot, x, err, dead, nsleep);
    bell_single_fma(a, x, s0, s1, s2);
  }
  part[w][0][lane] = s0; part[w][1][lane] = s1; part[w][2][lane] = s2;
  __syncthreads();
  if (w == 0) {
    s0 = part[0][0][lane]; s1 = part[0][1][lane]; s2 = part[0][2][lane];
#pragma unroll
    for (int k = 1; k < WPS; k++) { s0 += part[k][0][lane]; s1 += part[k][1][lane]; s2 += part[k][2][lane]; }
    double d = 0.0;
    if (node >= 0) {
      if (FWD) {
        double x1 = ri0 - s0, x2 = ri1 - s1, x3 = ri2 - s2;
        lusolve33_dev(u, x1, x2, x3);
        double *zi = zf + df_ix<SOA>(slot);
        df_store(zi, x1); df_store(zi + DF_ST(SOA), x2); df_store(zi + 2 * DF_ST(SOA), x3);
      } else {
        lusolve33_dev(u, s0, s1, s2);
        const double x1 = zo0 - s0, x2 = zo1 - s1, x3 = zo2 - s2;
        double *zi = zb + df_ix<SOA>(slot);
        df_store(zi, x1); df_store(zi + DF_ST(SOA), x2); df_store(zi + 2 * DF_ST(SOA), x3);
        if (z) {
          double *zn = z + (size_t)3 * node;
          zn[0] = x1; zn[1] = x2; zn[2] = x3;
        }
        if (partials) d = ri0 * x1 + ri1 * x2 + ri2 * x3;
      }
    }
    if (!FWD && partials) {
      d = wave_sum(d);
      if (lane == 0) partials[slice] = d;
    }
  }
}

template <int WPS, int POLL, bool SOA>
__global__ __launch_bounds__(64 * WPS) void k_tri_dataflow(int32_t nslices, const int32_t *__restrict__ Lptr,
                                                           const double *__restrict__ Lval, const int *__restrict__ Lcol,
                                                           const int32_t *__restrict__ Uptr, const double *__restrict__ Uval,
                                                           const int *__restrict__ Ucol,
                                                           const int32_t *__restrict__ slot_node,
                                                           const double *__restrict__ alu, const double *__restrict__ r,
                                                           double *__restrict__ zf, double *__restrict__ zb,
                                                           double *__restrict__ z, double *__restrict__ partials,
                                                           const int32_t *__restrict__ gate, int32_t *__restrict__ err,
                                                           int nsleep, const int32_t *__restrict__ slice_level, int presleep) {
  if (gate && *gate != 0) return;
  if (nsleep < 0) {  // test hook (FX_DEBUG_DF_FAIL): behave like a launch whose bounded wait ran out at once -- nothing usable written, err raised
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  __shared__ double part[2][WPS][3][64];  // double-buffered: one workgroup barrier per slice
  int buf = 0;
  bool dead = false;  // per wave: its bounded wait ran out (or another wave's did); it then stops waiting, never stops running
  // A workgroup that has just finished a slice of level l0 and goes on to one of level l1 is l1 - l0 - 1 levels ahead of the frontier:
  // it sleeps `presleep` x 0.1 us per level of that lead before it starts to poll (FX_DF_PRESLEEP; its polls would only compete with
  // the frontier's hand-offs).
  auto lead_sleep = [&](int from, int to) {
    if (!slice_level || presleep <= 0) return;
    const int lead = abs(slice_level[to] - slice_level[from]) - 1;
    for (int q = 0; q < lead * presleep; q++) __builtin_amdgcn_s_sleep(4);  // 256 clocks = 0.1 us each
  };
  for (int slice = blockIdx.x; slice < nslices; slice += gridDim.x, buf ^= 1) {
    if (slice != (int)blockIdx.x) lead_sleep(slice - (int)gridDim.x, slice);
    df_slice<true, WPS, POLL, SOA>(slice, Lptr, Lval, Lcol, slot_node, alu, r, zf, zb, z, partials, err, part[buf], dead, nsleep);
  }
  // backward: the SAME slices, last first (a row's forward value is then its own thread's earlier store)
  const int mine = nslices > (int)blockIdx.x ? (nslices - 1 - (int)blockIdx.x) / (int)gridDim.x : -1;
  for (int slice = (int)blockIdx.x + mine * (int)gridDim.x; mine >= 0 && slice >= 0; slice -= gridDim.x, buf ^= 1) {
    if (slice + (int)gridDim.x < nslices) lead_sleep(slice + (int)gridDim.x, slice);
    df_slice<false, WPS, POLL, SOA>(slice, Uptr, Uval, Ucol, slot_node, alu, r, zf, zb, z, partials, err, part[buf], dead, nsleep);
  }
}

// ------------------------------------------------------------------------

// LU of the (sigma-scaled) diagonal blocks in the natural [9*i] layout (ILU1a33 :1493-1528).
// In the reference the Schur update of the diagonal block (:309-319) never executes (row i is
// not in its own IW1/IW2 lists), so Dlu0 depends on D only.
__global__ void k_dlu_natural(int32_t n, const double *__restrict__ D, double sigma_diag, double *__restrict__ Dlu) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double a[9];
#pragma unroll
  for (int e = 0; e < 9; e++) a[e] = D[(size_t)9 * i + e];
  a[0] *= sigma_diag; a[4] *= sigma_diag; a[8] *= sigma_diag;
  lu33_dev(a);
#pragma unroll
  for (int e = 0; e < 9; e++) Dlu[(size_t)9 * i + e] = a[e];
}

__device__ __forceinline__ int32_t item_find(const int32_t *item, int32_t lo, int32_t hi, int32_t val) {
  while (lo < hi) {
    const int32_t mid = (lo + hi) >> 1;
    const int32_t v = item[mid];
    if (v < val) lo = mid + 1;
    else if (v > val) hi = mid;
    else return mid;
  }
  return -1;
}

// One level of FORM_ILU0_33 (:254-345): A_ij -= A_ik * Dk^-1 * A_kj for k in L(i), j in U(k) and in
// the pattern of row i (j /= i), k ascending as in the reference.  ILU1b33 (:1538-1596) inlined.
__global__ void k_ilu0_factor_level(int32_t slot0, int32_t slot1, const int32_t *__restrict__ slot_node, int32_t N,
                                    const int32_t *__restrict__ indexL, const int32_t *__restrict__ itemL,
                                    const int32_t *__restrict__ indexU, const int32_t *__restrict__ itemU,
                                    const double *__restrict__ Dlu, double *__restrict__ ALlu, double *__restrict__ AUlu) {
  const int s = slot0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= slot1) return;
  const int i = slot_node[s];
  if (i < 0) return;
  const int32_t iL0 = indexL[i], iL1 = indexL[i + 1], iU0 = indexU[i], iU1 = indexU[i + 1];
  for (int32_t kk = iL0; kk < iL1; kk++) {
    const int32_t k = itemL[kk] - 1;
    double dk[9], aik[9];
#pragma unroll
    for (int e = 0; e < 9; e++) { dk[e] = Dlu[(size_t)9 * k + e]; aik[e] = ALlu[(size_t)9 * kk + e]; }
    for (int32_t jj = indexU[k]; jj < indexU[k + 1]; jj++) {
      const int32_t j = itemU[jj] - 1;
      if (j >= N || j == i) continue;  // halo columns only ever multiply zeros; j == i: see k_dlu_natural
      double *dst;
      if (j < i) {
        const int32_t pos = item_find(itemL, iL0, iL1, j + 1);
        if (pos < 0) continue;
        dst = ALlu + (size_t)9 * pos;
      } else {
        const int32_t pos = item_find(itemU, iU0, iU1, j + 1);
        if (pos < 0) continue;
        dst = AUlu + (size_t)9 * pos;
      }
      const double *akj = AUlu + (size_t)9 * jj;
#pragma unroll
      for (int col = 0; col < 3; col++) {
        double x1 = akj[col], x2 = akj[3 + col], x3 = akj[6 + col];
        lusolve33_dev(dk, x1, x2, x3);
        dst[col] -= aik[0] * x1 + aik[1] * x2 + aik[2] * x3;
        dst[3 + col] -= aik[3] * x1 + aik[4] * x2 + aik[5] * x3;
        dst[6 + col] -= aik[6] * x1 + aik[7] * x2 + aik[8] * x3;
      }
    }
  }
}

// The same level with 32 lanes per row: lane t owns destination block t of row i (its nl lower, then nu upper
// entries; nl + nu <= 32), keeps it in registers and walks k over L(i) in ascending order exactly as the sequential
// loop does -- the (i,k) block a step needs is the owner lane's current value, broadcast by shuffle.  Every destination
// receives its updates in the reference's order, so the factors are bit-identical to the one-thread-per-row kernel;
// the 13 x 13 dependent searches of a hex-mesh row become 13 steps of one search per lane (430 -> ~25 us per level).
__global__ __launch_bounds__(256) void k_ilu0_factor_level32(int32_t slot0, int32_t slot1, const int32_t *__restrict__ slot_node,
                                                             int32_t N, const int32_t *__restrict__ indexL,
                                                             const int32_t *__restrict__ itemL,
                                                             const int32_t *__restrict__ indexU,
                                                             const int32_t *__restrict__ itemU, const double *__restrict__ Dlu,
                                                             double *__restrict__ ALlu, double *__restrict__ AUlu) {
  const int t = threadIdx.x & 31;
  const int s = slot0 + blockIdx.x * 8 + (threadIdx.x >> 5);
  if (s >= slot1) return;  // uniform over the 32-lane group
  const int i = slot_node[s];
  if (i < 0) return;
  const int32_t iL0 = indexL[i], nl = indexL[i + 1] - iL0, iU0 = indexU[i], nu = indexU[i + 1] - iU0;
  const bool active = t < nl + nu;
  int32_t jt = -1;
  double *ptr = nullptr;
  if (active) {
    if (t < nl) { jt = itemL[iL0 + t] - 1; ptr = ALlu + (size_t)9 * (iL0 + t); }
    else { jt = itemU[iU0 + (t - nl)] - 1; ptr = AUlu + (size_t)9 * (iU0 + (t - nl)); }
  }
  double a[9];
#pragma unroll
  for (int e = 0; e < 9; e++) a[e] = active ? ptr[e] : 0.0;
  for (int q = 0; q < nl; q++) {
    const int32_t k = itemL[iL0 + q] - 1;
    double aik[9];
#pragma unroll
    for (int e = 0; e < 9; e++) aik[e] = __shfl(a[e], q, 32);
    if (active && jt > k && jt < N) {
      const int32_t pos = item_find(itemU, indexU[k], indexU[k + 1], jt + 1);
      if (pos >= 0) {
        const double *akj = AUlu + (size_t)9 * pos;
        double dk[9];
#pragma unroll
        for (int e = 0; e < 9; e++) dk[e] = Dlu[(size_t)9 * k + e];
#pragma unroll
        for (int col = 0; col < 3; col++) {
          double x1 = akj[col], x2 = akj[3 + col], x3 = akj[6 + col];
          lusolve33_dev(dk, x1, x2, x3);
          a[col] -= aik[0] * x1 + aik[1] * x2 + aik[2] * x3;
          a[3 + col] -= aik[3] * x1 + aik[4] * x2 + aik[5] * x3;
          a[6 + col] -= aik[6] * x1 + aik[7] * x2 + aik[8] * x3;
        }
      }
    }
  }
  if (active) {
#pragma unroll
    for (int e = 0; e < 9; e++) ptr[e] = a[e];
  }
}

// ------------------------------------------------------------------------
// K3/K4: vector kernels (3*nn_internal entries), grid-stride, partial sums per block.
// ------------------------------------------------------------------------
__global__ __launch_bounds__(FX_BLOCK) void k_dot(int64_t n, const double *__restrict__ x,
                                                  const double *__restrict__ y, double *__restrict__ partials,
                                                  const int32_t *__restrict__ gate, int32_t gate_val) {
  if (gate && *gate != gate_val) return;
  double d[1] = {0.0};
  for (int64_t i = (int64_t)blockIdx.x * FX_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * FX_BLOCK)
    d[0] += x[i] * y[i];
  block_sum_store<1>(d, partials, 0);
}

// two dots in one pass: (t.s, t.t) of BiCGSTAB (hecmw_solver_BiCGSTAB.f90:217-218)
__global__ __launch_bounds__(FX_BLOCK) void k_dot2(int64_t n, const double *__restrict__ t,
                                                   const double *__restrict__ s, double *__restrict__ partials,
                                                   int32_t stride, const int32_t *__restrict__ gate) {
  if (gate && *gate != 0) return;
  double d[2] = {0.0, 0.0};
  for (int64_t i = (int64_t)blockIdx.x * FX_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * FX_BLOCK) {
    const double tv = t[i];
    d[0] += tv * s[i];
    d[1] += tv * tv;
  }
  block_sum_store<2>(d, partials, stride);
}

// The CG vector updates walk the vectors as 16-byte words, every workgroup its own contiguous chunk: the traversal that
// streams fastest on MI355X (scripts/stream_probe.hip; a grid-stride sweep of 8-byte words stays ~15 % below it).
// n = 3 * slots is even and the vectors are 256-byte aligned.
#define FX_CHUNK2(n, i, i1)                                                   \
  const int64_t n2_ = (n) >> 1, per_ = (n2_ + gridDim.x - 1) / gridDim.x;     \
  int64_t i = (int64_t)blockIdx.x * per_ + threadIdx.x;                       \
  const int64_t i1 = ((int64_t)(blockIdx.x + 1) * per_ < n2_) ? (int64_t)(blockIdx.x + 1) * per_ : n2_

// CG: p = z + beta p   (hecmw_solver_CG.f90:188-197; beta = 0 on the first iteration)
__global__ __launch_bounds__(FX_BLOCK) void k_cg_update_p(int64_t n, const KrylovState *__restrict__ st,
                                                          const double *__restrict__ z, double *__restrict__ p) {
  if (st->status != 0) return;
  const bool first = (st->iter == 1);
  const double beta = st->beta;
  const fx_d2 *z2 = (const fx_d2 *)z;
  fx_d2 *p2 = (fx_d2 *)p;
  FX_CHUNK2(n, i, i1);
  for (; i < i1; i += FX_BLOCK) {
    const fx_d2 zv = z2[i];
    fx_d2 pv = zv;
    if (!first) { const fx_d2 po = p2[i]; pv.x = zv.x + beta * po.x; pv.y = zv.y + beta * po.y; }
    p2[i] = pv;
  }
}

// CG: x += alpha p ; r -= alpha q ; partial ||r||^2   (hecmw_solver_CG.f90:227-240)
// UPDATE_R false on the iterations that recompute r = b - A x instead (:232-233).
template <bool UPDATE_R>
__global__ __launch_bounds__(FX_BLOCK) void k_cg_update_xr(int64_t n, const KrylovState *__restrict__ st,
                                                           const double *__restrict__ p,
                                                           const double *__restrict__ q, double *__restrict__ x,
                                                           double *__restrict__ r, double *__restrict__ partials) {
  if (st->status != 0) return;
  const double alpha = st->alpha;
  double d[1] = {0.0};
  const fx_d2 *p2 = (const fx_d2 *)p, *q2 = (const fx_d2 *)q;
  fx_d2 *x2 = (fx_d2 *)x, *r2 = (fx_d2 *)r;
  FX_CHUNK2(n, i, i1);
  for (; i < i1; i += FX_BLOCK) {
    fx_d2 xv = x2[i];
    const fx_d2 pv = p2[i];
    xv.x = xv.x + alpha * pv.x; xv.y = xv.y + alpha * pv.y;
    x2[i] = xv;
    if (UPDATE_R) {
      fx_d2 rv = r2[i];
      const fx_d2 qv = q2[i];
      rv.x = rv.x - alpha * qv.x; rv.y = rv.y - alpha * qv.y;
      r2[i] = rv;
      d[0] += rv.x * rv.x + rv.y * rv.y;
    }
  }
  if (UPDATE_R) block_sum_store<1>(d, partials, 0);
}

// BiCGSTAB vector updates (hecmw_solver_BiCGSTAB.f90:160-170, :194-196, :231-241)
__global__ __launch_bounds__(FX_BLOCK) void k_bi_update_p(int64_t n, const KrylovState *__restrict__ st,
                                                          const double *__restrict__ r,
                                                          const double *__restrict__ v, double *__restrict__ p) {
  if (st->status != 0) return;
  const bool first = (st->iter == 1);
  const double beta = st->beta, omega = st->omega;
  for (int64_t i = (int64_t)blockIdx.x * FX_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * FX_BLOCK)
    p[i] = first ? r[i] : r[i] + beta * (p[i] - omega * v[i]);
}

__global__ __launch_bounds__(FX_BLOCK) void k_bi_update_s(int64_t n, const KrylovState *__restrict__ st,
                                                          const double *__restrict__ r,
                                                          const double *__restrict__ v, double *__restrict__ s) {
  if (st->status != 0) return;
  const double alpha = st->alpha;
  for (int64_t i = (int64_t)blockIdx.x * FX_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * FX_BLOCK)
    s[i] = r[i] - alpha * v[i];
}

template <bool UPDATE_R>
__global__ __launch_bounds__(FX_BLOCK) void k_bi_update_xr(int64_t n, const KrylovState *__restrict__ st,
                                                           const double *__restrict__ pt,
                                                           const double *__restrict__ stl,
                                                           const double *__restrict__ s,
                                                           const double *__restrict__ t, double *__restrict__ x,
                                                           double *__restrict__ r, double *__restrict__ partials) {
  if (st->status != 0) return;
  const double alpha = st->alpha, omega = st->omega;
  double d[1] = {0.0};
  for (int64_t i = (int64_t)blockIdx.x * FX_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * FX_BLOCK) {
    x[i] = x[i] + alpha * pt[i] + omega * stl[i];
    if (UPDATE_R) {
      const double rv = s[i] - omega * t[i];
      r[i] = rv;
      d[0] += rv * rv;
    }
  }
  if (UPDATE_R) block_sum_store<1>(d, partials, 0);
}

// On-box streaming ceiling (SURVEY 8d asks for a measured one beside the 8 TB/s vendor peak): read-only sweep with
// non-temporal 16-byte loads, every workgroup walking its own contiguous chunk, 8 loads in flight per lane -- the
// fastest of the patterns probed by scripts/stream_probe.hip on MI355X (7.0-7.1 TB/s; a grid-stride sweep with a
// small grid, the first version of this kernel, stops at 6.2-6.4 TB/s, temporal loads at 6.2).
__global__ __launch_bounds__(FX_BLOCK) void k_stream_read(int64_t n2, const double2 *__restrict__ a,
                                                          double *__restrict__ partials) {
  double d[1] = {0.0};
  double s0 = 0.0, s1 = 0.0;
  const int64_t per = (n2 + gridDim.x - 1) / gridDim.x;
  int64_t i = (int64_t)blockIdx.x * per + threadIdx.x;
  const int64_t end = (int64_t)(blockIdx.x + 1) * per < n2 ? (int64_t)(blockIdx.x + 1) * per : n2;
  const int64_t stride = FX_BLOCK;
  for (; i + 7 * stride < end; i += 8 * stride) {
    fx_d2 v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = __builtin_nontemporal_load((const fx_d2 *)(a + i + k * stride));
#pragma unroll
    for (int k = 0; k < 8; k++) { s0 += v[k].x; s1 += v[k].y; }
  }
  for (; i < end; i += stride) {
    const fx_d2 v = __builtin_nontemporal_load((const fx_d2 *)(a + i));
    s0 += v.x; s1 += v.y;
  }
  d[0] = s0 + s1;
  block_sum_store<1>(d, partials, 0);
}

// SCALING=YES (hecmw_solver_scaling_fw_33 / _bk_33, las/hecmw_solver_scaling_33.f90:20-117, :119-208):
// scale(3i+k) = 1/sqrt|D_i(k,k)|, then A(ij) *= scale(i) scale(j), b *= scale  (back: /=, x *= scale).
__global__ void k_scaling_vector(int32_t N, const double *__restrict__ D, double *__restrict__ scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
#pragma unroll
  for (int k = 0; k < 3; k++) scale[(size_t)3 * i + k] = 1.0 / sqrt(fabs(D[(size_t)9 * i + 4 * k]));
}
template <bool BACK>
__global__ void k_scaling_matrix(int32_t NP, const int32_t *__restrict__ indexL, const int32_t *__restrict__ itemL,
                                 const int32_t *__restrict__ indexU, const int32_t *__restrict__ itemU,
                                 double *__restrict__ D, double *__restrict__ AL, double *__restrict__ AU,
                                 const double *__restrict__ scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NP) return;
  const double si[3] = {scale[(size_t)3 * i], scale[(size_t)3 * i + 1], scale[(size_t)3 * i + 2]};
  auto block = [&](double *v, int32_t j) {
    const double sj[3] = {scale[(size_t)3 * j], scale[(size_t)3 * j + 1], scale[(size_t)3 * j + 2]};
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int q = 0; q < 3; q++) v[3 * r + q] = BACK ? v[3 * r + q] / (si[r] * sj[q]) : v[3 * r + q] * si[r] * sj[q];
  };
  block(D + (size_t)9 * i, i);
  for (int32_t k = indexL[i]; k < indexL[i + 1]; k++) block(AL + (size_t)9 * k, itemL[k] - 1);
  for (int32_t k = indexU[i]; k < indexU[i + 1]; k++) block(AU + (size_t)9 * k, itemU[k] - 1);
}
template <bool BACK>
__global__ void k_scaling_rhs(int64_t n3, const double *__restrict__ scale, double *__restrict__ B, double *__restrict__ X) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n3; i += (int64_t)gridDim.x * blockDim.x) {
    if (BACK) { X[i] = X[i] * scale[i]; B[i] = B[i] / scale[i]; }
    else B[i] = B[i] * scale[i];
  }
}

__global__ void k_copy(int64_t n, const double *__restrict__ a, double *__restrict__ b) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    b[i] = a[i];
}

__global__ void k_axpy_plain(int64_t n, double a, const double *__restrict__ x, double *__restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] += a * x[i];
}

// ------------------------------------------------------------------------
// Scalar stage: reduce the per-block partials in a fixed order, then the
// reference's scalar logic, entirely on the device (no host round trip per dot).
// phase 0: reduce + logic (single GPU);  1: reduce only -> red[slot..] (an RCCL
// all-reduce follows);  2: logic only from red[].
// ------------------------------------------------------------------------
enum ScalarOp {
  OP_BNRM2 = 0,    // ||b||^2, hecmw_solver_CG.f90:123-129
  OP_CG_RHO,       // :168-180 (+ beta :193)
  OP_CG_C1,        // :211-219
  OP_RESID,        // :240-267 (shared with BiCGSTAB :243-262)
  OP_VERIFY,       // :261-266 true-residual re-check
  OP_BI_RHO,       // hecmw_solver_BiCGSTAB.f90:152, :161
  OP_BI_C2,        // :188-190
  OP_BI_OMEGA,     // :217-226
  OP_PLAIN,        // just the sum(s) -> red[]
  OP_RESID_RHO     // Eisenstat's form: ||r||^2 of iteration k (:240-267) and rho of iteration k+1 (:168-193) in ONE stage -- both partial
                   // sets leave k_eis_update together, so a decomposed run needs one 2-double all-reduce for them instead of two
};

__device__ __forceinline__ double reduce_partials(const double *partials, int n) {
  __shared__ double sm[1024 / 64];
  double s = 0.0;
  int i = threadIdx.x;
  const int st = blockDim.x;
  for (; i + 7 * st < n; i += 8 * st) {  // 8 independent loads in flight per thread, fixed summation order
    const double a0 = partials[i], a1 = partials[i + st], a2 = partials[i + 2 * st], a3 = partials[i + 3 * st];
    const double a4 = partials[i + 4 * st], a5 = partials[i + 5 * st], a6 = partials[i + 6 * st], a7 = partials[i + 7 * st];
    s += ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  }
  for (; i < n; i += st) s += partials[i];
  s = wave_sum(s);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sm[w] = s;
  __syncthreads();
  double tot = 0.0;
  if (threadIdx.x == 0)
    for (int k = 0; k < (int)(blockDim.x >> 6); k++) tot += sm[k];
  return tot;  // valid in thread 0
}

template <int OP>
__global__ void k_scalar(const double *__restrict__ partials, int32_t nparts, int32_t stride,
                         KrylovState *__restrict__ st, double *__restrict__ hist, double *__restrict__ red,
                         int phase, int32_t recompute_every) {
  // gating: VERIFY runs only when a check is pending, everything else only while running
  if (OP == OP_VERIFY) { if ((st->status != 0 && st->status != FX_ST_PAUSED) || st->need_verify != 1) return; }
  else if (OP != OP_BNRM2 && OP != OP_PLAIN) { if (st->status != 0) return; }
  double v0 = 0.0, v1 = 0.0;
  if (phase != 2) {
    v0 = reduce_partials(partials, nparts);
    if (OP == OP_BI_OMEGA || OP == OP_RESID_RHO || (OP == OP_PLAIN && stride > 0)) v1 = reduce_partials(partials + stride, nparts);
    if (phase == 1 || OP == OP_PLAIN) {
      if (threadIdx.x == 0) { red[0] = v0; red[1] = v1; }
      return;
    }
  }
  if (threadIdx.x != 0) return;
  if (phase == 2) { v0 = red[0]; v1 = red[1]; }
  if (OP == OP_BNRM2) {
    st->bnrm2 = v0;
    if (v0 == 0.0) { st->status = 1; st->resid = 0.0; st->iter = 1; }  // MAXIT=0: DO leaves ITER=1
  } else if (OP == OP_CG_RHO) {
    st->t_current = 1;
    st->rho = v0;
    if (v0 == 0.0) { st->status = 1; return; }
    if (st->iter > 1 && v0 * st->rho1 <= 0.0) {
      st->n_indef++;
      if (st->n_indef >= 3) { st->status = FX_ERROR_DIVERGE_PC; return; }
    }
    st->beta = (st->iter > 1) ? v0 / st->rho1 : 0.0;
  } else if (OP == OP_CG_C1) {
    st->c1 = v0;
    if (!(v0 > 0.0)) { st->status = FX_ERROR_DIVERGE_MAT; return; }  // `C1 <= 0` (NaN falls here too)
    st->alpha = st->rho / v0;
  } else if (OP == OP_RESID || OP == OP_VERIFY || OP == OP_RESID_RHO) {
    st->dnrm2 = v0;
    const double resid = sqrt(v0 / st->bnrm2);
    st->resid = resid;
    const int it = st->iter;
    if (!(resid == resid) || resid > 1.0e300) {
      // Breakdown (BiCGSTAB rho or r~.v -> 0; the reference has no guard, hecmw_solver_BiCGSTAB.f90 and
      // SURVEY 3.2, and would spin on NaN until MAXIT).  Deliberate deviation: stop now with the same
      // outcome the reference reaches at MAXIT -- W-3001, not converged.
      st->error = FX_ERROR_NOCONV_MAXIT; st->status = FX_ERROR_NOCONV_MAXIT; st->need_verify = 0;
      return;
    }
    if (OP == OP_RESID || OP == OP_RESID_RHO) {
      if (hist) hist[it - 1] = resid;
      st->n_hist = it;
      if (resid <= st->tol) {
        if (it % recompute_every == 0) { st->status = 1; return; }
        st->need_verify = 1;
        if (st->pause_verify) st->status = FX_ST_PAUSED;  // everything enqueued behind this is a no-op until the host has run the check
        return;
      }
    } else {
      st->need_verify = 0;
      if (resid <= st->tol) { st->status = 1; return; }
      st->status = 0;     // (was FX_ST_PAUSED when the host ran the check) the loop goes on
      st->t_current = 0;  // r now holds the true residual and the loop goes on: the Eisenstat form refreshes t from it
    }
    if (it == st->maxit) { st->error = FX_ERROR_NOCONV_MAXIT; st->status = FX_ERROR_NOCONV_MAXIT; st->iter = it + 1; return; }
    st->rho1 = st->rho;
    st->iter = it + 1;
    if (OP == OP_RESID_RHO) {  // the loop goes on: OP_CG_RHO of iteration it + 1, literally
      st->t_current = 1;
      st->rho = v1;
      if (v1 == 0.0) { st->status = 1; return; }
      if (v1 * st->rho1 <= 0.0) {
        st->n_indef++;
        if (st->n_indef >= 3) { st->status = FX_ERROR_DIVERGE_PC; return; }
      }
      st->beta = v1 / st->rho1;
    }
  } else if (OP == OP_BI_RHO) {
    st->rho = v0;
    st->beta = (st->iter > 1) ? (v0 / st->rho1) * (st->alpha / st->omega) : 0.0;
  } else if (OP == OP_BI_C2) {
    st->c2 = v0;
    st->alpha = st->rho / v0;
  } else if (OP == OP_BI_OMEGA) {
    st->cg0 = v0; st->cg1 = v1;
    st->omega = v0 / v1;
  }
}

// ------------------------------------------------------------------------
// Level ordering on the device (set-up of the multicolour SSOR): the breadth-first levels of hecmw_matrix_ordering_CM
// (ordering_CM_inner, hecmw_matrix_ordering_CM.f90:68-136) on the resident CRS arrays.  The reference appends a node when
// its first parent -- in visiting order -- reaches it, neighbours in the order lower items then upper items; here every
// unvisited neighbour is claimed by the smallest parent position (atomicMin), each parent counts and then writes the
// children it won, in its adjacency order, behind an exclusive scan: the sequence is the sequential one, node for node.
// ------------------------------------------------------------------------
template <class F>
__device__ __forceinline__ void bfs_for_neighbours(int32_t u, int32_t N, const int32_t *__restrict__ indexL,
                                                   const int32_t *__restrict__ itemL, const int32_t *__restrict__ indexU,
                                                   const int32_t *__restrict__ itemU, F f) {
  for (int32_t j = indexL[u]; j < indexL[u + 1]; j++) f(itemL[j] - 1);
  for (int32_t j = indexU[u]; j < indexU[u + 1]; j++) {
    const int32_t v = itemU[j] - 1;
    if (v < N) f(v);  // halo columns are not part of the graph
  }
}
// exclusive scan of n ints by ONE workgroup (n up to a few hundred thousand: slice widths, block sums), total -> *total
__global__ __launch_bounds__(1024) void k_scan_excl(int32_t n, const int32_t *__restrict__ in, int32_t *__restrict__ out,
                                                    int32_t *__restrict__ total) {
  __shared__ int32_t sh[1024];
  const int t = threadIdx.x;
  const int chunk = (n + 1023) / 1024;
  const int a = min(n, t * chunk), b = min(n, a + chunk);
  int32_t s = 0;
  for (int i = a; i < b; i++) s += in[i];
  sh[t] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int32_t v = (t >= d) ? sh[t - d] : 0;
    __syncthreads();
    sh[t] += v;
    __syncthreads();
  }
  int32_t run = sh[t] - s;  // exclusive prefix of this thread's chunk
  for (int i = a; i < b; i++) { const int32_t v = in[i]; out[i] = run; run += v; }
  if (t == 1023) *total = sh[1023];
}
// The walk runs for ALL candidate starts at once and without a host round trip per level (blockIdx.y = start): the level
// state {lo, hi, nlevel, stuck} lives on the device, double-buffered by level parity -- the scan kernel (one workgroup per
// start) writes the next level's state while the write and mark kernels still read the current one.  Frontier entries are
// grouped in virtual blocks of 256 (grid-stride): k_bfsb_count leaves per-parent counts and per-block sums, k_bfsb_scan
// turns the sums into offsets, k_bfsb_write rebuilds the in-block prefix in LDS and appends the children in parent order,
// adjacency order -- the sequential sequence, node for node.  A finished start (hi = N) or a stuck one (nothing discovered
// with nodes left: a disconnected graph, the host walk takes over) turns further levels into no-ops.
struct BfsState { int32_t lo, hi, nlevel, stuck; };
__global__ __launch_bounds__(256) void k_bfsb_claim(int32_t N, const BfsState *__restrict__ st, const int32_t *__restrict__ indexL,
                                                    const int32_t *__restrict__ itemL, const int32_t *__restrict__ indexU,
                                                    const int32_t *__restrict__ itemU, const int32_t *__restrict__ seq_all,
                                                    const uint8_t *__restrict__ seen_all, uint32_t *__restrict__ claim_all) {
  const int s = blockIdx.y;
  const BfsState S = st[s];
  const int32_t nf = S.hi - S.lo;
  const int32_t *seq = seq_all + (size_t)s * N + S.lo;
  const uint8_t *seen = seen_all + (size_t)s * N;
  uint32_t *claim = claim_all + (size_t)s * N;
  for (int32_t q = blockIdx.x * 256 + threadIdx.x; q < nf; q += gridDim.x * 256)
    bfs_for_neighbours(seq[q], N, indexL, itemL, indexU, itemU, [&](int32_t v) {
      if (!seen[v]) atomicMin(&claim[v], (uint32_t)(S.lo + q));
    });
}
__global__ __launch_bounds__(256) void k_bfsb_count(int32_t N, int32_t nvb_max, const BfsState *__restrict__ st,
                                                    const int32_t *__restrict__ indexL, const int32_t *__restrict__ itemL,
                                                    const int32_t *__restrict__ indexU, const int32_t *__restrict__ itemU,
                                                    const int32_t *__restrict__ seq_all, const uint8_t *__restrict__ seen_all,
                                                    const uint32_t *__restrict__ claim_all, int32_t *__restrict__ cnt_all,
                                                    int32_t *__restrict__ bsum_all) {
  __shared__ int32_t sh;
  const int s = blockIdx.y;
  const BfsState S = st[s];
  const int32_t nf = S.hi - S.lo;
  const int32_t *seq = seq_all + (size_t)s * N + S.lo;
  const uint8_t *seen = seen_all + (size_t)s * N;
  const uint32_t *claim = claim_all + (size_t)s * N;
  int32_t *cnt = cnt_all + (size_t)s * N, *bsum = bsum_all + (size_t)s * nvb_max;
  for (int32_t vb = blockIdx.x; (int64_t)vb * 256 < nf; vb += gridDim.x) {
    if (threadIdx.x == 0) sh = 0;
    __syncthreads();
    const int32_t q = vb * 256 + threadIdx.x;
    int32_t k = 0;
    if (q < nf) {
      bfs_for_neighbours(seq[q], N, indexL, itemL, indexU, itemU, [&](int32_t v) { k += !seen[v] && claim[v] == (uint32_t)(S.lo + q); });
      cnt[q] = k;
    }
    if (k) atomicAdd(&sh, k);
    __syncthreads();
    if (threadIdx.x == 0) bsum[vb] = sh;
    __syncthreads();
  }
}
__global__ __launch_bounds__(1024) void k_bfsb_scan(int32_t N, int32_t nvb_max, const BfsState *__restrict__ st,
                                                    BfsState *__restrict__ st_next, const int32_t *__restrict__ bsum_all,
                                                    int32_t *__restrict__ boff_all) {
  __shared__ int32_t sh[1024];
  const int s = blockIdx.x, t = threadIdx.x;
  const BfsState S = st[s];
  const int32_t nf = S.hi - S.lo, n = (nf + 255) / 256;
  const int32_t *in = bsum_all + (size_t)s * nvb_max;
  int32_t *out = boff_all + (size_t)s * nvb_max;
  const int chunk = (n + 1023) / 1024;
  const int a = min(n, t * chunk), b = min(n, a + chunk);
  int32_t sum = 0;
  for (int i = a; i < b; i++) sum += in[i];
  sh[t] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int32_t v = (t >= d) ? sh[t - d] : 0;
    __syncthreads();
    sh[t] += v;
    __syncthreads();
  }
  int32_t run = sh[t] - sum;
  for (int i = a; i < b; i++) { const int32_t v = in[i]; out[i] = run; run += v; }
  if (t == 1023) {
    const int32_t nnew = sh[1023];
    BfsState nx = S;
    if (nf > 0) {
      nx.lo = S.hi;
      nx.hi = S.hi + nnew;
      if (nnew > 0) nx.nlevel = S.nlevel + 1;
      else nx.stuck = S.hi < N;
    }
    st_next[s] = nx;
  }
}
__global__ __launch_bounds__(256) void k_bfsb_write(int32_t N, int32_t nvb_max, const BfsState *__restrict__ st,
                                                    const int32_t *__restrict__ indexL, const int32_t *__restrict__ itemL,
                                                    const int32_t *__restrict__ indexU, const int32_t *__restrict__ itemU,
                                                    int32_t *seq_all, const uint8_t *__restrict__ seen_all,
                                                    const uint32_t *__restrict__ claim_all, const int32_t *__restrict__ cnt_all,
                                                    const int32_t *__restrict__ boff_all) {
  __shared__ int32_t sh[256];
  const int s = blockIdx.y, t = threadIdx.x;
  const BfsState S = st[s];
  const int32_t nf = S.hi - S.lo;
  int32_t *seq0 = seq_all + (size_t)s * N;
  const uint8_t *seen = seen_all + (size_t)s * N;
  const uint32_t *claim = claim_all + (size_t)s * N;
  const int32_t *cnt = cnt_all + (size_t)s * N, *boff = boff_all + (size_t)s * nvb_max;
  for (int32_t vb = blockIdx.x; (int64_t)vb * 256 < nf; vb += gridDim.x) {
    const int32_t q = vb * 256 + t;
    const int32_t k = q < nf ? cnt[q] : 0;
    sh[t] = k;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
      const int32_t v = t >= d ? sh[t - d] : 0;
      __syncthreads();
      sh[t] += v;
      __syncthreads();
    }
    int32_t o = S.hi + boff[vb] + sh[t] - k;
    if (k) bfs_for_neighbours(seq0[S.lo + q], N, indexL, itemL, indexU, itemU, [&](int32_t v) {
      if (!seen[v] && claim[v] == (uint32_t)(S.lo + q)) seq0[o++] = v;
    });
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void k_bfsb_mark(int32_t N, const BfsState *__restrict__ st, const BfsState *__restrict__ st_next,
                                                   const int32_t *__restrict__ seq_all, uint8_t *__restrict__ seen_all) {
  const int s = blockIdx.y;
  const int32_t hi = st[s].hi, nnew = st_next[s].hi - hi;
  const int32_t *seq = seq_all + (size_t)s * N + hi;
  uint8_t *seen = seen_all + (size_t)s * N;
  for (int32_t i = blockIdx.x * 256 + threadIdx.x; i < nnew; i += gridDim.x * 256) seen[seq[i]] = 1;
}

// ------------------------------------------------------------------------
// Greedy capped multicolouring on the device (hecmw_matrix_ordering_MC, hecmw_matrix_ordering_MC.f90:15-72).  The reference
// walks the level sequence once per colour: a node joins the colour unless a neighbour EARLIER in the walk joined it, and
// the colour closes after N / ncolor_in picks.  A colour is therefore the lexicographically first independent set of the
// nodes still uncoloured, cut off at the cap, and the fate of a node depends on earlier positions only:
//   OUT as soon as one earlier neighbour is IN;  IN as soon as all earlier (uncoloured) neighbours are OUT.
// Event-driven, one edge visit per decision instead of one scan per round: cnt[v] = earlier uncoloured neighbours not yet
// OUT.  Nodes with cnt = 0 start as IN; a round lets the new IN nodes mark their later undecided neighbours OUT
// (k_mc_in, compare-and-swap so that each is queued once), then the new OUT nodes decrement their later neighbours'
// counters and whatever reaches zero is IN for the next round (k_mc_out).  A node with an earlier IN neighbour never
// reaches zero (that neighbour does not decrement), so the two transitions cannot collide.  ~260 rounds per colour at
// 150^3 nodes.  The cap is applied afterwards (k_mc_assign): the picks are ranked in walk order, the first `cap` keep the
// colour, everything else returns to the pool -- the sequential walk never visited those.  Same colours and the same order
// inside a colour as the reference's walk (tests: device against host ordering).
//   info[node] = position in the walk << 2 | state;  queues hold node ids;  n_in[r] / n_out[r]: queue lengths of round r.
// ------------------------------------------------------------------------
enum { MC_UNDECIDED = 0, MC_IN = 1, MC_OUT = 2, MC_DONE = 3 };
__global__ void k_mc_pos(int32_t n, const int32_t *__restrict__ seq, int32_t *__restrict__ info) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) info[seq[i]] = i << 2;
}
// Queue append aggregated per wave: ONE atomic on the queue length per wave instruction instead of one per node (3.4 M
// appends to one address per colour otherwise).  Must be reached by the lanes of a wave together.
__device__ __forceinline__ void mc_push(bool pred, int32_t value, int32_t *__restrict__ q, int32_t *__restrict__ counter) {
  const uint64_t m = __ballot(pred);
  if (!m) return;
  const int lane = __lane_id(), leader = __ffsll((unsigned long long)m) - 1;
  int32_t base = 0;
  if (lane == leader) base = atomicAdd(counter, __popcll(m));
  base = __shfl(base, leader);
  if (pred) q[base + __popcll(m & ((1ull << lane) - 1))] = value;
}
// k-th neighbour of v: lower items, then upper items; -1 for a halo column (not part of the graph) or k past the row
__device__ __forceinline__ int32_t mc_neighbour(int32_t k, int32_t l0, int32_t nl, int32_t u0, int32_t nu, int32_t N,
                                                const int32_t *__restrict__ itemL, const int32_t *__restrict__ itemU) {
  if (k < nl) return itemL[l0 + k] - 1;
  if (k < nl + nu) {
    const int32_t w = itemU[u0 + k - nl] - 1;
    return w < N ? w : -1;
  }
  return -1;
}
// The three event kernels give every node HALF A WAVE (32 lanes, one neighbour each per step): the gathers and atomics of a
// node are one instruction in flight instead of a chain of dependent round trips -- a round is latency-, not work-bound.
__global__ __launch_bounds__(256) void k_mc_init(int32_t N, const int32_t *__restrict__ indexL, const int32_t *__restrict__ itemL,
                                                 const int32_t *__restrict__ indexU, const int32_t *__restrict__ itemU, int32_t *info,
                                                 int32_t *__restrict__ cnt, int32_t *__restrict__ inq, int32_t *__restrict__ n_in) {
  const int sub = threadIdx.x & 31, half = (threadIdx.x >> 5) & 1;
  const int32_t v = blockIdx.x * 8 + (threadIdx.x >> 5);
  const int32_t iv = v < N ? info[v] : MC_DONE;
  const bool pool = (iv & 3) != MC_DONE;
  int32_t l0 = 0, nl = 0, u0 = 0, nu = 0;
  if (pool) { l0 = indexL[v]; nl = indexL[v + 1] - l0; u0 = indexU[v]; nu = indexU[v + 1] - u0; }
  int32_t k = 0;
  for (int32_t kb = 0; __any(kb < nl + nu); kb += 32) {
    const int32_t w = mc_neighbour(kb + sub, l0, nl, u0, nu, N, itemL, itemU);
    bool earlier = false;
    if (w >= 0) {
      const int32_t iw = info[w];
      earlier = (iw >> 2) < (iv >> 2) && (iw & 3) != MC_DONE;
    }
    k += __popc((uint32_t)(__ballot(earlier) >> (32 * half)));
  }
  const bool first = pool && sub == 0;
  if (first) {
    cnt[v] = k;
    if (k == 0) info[v] = iv | MC_IN;
  }
  mc_push(first && k == 0, v, inq, n_in);
}
__global__ __launch_bounds__(256) void k_mc_in(int32_t N, const int32_t *__restrict__ indexL, const int32_t *__restrict__ itemL,
                                               const int32_t *__restrict__ indexU, const int32_t *__restrict__ itemU, int32_t *info,
                                               const int32_t *__restrict__ inq, const int32_t *__restrict__ n_in,
                                               int32_t *__restrict__ outq, int32_t *__restrict__ n_out) {
  const int32_t n = *n_in;
  const int sub = threadIdx.x & 31;
  for (int32_t i0 = blockIdx.x * 8; i0 < n; i0 += gridDim.x * 8) {  // wave-uniform trip counts (mc_push)
    const int32_t i = i0 + (threadIdx.x >> 5);
    const int32_t v = i < n ? inq[i] : -1;
    int32_t pv = 0, l0 = 0, nl = 0, u0 = 0, nu = 0;
    if (v >= 0) { pv = info[v] >> 2; l0 = indexL[v]; nl = indexL[v + 1] - l0; u0 = indexU[v]; nu = indexU[v + 1] - u0; }
    for (int32_t kb = 0; __any(kb < nl + nu); kb += 32) {
      const int32_t w = mc_neighbour(kb + sub, l0, nl, u0, nu, N, itemL, itemU);
      bool won = false;
      if (w >= 0) {
        const int32_t iw = info[w];
        won = (iw >> 2) > pv && (iw & 3) == MC_UNDECIDED && atomicCAS(&info[w], iw, iw | MC_OUT) == iw;
      }
      mc_push(won, w, outq, n_out);
    }
  }
}
__global__ __launch_bounds__(256) void k_mc_out(int32_t N, const int32_t *__restrict__ indexL, const int32_t *__restrict__ itemL,
                                                const int32_t *__restrict__ indexU, const int32_t *__restrict__ itemU, int32_t *info,
                                                int32_t *cnt, const int32_t *__restrict__ outq, const int32_t *__restrict__ n_out,
                                                int32_t *__restrict__ inq, int32_t *__restrict__ n_in_next) {
  const int32_t n = *n_out;
  const int sub = threadIdx.x & 31;
  for (int32_t i0 = blockIdx.x * 8; i0 < n; i0 += gridDim.x * 8) {
    const int32_t i = i0 + (threadIdx.x >> 5);
    const int32_t v = i < n ? outq[i] : -1;
    int32_t pv = 0, l0 = 0, nl = 0, u0 = 0, nu = 0;
    if (v >= 0) { pv = info[v] >> 2; l0 = indexL[v]; nl = indexL[v + 1] - l0; u0 = indexU[v]; nu = indexU[v + 1] - u0; }
    for (int32_t kb = 0; __any(kb < nl + nu); kb += 32) {
      const int32_t w = mc_neighbour(kb + sub, l0, nl, u0, nu, N, itemL, itemU);
      bool freed = false;
      if (w >= 0) {
        const int32_t iw = info[w];
        freed = (iw >> 2) > pv && (iw & 3) != MC_DONE && atomicSub(&cnt[w], 1) == 1;
        if (freed) info[w] = (iw & ~3) | MC_IN;  // its last earlier neighbour went OUT: none of them is IN, so nobody marks w OUT
      }
      mc_push(freed, w, inq, n_in_next);
    }
  }
}
// picks of 4096 consecutive positions of the walk (16 per thread): bsum[block]
__global__ __launch_bounds__(256) void k_mc_blockcount(int32_t N, const int32_t *__restrict__ seq, const int32_t *__restrict__ info,
                                                       int32_t *__restrict__ bsum) {
  __shared__ int32_t sh;
  if (threadIdx.x == 0) sh = 0;
  __syncthreads();
  const int64_t p0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 16;
  int32_t k = 0;
  for (int i = 0; i < 16; i++)
    if (p0 + i < N) k += (info[seq[p0 + i]] & 3) == MC_IN;
  if (k) atomicAdd(&sh, k);
  __syncthreads();
  if (threadIdx.x == 0) bsum[blockIdx.x] = sh;
}
// rank the picks in walk order; the first `cap` of them become the colour (perm[coloff + rank] = node), the rest and the
// blocked nodes go back to the pool
__global__ __launch_bounds__(256) void k_mc_assign(int32_t N, int32_t cap, int32_t coloff, const int32_t *__restrict__ seq,
                                                   const int32_t *__restrict__ boff, int32_t *__restrict__ info,
                                                   int32_t *__restrict__ perm) {
  __shared__ int32_t sh[256];
  const int t = threadIdx.x;
  const int64_t p0 = ((int64_t)blockIdx.x * 256 + t) * 16;
  int32_t st[16], node[16];
  int32_t k = 0;
  for (int i = 0; i < 16; i++) {
    node[i] = p0 + i < N ? seq[p0 + i] : -1;
    st[i] = node[i] >= 0 ? info[node[i]] & 3 : MC_DONE;
    k += st[i] == MC_IN;
  }
  sh[t] = k;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    const int32_t v = t >= d ? sh[t - d] : 0;
    __syncthreads();
    sh[t] += v;
    __syncthreads();
  }
  int32_t run = boff[blockIdx.x] + sh[t] - k;
  for (int i = 0; i < 16; i++) {
    if (st[i] != MC_IN && st[i] != MC_OUT) continue;
    int32_t ns = MC_UNDECIDED;
    if (st[i] == MC_IN) {
      if (run < cap) { perm[coloff + run] = node[i]; ns = MC_DONE; }
      run++;
    }
    info[node[i]] = (int32_t)(p0 + i) << 2 | ns;
  }
}

// ------------------------------------------------------------------------
// BELL source maps built on the DEVICE (set-up): which block goes where is a pure function of the resident CRS profile, the
// slot order and -- for the SSOR sweeps -- the new numbering, so it needs neither the host threads nor a 720 MB upload.
// One thread per slot generates its row's entries in the order the sweep consumes them:
//   FULL    D, then the lower items, then the upper items, ascending (the reference's summation order, las_33.f90:263-300)
//   SSOR_L  blocks whose column comes EARLIER in the colour ordering, ascending new index   (SSOR_33.f90:312)
//   SSOR_U  blocks whose column comes LATER, descending new index                          (SSOR_33.f90:369); halo columns dropped
//   ILU_L   lower items ascending (BILU_33.f90:104-111);  ILU_U  upper items descending, halo columns dropped (:133)
//   HALO    the halo-column blocks alone (what the localized preconditioner drops), ascending: Eisenstat's form on a subdomain
// Pass 1 (k_bell_count) gives every slice its width (the longest row) and the block total; an exclusive scan turns widths into
// pair_ptr; pass 2 (k_bell_map) writes column slots and source codes (3 * index + {0 D, 1 AL, 2 AU}, -1 padding) into the layout.
// ------------------------------------------------------------------------
enum BellVariant { BV_FULL = 0, BV_SSOR_L, BV_SSOR_U, BV_ILU_L, BV_ILU_U, BV_HALO };
#define FX_BELL_MAXROW 160

template <int VAR>
__device__ __forceinline__ int bell_row_entries(int32_t r, int32_t self_slot, int32_t N, const int32_t *__restrict__ iL,
                                                const int32_t *__restrict__ jL, const int32_t *__restrict__ iU,
                                                const int32_t *__restrict__ jU, const int32_t *__restrict__ slot_of,
                                                const int32_t *__restrict__ newpos, int32_t *src, int32_t *col, int32_t *key) {
  int k = 0;
  if (VAR == BV_FULL) {
    if (src) { src[0] = 3 * r; col[0] = self_slot; }
    k = 1;
    for (int32_t j = iL[r]; j < iL[r + 1]; j++, k++)
      if (src && k < FX_BELL_MAXROW) { src[k] = 3 * j + 1; col[k] = slot_of[jL[j] - 1]; }
    for (int32_t j = iU[r]; j < iU[r + 1]; j++, k++)
      if (src && k < FX_BELL_MAXROW) { src[k] = 3 * j + 2; col[k] = slot_of[jU[j] - 1]; }
    return k;
  }
  if (VAR == BV_ILU_L) {
    for (int32_t j = iL[r]; j < iL[r + 1]; j++, k++)
      if (src && k < FX_BELL_MAXROW) { src[k] = 3 * j + 1; col[k] = slot_of[jL[j] - 1]; }
    return k;
  }
  if (VAR == BV_HALO) {  // the blocks the localized preconditioner drops: columns owned by other ranks (upper items > N), ascending
    for (int32_t j = iU[r]; j < iU[r + 1]; j++) {
      if (jU[j] <= N) continue;
      if (src && k < FX_BELL_MAXROW) { src[k] = 3 * j + 2; col[k] = slot_of[jU[j] - 1]; }
      k++;
    }
    return k;
  }
  if (VAR == BV_ILU_U) {
    for (int32_t j = iU[r + 1] - 1; j >= iU[r]; j--) {
      if (jU[j] > N) continue;
      if (src && k < FX_BELL_MAXROW) { src[k] = 3 * j + 2; col[k] = slot_of[jU[j] - 1]; }
      k++;
    }
    return k;
  }
  // SSOR_L / SSOR_U: select by the new index of the column, then order by it
  const bool lower = (VAR == BV_SSOR_L);
  const int32_t me = newpos[r];
  for (int32_t j = iL[r]; j < iL[r + 1]; j++) {
    const int32_t co = jL[j] - 1, kp = newpos[co];
    if ((kp < me) != lower) continue;
    if (src && k < FX_BELL_MAXROW) { src[k] = 3 * j + 1; col[k] = slot_of[co]; key[k] = kp; }
    k++;
  }
  for (int32_t j = iU[r]; j < iU[r + 1]; j++) {
    const int32_t co = jU[j] - 1;
    if (co >= N) continue;
    const int32_t kp = newpos[co];
    if ((kp < me) != lower) continue;
    if (src && k < FX_BELL_MAXROW) { src[k] = 3 * j + 2; col[k] = slot_of[co]; key[k] = kp; }
    k++;
  }
  if (src) {  // insertion sort by key (distinct keys: the order is unique); ascending for L, descending for U
    const int n = k < FX_BELL_MAXROW ? k : FX_BELL_MAXROW;
    for (int a = 1; a < n; a++) {
      const int32_t ks = key[a], ss = src[a], cs = col[a];
      int b = a - 1;
      while (b >= 0 && (lower ? key[b] > ks : key[b] < ks)) { key[b + 1] = key[b]; src[b + 1] = src[b]; col[b + 1] = col[b]; b--; }
      key[b + 1] = ks; src[b + 1] = ss; col[b + 1] = cs;
    }
  }
  return k;
}

template <int VAR>
__global__ __launch_bounds__(256) void k_bell_count(int32_t nslots, const int32_t *__restrict__ slot_row, int32_t N,
                                                    const int32_t *__restrict__ iL, const int32_t *__restrict__ jL,
                                                    const int32_t *__restrict__ iU, const int32_t *__restrict__ jU,
                                                    const int32_t *__restrict__ newpos, int32_t *__restrict__ width,
                                                    unsigned long long *__restrict__ totals /* [0] blocks, [1] positions, [2] longest row */) {
  const int slot = blockIdx.x * 256 + threadIdx.x;
  int k = 0;
  if (slot < nslots) {
    const int32_t r = slot_row[slot];
    if (r >= 0) k = bell_row_entries<VAR>(r, slot, N, iL, jL, iU, jU, nullptr, newpos, nullptr, nullptr, nullptr);
  }
  int w = k;
  long long sum = k;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    w = max(w, __shfl_down(w, off, 64));
    sum += __shfl_down(sum, off, 64);
  }
  if ((threadIdx.x & 63) == 0 && slot < nslots) {
    width[slot >> 6] = w;
    atomicAdd(&totals[0], (unsigned long long)sum);
    atomicAdd(&totals[1], (unsigned long long)w);
    atomicMax(&totals[2], (unsigned long long)w);
  }
}

template <int VAR>
__global__ __launch_bounds__(64) void k_bell_map(int32_t nslots, int32_t nslices, const int32_t *__restrict__ slot_row, int32_t N,
                                                 const int32_t *__restrict__ iL, const int32_t *__restrict__ jL,
                                                 const int32_t *__restrict__ iU, const int32_t *__restrict__ jU,
                                                 const int32_t *__restrict__ slot_of, const int32_t *__restrict__ newpos,
                                                 const int32_t *__restrict__ pair_ptr, int *__restrict__ col2, int *__restrict__ src2) {
  const int slice = blockIdx.x, lane = threadIdx.x;
  const int slot = slice * 64 + lane;
  const int32_t h0 = pair_ptr[slice], h1 = pair_ptr[slice + 1];
  const int32_t npair2 = ((h1 - h0) >> 1) << 1;
  int32_t src[FX_BELL_MAXROW], col[FX_BELL_MAXROW], key[FX_BELL_MAXROW];
  int k = 0;
  const int32_t r = slot < nslots ? slot_row[slot] : -1;
  if (r >= 0) k = bell_row_entries<VAR>(r, slot, N, iL, jL, iU, jU, slot_of, newpos, src, col, key);
  const int32_t self = min(slot, nslices * 64 - 1);  // padding: value 0 against the row's own (always valid) vector slot
  for (int32_t q = 0; q < h1 - h0; q++) {
    const size_t idx = q < npair2 ? (size_t)(h0 + (q & ~1)) * 64 + (size_t)lane * 2 + (q & 1) : (size_t)(h0 + q) * 64 + lane;
    col2[idx] = q < k ? col[q] : self;
    src2[idx] = q < k ? src[q] : -1;
  }
}

// natural <-> slot numbering of a 3-dof vector
__global__ void k_to_slots(int32_t vslots, const int32_t *__restrict__ slot_node, const double *__restrict__ nat,
                           double *__restrict__ out) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= vslots) return;
  const int node = slot_node[s];
  double a = 0.0, b = 0.0, c = 0.0;
  if (node >= 0) { a = nat[(size_t)3 * node]; b = nat[(size_t)3 * node + 1]; c = nat[(size_t)3 * node + 2]; }
  out[(size_t)3 * s] = a; out[(size_t)3 * s + 1] = b; out[(size_t)3 * s + 2] = c;
}
__global__ void k_from_slots(int32_t nnode, const int32_t *__restrict__ slot_of, const double *__restrict__ in,
                             double *__restrict__ nat) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nnode) return;
  const int s = slot_of[i];
  nat[(size_t)3 * i] = in[(size_t)3 * s]; nat[(size_t)3 * i + 1] = in[(size_t)3 * s + 1]; nat[(size_t)3 * i + 2] = in[(size_t)3 * s + 2];
}

// Halo pack / unpack (hecmw_solve_send_recv_33, hecmw_solver_SR_33.F90:80-121)
__global__ void k_halo_pack(int32_t n, const int32_t *__restrict__ item, const double *__restrict__ x,
                            double *__restrict__ buf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int node = item[i];
  buf[(size_t)3 * i] = x[(size_t)3 * node];
  buf[(size_t)3 * i + 1] = x[(size_t)3 * node + 1];
  buf[(size_t)3 * i + 2] = x[(size_t)3 * node + 2];
}
__global__ void k_halo_unpack(int32_t n, const int32_t *__restrict__ item, const double *__restrict__ buf,
                              double *__restrict__ x) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int node = item[i];
  x[(size_t)3 * node] = buf[(size_t)3 * i];
  x[(size_t)3 * node + 1] = buf[(size_t)3 * i + 1];
  x[(size_t)3 * node + 2] = buf[(size_t)3 * i + 2];
}
