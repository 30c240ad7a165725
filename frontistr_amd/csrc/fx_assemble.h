// Device assembly kernels of libfistr_hip: C3D8 element stiffness (K10), scatter-add
// into the block CRS arrays (K11) and Dirichlet elimination (K12).
//
// Reference: fstr_StiffMatrix.f90:58-207 (element loop), static_LIB_3dIC.f90:21-215
// (STF_C3D8IC), static_LIB_C3D8.f90:23-200 (STF_C3D8Bbar), static_LIB_3d.f90:47-205
// (STF_C3), hecmw_mat_ass.f90:31-134 (scatter), :292-429 (BC).
//
// Work decomposition: 8 lanes per element (8 elements per wave64).  Lane a owns the 3-row block of node a of the element
// matrix against all its column blocks (8 nodes; for the IC element also its 3 incompatible modes) and accumulates it over
// the 2x2x2 Gauss points in registers; the Jacobian is recomputed per lane (72 FMAs) rather than exchanged.  The IC
// element's static condensation goes through LDS: the mode rows against the node columns are the transposes of what the
// node lanes hold (symmetry), the 9x9 mode block is accumulated by lanes 0..2, one lane factors it (Cholesky; the reference
// inverts it with calInverse utilities.f90:247-316 -- same condensed matrix to rounding), all lanes then condense their own rows
// with two triangular solves.  The scatter goes colour
// by colour (no two elements of a launch share a node) with plain read-modify-writes at positions looked up in a map built
// once per profile and mesh; hardware fp64 atomics + binary searches remain as the fallback (the reference uses
// `!$omp atomic` for the same purpose).
#pragma once
#include "fx_internal.h"

#define FXA_BLOCK 256
#define FXA_LPE(EO) 8                               // lanes per element: lane a owns the 3-row block of node a
#define FXA_BS(EO) ((EO) == 1 ? 128 : 256)          // workgroup size (IC: 16 elements x 3.2 KB of LDS for the condensation)
#define FXA_EPB(EO) (FXA_BS(EO) / FXA_LPE(EO))     // elements per block

__device__ __forceinline__ void hex8_shape_deriv(double xi, double et, double ze, double (&dN)[8][3]) {
  // hex8n.f90:24-53
  const double sx[8] = {-1, 1, 1, -1, -1, 1, 1, -1};
  const double sy[8] = {-1, -1, 1, 1, -1, -1, 1, 1};
  const double sz[8] = {-1, -1, -1, -1, 1, 1, 1, 1};
#pragma unroll
  for (int a = 0; a < 8; a++) {
    const double fx = 1.0 + sx[a] * xi, fy = 1.0 + sy[a] * et, fz = 1.0 + sz[a] * ze;
    dN[a][0] = sx[a] * 0.125 * fy * fz;
    dN[a][1] = sy[a] * 0.125 * fx * fz;
    dN[a][2] = sz[a] * 0.125 * fx * fy;
  }
}

// Jacobian, determinant, inverse (element.f90:772-818) and global derivatives (:693-744)
__device__ __forceinline__ void hex8_global_deriv(const double (&ec)[8][3], double xi, double et, double ze, double &det,
                                                  double (&inv)[3][3], double (&gd)[11][3]) {
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

// strain-displacement block of one node: rows xx,yy,zz,xy,yz,zx (static_LIB_3d.f90:126-136);
// h = (Bbar - g)/3 adds the B-bar dilatational correction (static_LIB_C3D8.f90:103-126).
__device__ __forceinline__ void node_B(const double *g, const double *h, double (&B)[6][3]) {
  B[0][0] = g[0] + h[0]; B[0][1] = h[1];        B[0][2] = h[2];
  B[1][0] = h[0];        B[1][1] = g[1] + h[1]; B[1][2] = h[2];
  B[2][0] = h[0];        B[2][1] = h[1];        B[2][2] = g[2] + h[2];
  B[3][0] = g[1]; B[3][1] = g[0]; B[3][2] = 0.0;
  B[4][0] = 0.0;  B[4][1] = g[2]; B[4][2] = g[1];
  B[5][0] = g[2]; B[5][1] = 0.0;  B[5][2] = g[0];
}

// 1-based binary search as hecmw_array_search_i (hecmw_mat_ass.f90:137-166); returns 0-based
// position or -1.
__device__ __forceinline__ int32_t item_search(const int32_t *item, int32_t lo, int32_t hi, int32_t val) {
  while (lo < hi) {
    const int32_t mid = (lo + hi) >> 1;
    const int32_t v = item[mid];
    if (v < val) lo = mid + 1;
    else if (v > val) hi = mid;
    else return mid;
  }
  return -1;
}

// Position of every (row node a, column node b) block of every element in AL / AU, found once per (profile, mesh) so that the
// assembly kernels do not repeat 56 binary searches per element in every Newton iteration: pos[64 * elem + 8 * a + b]
// (-1: not in the profile; the diagonal entries a == b are unused).
__global__ void k_scatter_map(int32_t n_elem, const int32_t *__restrict__ conn, const int32_t *__restrict__ indexL,
                              const int32_t *__restrict__ itemL, const int32_t *__restrict__ indexU,
                              const int32_t *__restrict__ itemU, int32_t *__restrict__ pos) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)64 * n_elem) return;
  const int32_t elem = (int32_t)(t >> 6), a = (int)(t >> 3) & 7, b = (int)t & 7;
  const int32_t inod = conn[(size_t)8 * elem + a], jnod = conn[(size_t)8 * elem + b];
  int32_t k = 0;
  if (jnod < inod) k = item_search(itemL, indexL[inod - 1], indexL[inod], jnod);
  else if (jnod > inod) k = item_search(itemU, indexU[inod - 1], indexU[inod], jnod);
  pos[t] = k;
}

// First-write flags for the coloured scatter (round 4).  A block of the matrix receives one contribution from every element that holds
// both of its nodes, in the order of the colour launches; the FIRST of them can be stored instead of added -- no read of the old value,
// and no clearing of the matrix before the assembly.  Pass 1: per block the lowest colour among its contributions (atomicMin); pass 2: bit
// 30 of the contribution's map entry (the diagonal's otherwise unused entry too) says "this one is the first".
#define FXA_FIRST_BIT 0x40000000
#define FXA_NO_COLOR 0x7F7F7F7F  // what hipMemset(0x7F) leaves: above every colour
__global__ void k_scatter_first_min(int32_t n_elem, const int32_t *__restrict__ conn, const int32_t *__restrict__ pos,
                                    const int32_t *__restrict__ elem_color, int32_t *__restrict__ minD, int32_t *__restrict__ minL,
                                    int32_t *__restrict__ minU) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)64 * n_elem) return;
  const int32_t elem = (int32_t)(t >> 6), a = (int)(t >> 3) & 7, b = (int)t & 7;
  const int32_t inod = conn[(size_t)8 * elem + a], jnod = conn[(size_t)8 * elem + b], k = pos[t], col = elem_color[elem];
  if (k < 0) return;
  if (inod == jnod) atomicMin(minD + (inod - 1), col);
  else if (jnod < inod) atomicMin(minL + k, col);
  else atomicMin(minU + k, col);
}
__global__ void k_scatter_first_flag(int32_t n_elem, const int32_t *__restrict__ conn, int32_t *__restrict__ pos,
                                     const int32_t *__restrict__ elem_color, const int32_t *__restrict__ minD,
                                     const int32_t *__restrict__ minL, const int32_t *__restrict__ minU) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)64 * n_elem) return;
  const int32_t elem = (int32_t)(t >> 6), a = (int)(t >> 3) & 7, b = (int)t & 7;
  const int32_t inod = conn[(size_t)8 * elem + a], jnod = conn[(size_t)8 * elem + b], k = pos[t], col = elem_color[elem];
  if (k < 0) return;
  const int32_t m = inod == jnod ? minD[inod - 1] : (jnod < inod ? minL[k] : minU[k]);
  if (m == col) pos[t] = k | FXA_FIRST_BIT;
}
// blocks that no element contributes to (a profile wider than the mesh's): they would keep stale values without the clearing
__global__ void k_count_uncovered(int64_t n, const int32_t *__restrict__ m, unsigned long long *__restrict__ count) {
  unsigned long long c = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) c += (m[i] == FXA_NO_COLOR);
  if (c) atomicAdd(count, c);
}

// Global derivatives of node b's shape function at a Gauss point from the stored inverse Jacobian: the expression of
// hex8_global_deriv for one node (hex8n.f90:24-53, element.f90:693-744).
__device__ __forceinline__ void hex8_node_deriv(int b, double xi, double et, double ze, const double *inv, double *g) {
  const double sx = ((b & 3) == 1 || (b & 3) == 2) ? 1.0 : -1.0, sy = (b & 2) ? 1.0 : -1.0, sz = (b & 4) ? 1.0 : -1.0;
  const double fx = 1.0 + sx * xi, fy = 1.0 + sy * et, fz = 1.0 + sz * ze;
  const double d0 = sx * 0.125 * fy * fz, d1 = sy * 0.125 * fx * fz, d2 = sz * 0.125 * fx * fy;
#pragma unroll
  for (int j = 0; j < 3; j++) g[j] = d0 * inv[j] + d1 * inv[3 + j] + d2 * inv[6 + j];
}
// K += B_a^T D B_b * wg for one Gauss point (static_LIB_3d.f90:138-176; D of calElasticMatrix, ElasticLinear.f90:43-55)
__device__ __forceinline__ void btdb_accumulate(const double (&Ba)[6][3], const double (&Bb)[6][3], double D11, double D12, double D44,
                                                double wg, double *K) {
  double DB[6][3];
#pragma unroll
  for (int j = 0; j < 3; j++) {
    DB[0][j] = D11 * Bb[0][j] + D12 * Bb[1][j] + D12 * Bb[2][j];
    DB[1][j] = D12 * Bb[0][j] + D11 * Bb[1][j] + D12 * Bb[2][j];
    DB[2][j] = D12 * Bb[0][j] + D12 * Bb[1][j] + D11 * Bb[2][j];
    DB[3][j] = D44 * Bb[3][j]; DB[4][j] = D44 * Bb[4][j]; DB[5][j] = D44 * Bb[5][j];
  }
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      double s = 0.0;
#pragma unroll
      for (int q = 0; q < 6; q++) s += Ba[q][i] * DB[q][j];
      K[3 * i + j] += s * wg;
    }
}

// Round 2: column strips.  Lane a still owns the 3-row block of node a, but it builds ONE 3x3 block at a time (9 accumulators
// over the 8 Gauss points) and scatters it before the next -- round 1 kept all 8 (11 for the IC element) column blocks live:
// 310-512 VGPRs plus scratch, one wave per SIMD.  What the strips share is small: lane g computes the Jacobian of Gauss point g
// once and leaves its inverse and determinant in LDS (80 doubles per element); a node's global derivatives are three FMAs away
// from that.  The IC element goes in two passes: the mode columns first (27 accumulators, kept for the condensation), the 9x9
// mode block factored in LDS as before, then the node columns, each condensed and scattered as it is finished.  Every entry
// is accumulated over the Gauss points in the same order with the same expressions as before: bit-identical element matrices.
template <int ELEMOPT>
__global__ __launch_bounds__(FXA_BS(ELEMOPT)) void k_assemble_c3d8(int32_t n_elem, const double *__restrict__ coord,
                                                             const int32_t *__restrict__ conn, double D11, double D12,
                                                             double D44, const int32_t *__restrict__ indexL,
                                                             const int32_t *__restrict__ itemL,
                                                             const int32_t *__restrict__ indexU,
                                                             const int32_t *__restrict__ itemU, double *__restrict__ D,
                                                             double *__restrict__ AL, double *__restrict__ AU,
                                                             double *__restrict__ Kout, int32_t *__restrict__ err,
                                                             const int32_t *__restrict__ elem_mat,
                                                             const double *__restrict__ mat_tab,
                                                             const int32_t *__restrict__ elem_list, int32_t e0,
                                                             const int32_t *__restrict__ pos_map) {
  // elem_list != nullptr: positions [e0, n_elem) of elem_list are the elements of ONE colour (no shared nodes), scattered
  // without atomics; nullptr: elements e0..n_elem-1 in their own order with hardware fp64 atomics
  constexpr int EPB = FXA_EPB(ELEMOPT);
  constexpr bool IC = (ELEMOPT == 1);
  __shared__ double Jsh[EPB][8][10];                 // per Gauss point: inverse Jacobian (row-major), determinant
  __shared__ double Ksh[IC ? EPB : 1][9][25];        // IC: mode rows against the node columns
  __shared__ double Xinv[IC ? EPB : 1][9][10];       // IC: the 9x9 mode block, then its Cholesky factor (1 / L_ii in column 9)
  constexpr int LPE = FXA_LPE(ELEMOPT);
  const int el = threadIdx.x / LPE, a = threadIdx.x % LPE;
  const int32_t epos = e0 + blockIdx.x * EPB + el;
  const bool active = epos < n_elem;
  const int32_t elem = (elem_list && active) ? elem_list[epos] : epos;
  const double GP = 0.577350269189626;  // quadrature.f90:83-91, unit weights (:221)
  if (active && elem_mat) {  // several sections: (D11, D12, D44) of this element's material (hecMESH%section_ID)
    const int32_t mid = elem_mat[elem] - 1;
    D11 = mat_tab[3 * mid]; D12 = mat_tab[3 * mid + 1]; D44 = mat_tab[3 * mid + 2];
  }
  int32_t nod[8];
  double c0[10];  // IC: inverse Jacobian at the element centre times its determinant (3dIC.f90:79-81); B-bar: the centroid's inverse (C3D8.f90:72-73)
#pragma unroll
  for (int e = 0; e < 10; e++) c0[e] = 0.0;
  if (active) {
    double ec[8][3];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      nod[j] = conn[(size_t)8 * elem + j];
#pragma unroll
      for (int d = 0; d < 3; d++) ec[j][d] = coord[(size_t)3 * (nod[j] - 1) + d];
    }
    double det, inv[3][3], gd[11][3];
    if (ELEMOPT == 1 || ELEMOPT == 2) {
      hex8_global_deriv(ec, 0.0, 0.0, 0.0, det, inv, gd);
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) c0[3 * i + j] = IC ? inv[i][j] * det : inv[i][j];
    }
    const double xi = (a & 1) ? GP : -GP, et = (a & 2) ? GP : -GP, ze = (a & 4) ? GP : -GP;
    hex8_global_deriv(ec, xi, et, ze, det, inv, gd);  // this lane's Gauss point
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) Jsh[el][a][3 * i + j] = inv[i][j];
    Jsh[el][a][9] = det;
  }
  __syncthreads();
  // B of node b at Gauss point LX (h: the B-bar correction)
  auto node_B_at = [&](int b, int LX, double (&B)[6][3]) {
    const double xi = (LX & 1) ? GP : -GP, et = (LX & 2) ? GP : -GP, ze = (LX & 4) ? GP : -GP;
    double g[3], h[3] = {0.0, 0.0, 0.0};
    hex8_node_deriv(b, xi, et, ze, Jsh[el][LX], g);
    if (ELEMOPT == 2) {
      double bb[3];
      hex8_node_deriv(b, 0.0, 0.0, 0.0, c0, bb);
      h[0] = (bb[0] - g[0]) / 3.0; h[1] = (bb[1] - g[1]) / 3.0; h[2] = (bb[2] - g[2]) / 3.0;
    }
    node_B(g, h, B);
  };
  auto mode_B_at = [&](int m, int LX, double (&B)[6][3]) {  // incompatible-mode derivatives (3dIC.f90:120-122)
    const double xi = (LX & 1) ? GP : -GP, et = (LX & 2) ? GP : -GP, ze = (LX & 4) ? GP : -GP;
    const double x = (m == 0) ? xi : ((m == 1) ? et : ze), det = Jsh[el][LX][9];
    double g[3], h0[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int d = 0; d < 3; d++) g[d] = -2.0 * x * c0[3 * m + d] / det;
    node_B(g, h0, B);
  };
  double tk[IC ? 3 : 1][9];  // IC: row i of K_a,alpha (K_alpha,alpha)^-1
  if (IC) {
    if (active) {
      {  // node row block a against the three mode columns
        double Kam[3][9];
#pragma unroll
        for (int m = 0; m < 3; m++)
#pragma unroll
          for (int e = 0; e < 9; e++) Kam[m][e] = 0.0;
#pragma unroll 1
        for (int LX = 0; LX < 8; LX++) {
          const double wg = Jsh[el][LX][9];
          double Ba[6][3];
          node_B_at(a, LX, Ba);
#pragma unroll
          for (int m = 0; m < 3; m++) {
            double Bb[6][3];
            mode_B_at(m, LX, Bb);
            btdb_accumulate(Ba, Bb, D11, D12, D44, wg, Kam[m]);
          }
        }
        // rows 24..32 against the node columns: the transposes of what the node lanes hold (the element matrix is symmetric)
#pragma unroll
        for (int al = 0; al < 3; al++)
#pragma unroll
          for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) Ksh[el][3 * al + i][3 * a + j] = Kam[al][3 * j + i];
      }
      if (a < 3) {  // lanes 0..2 also own mode row block a against the mode columns
        double Kmm[3][9];
#pragma unroll
        for (int m = 0; m < 3; m++)
#pragma unroll
          for (int e = 0; e < 9; e++) Kmm[m][e] = 0.0;
#pragma unroll 1
        for (int LX = 0; LX < 8; LX++) {
          const double wg = Jsh[el][LX][9];
          double Bm[6][3];
          mode_B_at(a, LX, Bm);
#pragma unroll
          for (int m = 0; m < 3; m++) {
            double Bb[6][3];
            mode_B_at(m, LX, Bb);
            btdb_accumulate(Bm, Bb, D11, D12, D44, wg, Kmm[m]);
          }
        }
#pragma unroll
        for (int be = 0; be < 3; be++)
#pragma unroll
          for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) Xinv[el][3 * a + i][3 * be + j] = Kmm[be][3 * i + j];
      }
    }
    __syncthreads();
    if (active && a == 0) {
      // Cholesky factor of the 9x9 mode block (symmetric positive definite for a valid element), in place, instead of the
      // explicit inverse of calInverse (utilities.f90:247-316): K_cond = K - K_a,alpha (K_alpha,alpha)^-1 K_alpha,b is the
      // same to rounding, the serial part shrinks from ~1500 to ~250 flops.  L in Xinv[el][i][j] (j <= i), 1/L_ii in column 9.
      double (*L)[10] = Xinv[el];
      for (int k = 0; k < 9; k++) {
        double d = L[k][k];
        for (int j = 0; j < k; j++) d -= L[k][j] * L[k][j];
        if (!(d > 1.0e-35)) { if (err) atomicExch(err, 1); d = 1.0; }  // the reference's PIVOT ERROR threshold
        const double lkk = sqrt(d), inv = 1.0 / lkk;
        L[k][k] = lkk;
        L[k][9] = inv;
        for (int i = k + 1; i < 9; i++) {
          double v = L[i][k];
          for (int j = 0; j < k; j++) v -= L[i][j] * L[k][j];
          L[i][k] = v * inv;
        }
      }
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int i = 0; i < 3; i++) {
        double y[9];
#pragma unroll
        for (int q = 0; q < 9; q++) y[q] = Ksh[el][q][3 * a + i];  // K_a,alpha row i (= the published transpose)
#pragma unroll
        for (int q = 0; q < 9; q++) {  // L y' = y
#pragma unroll
          for (int j = 0; j < q; j++) y[q] -= Xinv[el][q][j] * y[j];
          y[q] *= Xinv[el][q][9];
        }
#pragma unroll
        for (int q = 8; q >= 0; q--) {  // L^T z = y'
#pragma unroll
          for (int j = q + 1; j < 9; j++) y[q] -= Xinv[el][j][q] * y[j];
          y[q] *= Xinv[el][q][9];
        }
#pragma unroll
        for (int q = 0; q < 9; q++) tk[i][q] = y[q];
      }
    }
  }
  if (!active) return;
  // Strips, using the symmetry of the element matrix (K_ba = K_ab^T: B_b^T D B_a is the transpose of B_a^T D B_b, and so is the
  // condensation term): lane a computes the blocks (a, a), (a, a+1), (a, a+2), (a, a+3) (mod 8) and -- lanes 0..3 only -- (a, a+4), 36
  // of the 64, and scatters each off-diagonal one twice, as it stands into row a and transposed into row b.  4.5 strips per lane
  // instead of 8; the transposed copies differ from separately accumulated ones in the last bit at most (sums of the same products in the
  // same order, transposed), far inside the 1e-12 of the parity tests.  No other element of the launch touches these rows (colouring).
  const int32_t inod = conn[(size_t)8 * elem + a];
  auto block_ptr = [&](int ra, int rb, int32_t rnod, int32_t cnod, bool &first) -> double * {  // hecmw_mat_add_node, hecmw_mat_ass.f90:72-134
    const int32_t raw = pos_map ? pos_map[(size_t)64 * elem + 8 * ra + rb] : 0;
    first = pos_map && raw >= 0 && (raw & FXA_FIRST_BIT);  // k_scatter_first_flag: no earlier colour launch touches this block
    if (rnod == cnod) return D + (size_t)9 * (rnod - 1);
    if (cnod < rnod) {
      const int32_t k = pos_map ? (raw < 0 ? raw : (raw & ~FXA_FIRST_BIT)) : item_search(itemL, indexL[rnod - 1], indexL[rnod], cnod);
      return k < 0 ? nullptr : AL + (size_t)9 * k;
    }
    const int32_t k = pos_map ? (raw < 0 ? raw : (raw & ~FXA_FIRST_BIT)) : item_search(itemU, indexU[rnod - 1], indexU[rnod], cnod);
    return k < 0 ? nullptr : AU + (size_t)9 * k;
  };
#pragma unroll 1
  for (int st = 0; st < 5; st++) {
    if (st == 4 && a >= 4) break;
    const int b = (a + st) & 7;
    // the destination blocks of this strip and -- coloured scatter: plain read-modify-write -- their old values, requested BEFORE the
    // strip's arithmetic: the reads' latency runs under ~500 flops instead of after them
    double *dst = nullptr, *dstT = nullptr;
    double old[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, oldT[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (!Kout) {
      const int32_t jnod = conn[(size_t)8 * elem + b];
      bool first = false, firstT = false;
      dst = block_ptr(a, b, inod, jnod, first);
      if (st > 0) dstT = block_ptr(b, a, jnod, inod, firstT);
      if (!dst || (st > 0 && !dstT)) { if (err) atomicExch(err, 2); continue; }
#ifndef FXA_EXP_NOSCATTER
      if (elem_list && !first) {  // the first contribution to a block is stored, not added: nothing to read
#pragma unroll
        for (int e = 0; e < 9; e++) old[e] = dst[e];
      }
      if (elem_list && !firstT) {
        if (st > 0) {
#pragma unroll
          for (int e = 0; e < 9; e++) oldT[e] = dstT[e];
        }
      }
#endif
    }
    double K[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#ifdef FXA_EXP_NOCOMPUTE  // timing experiment: the scatter alone
    K[0] = 1e-3 * (double)(a + b);
#else
#pragma unroll 1
    for (int LX = 0; LX < 8; LX++) {
      double Ba[6][3], Bb[6][3];
      node_B_at(a, LX, Ba);
      node_B_at(b, LX, Bb);
      btdb_accumulate(Ba, Bb, D11, D12, D44, Jsh[el][LX][9], K);
    }
#endif
    if (IC) {  // condense (3dIC.f90:206-209)
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
          double sm = 0.0;
#pragma unroll
          for (int q = 0; q < 9; q++) sm += tk[i][q] * Ksh[el][q][3 * b + j];
          K[3 * i + j] -= sm;
        }
    }
    if (Kout) {
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
          Kout[(size_t)elem * 576 + (size_t)(3 * a + i) * 24 + 3 * b + j] = K[3 * i + j];
          if (st > 0) Kout[(size_t)elem * 576 + (size_t)(3 * b + j) * 24 + 3 * a + i] = K[3 * i + j];
        }
      continue;
    }
#ifdef FXA_EXP_NOSCATTER  // timing experiment: the arithmetic alone (one word written so that nothing is optimised away)
    if (K[0] + K[4] + K[8] == 1.2345e300) dst[0] = K[0];
    continue;
#endif
    if (elem_list) {
#pragma unroll
      for (int e = 0; e < 9; e++) dst[e] = old[e] + K[e];
      if (st > 0) {
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int j = 0; j < 3; j++) dstT[3 * j + i] = oldT[3 * j + i] + K[3 * i + j];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 9; e++) unsafeAtomicAdd(dst + e, K[e]);
      if (st > 0) {
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int j = 0; j < 3; j++) unsafeAtomicAdd(dstT + 3 * j + i, K[3 * i + j]);
      }
    }
  }
}

// ---- Dirichlet elimination (hecmw_mat_ass_bc, hecmw_mat_ass.f90:292-429) ----------------
// The reference eliminates one dof at a time; the device does the same algebra in two
// passes over the rows: (1) RHS fix-up of the free rows with the original columns,
// (2) zero prescribed rows/columns, unit diagonal, B = prescribed value.
__global__ void k_bc_mark(int32_t n_bc, const int32_t *__restrict__ node, const int32_t *__restrict__ dof,
                          const double *__restrict__ val, uint8_t *__restrict__ flag, double *__restrict__ bcv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_bc) return;
  if (dof[i] < 1 || dof[i] > 3) return;
  const size_t k = (size_t)3 * (node[i] - 1) + (dof[i] - 1);
  flag[k] = 1;
  bcv[k] = val[i];
}

template <int PASS>
__global__ void k_bc_apply(int32_t NP, const int32_t *__restrict__ indexL, const int32_t *__restrict__ itemL,
                           const int32_t *__restrict__ indexU, const int32_t *__restrict__ itemU, double *__restrict__ D,
                           double *__restrict__ AL, double *__restrict__ AU, double *__restrict__ B,
                           const uint8_t *__restrict__ flag, const double *__restrict__ bcv) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NP) return;
  const bool fr[3] = {flag[(size_t)3 * i] != 0, flag[(size_t)3 * i + 1] != 0, flag[(size_t)3 * i + 2] != 0};
  double acc[3] = {0.0, 0.0, 0.0};
  auto visit = [&](double *blk, int32_t col, bool diag) {
    const bool fc[3] = {flag[(size_t)3 * col] != 0, flag[(size_t)3 * col + 1] != 0, flag[(size_t)3 * col + 2] != 0};
    const bool anyc = fc[0] | fc[1] | fc[2], anyr = fr[0] | fr[1] | fr[2];
    if (!anyc && !anyr) return;
    if (PASS == 1) {
#pragma unroll
      for (int c = 0; c < 3; c++)
        if (fc[c]) {
          const double v = bcv[(size_t)3 * col + c];
          if (v != 0.0) {
#pragma unroll
            for (int r = 0; r < 3; r++)
              if (!fr[r]) acc[r] += blk[3 * r + c] * v;
          }
        }
    } else {
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++)
          if (fr[r] || fc[c]) blk[3 * r + c] = (diag && r == c && fr[r]) ? 1.0 : 0.0;
    }
  };
  visit(D + (size_t)9 * i, i, true);
  for (int32_t j = indexL[i]; j < indexL[i + 1]; j++) visit(AL + (size_t)9 * j, itemL[j] - 1, false);
  for (int32_t j = indexU[i]; j < indexU[i + 1]; j++) visit(AU + (size_t)9 * j, itemU[j] - 1, false);
  if (PASS == 1) {
#pragma unroll
    for (int r = 0; r < 3; r++)
      if (!fr[r] && acc[r] != 0.0) B[(size_t)3 * i + r] -= acc[r];
  } else {
#pragma unroll
    for (int r = 0; r < 3; r++)
      if (fr[r]) B[(size_t)3 * i + r] = bcv[(size_t)3 * i + r];
  }
}

__global__ void k_check_zero_diag(int32_t N, const double *__restrict__ D, int32_t *__restrict__ flag) {
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    const double *d = D + (size_t)9 * i;
    if (fabs(d[0]) == 0.0 || fabs(d[4]) == 0.0 || fabs(d[8]) == 0.0) atomicExch(flag, 1);
  }
}
