// Probe for the "tile march" schedule of the level-scheduled sweeps: what does ONE step of a tile cost when the
// previous-level dependencies are the tile's own earlier steps (LDS ring) and everything that comes from memory -- the matrix stream
// and the entries of other tiles, which have slack -- is prefetched DEPTH steps ahead?  Synthetic, structured like the 150^3 cube:
// a tile = 64 lanes marching T steps, a row = 7 block pairs (waves 1..7 take one pair each, wave 0 reduces, "solves" and publishes),
// pairs 0..3 gather entries of another tile from memory (sc1 loads of a vector written by an earlier launch), pairs 4..6 gather from
// the tile's own last steps.  All NT tiles run at once, one workgroup per tile: the number printed is the dependent step time under
// the bandwidth load of the whole chip, the quantity that replaces the 2.0 us hand-off of k_tri_dataflow.
//   hipcc --offload-arch=gfx950 -O3 march_probe.hip -o march_probe && ./march_probe [NT] [T]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define RING 16

__device__ __forceinline__ double ld_sc1(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding vector-memory load of the wave
// (s_waitcnt vmcnt(0)), i.e. it would drain the prefetch pipeline at every step
__device__ __forceinline__ void lds_barrier() {
#ifdef PROBE_SYNCTHREADS
  __syncthreads();
#else
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

#if defined(PROBE_UNIFORM) && defined(PROBE_PUBLISHER)
#define NTHR 512   // six pair waves + finishing wave + store-only publisher: 256 VGPRs per lane, prefetch depth 4 without spills
#define NPW 6
#define PUBW 7
#elif defined(PROBE_UNIFORM)
#define NTHR 512   // every wave stores anyway: wave 1 publishes, no ninth wave
#define NPW 7
#define PUBW 1
#else
#define NTHR 576
#define NPW 7
#define PUBW 8
#endif
template <int DEPTH>
__global__ __launch_bounds__(NTHR) void k_march(int T, const double2 *__restrict__ val, const double *__restrict__ zext,
                                               const double *__restrict__ rhs, double *__restrict__ zout) {
  __shared__ double ring[RING][3][64];
  __shared__ double part[8][3][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, tile = blockIdx.x;
  for (int k = threadIdx.x; k < RING * 3 * 64; k += NTHR) (&ring[0][0][0])[k] = 0.0;
  __syncthreads();
  const size_t tstride = (size_t)7 * 9 * 64;  // double2 per step
  const double2 *vt = val + (size_t)tile * T * tstride + lane;
  const double *ze = zext + (size_t)tile * T * 192;  // the "other tile": [step][k][lane]
  double2 A[DEPTH][9];
  double XE[DEPTH][6];
  double RH[DEPTH][3];
  const bool ext = (w >= 1 && w <= 4);
  auto prefetch = [&](int t, int b) {
#ifdef PROBE_NOLOADS   // the chain alone: LDS gathers, FMAs, two barriers, reduction, substitution, ring write -- nothing from memory
    if (t < 2 * DEPTH) {
#pragma unroll
      for (int e = 0; e < 9; e++) A[b][e] = make_double2(1e-3 * (e + lane), 2e-3);
#pragma unroll
      for (int k = 0; k < 6; k++) XE[b][k] = 1e-3 * k;
#pragma unroll
      for (int k = 0; k < 3; k++) RH[b][k] = 1.0 + k;
    }
    return;
#endif
#ifdef PROBE_UNIFORM
    // EVERY wave issues the same vector-memory instructions at every step (the finishing and the publishing wave load a pair they
    // do not use, steps past the end re-load the last one): no join with different numbers of loads in flight, so the compiler's
    // wait insertion can count, and the younger prefetches stay in flight
    const int tt = t < T ? t : T - 1;
    const int wp = (w >= 1 && w <= NPW) ? w - 1 : 0;
#ifdef PROBE_ACTIVE   // a tile of PROBE_ACTIVE lines instead of 64: the other lanes re-read lane 0's words (no extra bytes, no divergence)
    const double2 *v = vt + (size_t)tt * tstride + (size_t)wp * 9 * 64 - (lane < PROBE_ACTIVE ? 0 : lane);
#else
    const double2 *v = vt + (size_t)tt * tstride + (size_t)wp * 9 * 64;
#endif
#pragma unroll
    for (int e = 0; e < 9; e++) A[b][e] = v[e * 64];
    const int la = (lane + w) & 63, lb = (lane + 2 * w + 1) & 63;
#pragma unroll
    for (int k = 0; k < 3; k++) { XE[b][k] = ld_sc1(ze + (size_t)tt * 192 + k * 64 + la); XE[b][3 + k] = ld_sc1(ze + (size_t)tt * 192 + k * 64 + lb); }
#pragma unroll
    for (int k = 0; k < 3; k++) RH[b][k] = rhs[((size_t)tile * T + tt) * 192 + k * 64 + lane];
#else
    if (t >= T) return;
    if (w == 8) return;   // the publisher wave loads nothing
    if (w >= 1) {
      const double2 *v = vt + (size_t)t * tstride + (size_t)(w - 1) * 9 * 64;
#pragma unroll
      for (int e = 0; e < 9; e++) A[b][e] = v[e * 64];
      if (ext) {
        const int la = (lane + w) & 63, lb = (lane + 2 * w + 1) & 63;
#pragma unroll
        for (int k = 0; k < 3; k++) { XE[b][k] = ld_sc1(ze + (size_t)t * 192 + k * 64 + la); XE[b][3 + k] = ld_sc1(ze + (size_t)t * 192 + k * 64 + lb); }
      }
    } else {
#pragma unroll
      for (int k = 0; k < 3; k++) RH[b][k] = rhs[((size_t)tile * T + t) * 192 + k * 64 + lane];
    }
#endif
  };
#pragma unroll
  for (int b = 0; b < DEPTH; b++) prefetch(b, b);
  for (int t0 = 0; t0 < T; t0 += DEPTH) {
#pragma unroll
    for (int b = 0; b < DEPTH; b++) {
      const int t = t0 + b;
      if (t < T) {
        if (w >= 1 && w <= NPW) {
          double x[6];
          if (ext) {
#pragma unroll
            for (int k = 0; k < 6; k++) x[k] = XE[b][k];
          } else {  // the tile's own earlier steps: (t - 1, lane), (t - 1 - (w - 4), lane - 1)
            const int s1 = (t + RING - 1) & (RING - 1), s2 = (t + RING - 1 - (w - 4)) & (RING - 1), l2 = (lane + 63) & 63;
#pragma unroll
            for (int k = 0; k < 3; k++) { x[k] = ring[s1][k][lane]; x[3 + k] = ring[s2][k][l2]; }
          }
          double s0 = 0.0, s1v = 0.0, s2v = 0.0;
          s0 += fma(A[b][2].x, x[2], fma(A[b][1].x, x[1], A[b][0].x * x[0]));
          s1v += fma(A[b][5].x, x[2], fma(A[b][4].x, x[1], A[b][3].x * x[0]));
          s2v += fma(A[b][8].x, x[2], fma(A[b][7].x, x[1], A[b][6].x * x[0]));
          s0 += fma(A[b][2].y, x[5], fma(A[b][1].y, x[4], A[b][0].y * x[3]));
          s1v += fma(A[b][5].y, x[5], fma(A[b][4].y, x[4], A[b][3].y * x[3]));
          s2v += fma(A[b][8].y, x[5], fma(A[b][7].y, x[4], A[b][6].y * x[3]));
          part[w][0][lane] = s0; part[w][1][lane] = s1v; part[w][2][lane] = s2v;
        }
        lds_barrier();
        if (w == 0) {
          double s0 = part[1][0][lane], s1v = part[1][1][lane], s2v = part[1][2][lane];
#pragma unroll
          for (int k = 2; k <= NPW; k++) { s0 += part[k][0][lane]; s1v += part[k][1][lane]; s2v += part[k][2][lane]; }
          // a 3x3 substitution's worth of dependent arithmetic
          double x1 = (RH[b][0] - s0) * 0.25, x2 = (RH[b][1] - s1v - 0.125 * x1) * 0.25, x3 = (RH[b][2] - s2v - 0.125 * x1 - 0.125 * x2) * 0.25;
          x3 *= 0.5; x2 = (x2 - 0.125 * x3) * 0.5; x1 = (x1 - 0.125 * x2 - 0.125 * x3) * 0.5;
          ring[t & (RING - 1)][0][lane] = x1; ring[t & (RING - 1)][1][lane] = x2; ring[t & (RING - 1)][2][lane] = x3;
#ifdef PROBE_FINISHER_STORES
          double *zo = zout + ((size_t)tile * T + t) * 192 + lane;
          st_sc1(zo, x1); st_sc1(zo + 64, x2); st_sc1(zo + 128, x3);
#endif
        }
        lds_barrier();
#ifdef PROBE_UNIFORM
#ifdef PROBE_PUBLISHER
        if (w == PUBW)   // ONLY the ninth wave stores: it loads nothing that matters, so nobody's loads wait behind a write-through acknowledgement
#endif
        {  // every wave stores three words per step (the publisher the result, the others into the scratch tail of zout)
          double *zo = (w == PUBW ? zout + ((size_t)tile * T + t) * 192 : zout + ((size_t)gridDim.x * T + (size_t)tile * 9 + w) * 192) + lane;
          st_sc1(zo, ring[t & (RING - 1)][0][lane]); st_sc1(zo + 64, ring[t & (RING - 1)][1][lane]); st_sc1(zo + 128, ring[t & (RING - 1)][2][lane]);
        }
#elif !defined(PROBE_FINISHER_STORES)
        if (w == 8) {  // the publisher: the write-through stores (and their acknowledgements, which a wave's later loads queue behind) are nobody's critical path
          double *zo = zout + ((size_t)tile * T + t) * 192 + lane;
          st_sc1(zo, ring[t & (RING - 1)][0][lane]); st_sc1(zo + 64, ring[t & (RING - 1)][1][lane]); st_sc1(zo + 128, ring[t & (RING - 1)][2][lane]);
        }
#endif
      }
      prefetch(t + DEPTH, b);
    }
  }
}

template <int DEPTH>
static float run(int NT, int T, const double2 *val, const double *zext, const double *rhs, double *zout) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_march<DEPTH>), dim3(NT), dim3(NTHR), 0, 0, T, val, zext, rhs, zout);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, 0));
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL((k_march<DEPTH>), dim3(NT), dim3(NTHR), 0, 0, T, val, zext, rhs, zout);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / 3.f;
}

int main(int argc, char **argv) {
  const int NT = argc > 1 ? atoi(argv[1]) : 256, T = argc > 2 ? atoi(argv[2]) : 276;
  const size_t nval = (size_t)NT * T * 7 * 9 * 64, nvec = (size_t)NT * T * 192;
  double2 *val; double *zext, *rhs, *zout;
  CHECK(hipMalloc(&val, nval * 16)); CHECK(hipMalloc(&zext, nvec * 8)); CHECK(hipMalloc(&rhs, nvec * 8)); CHECK(hipMalloc(&zout, (nvec + (size_t)NT * 9 * 192) * 8));
  std::vector<double> h(1 << 20);
  for (size_t i = 0; i < h.size(); i++) h[i] = 1e-3 * (double)((i * 2654435761u) % 1000);
  for (size_t off = 0; off < nval * 2; off += h.size()) CHECK(hipMemcpy((double *)val + off, h.data(), std::min(h.size(), nval * 2 - off) * 8, hipMemcpyHostToDevice));
  for (size_t off = 0; off < nvec; off += h.size()) {
    const size_t n = std::min(h.size(), nvec - off);
    CHECK(hipMemcpy(zext + off, h.data(), n * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(rhs + off, h.data(), n * 8, hipMemcpyHostToDevice));
  }
  const double gb = (double)nval * 16 / 1e9;
  printf("tiles %d, steps %d, matrix stream %.2f GB (a 10.1 M-DOF ILU(0) lower part: 3.4 GB, 1,044 levels per sweep)\n", NT, T, gb);
  float ms;
  ms = run<1>(NT, T, val, zext, rhs, zout); printf("prefetch depth 1: %.3f ms = %.3f us per step, %.0f GB/s\n", ms, 1e3 * ms / T, gb / (ms * 1e-3));
  ms = run<2>(NT, T, val, zext, rhs, zout); printf("prefetch depth 2: %.3f ms = %.3f us per step, %.0f GB/s\n", ms, 1e3 * ms / T, gb / (ms * 1e-3));
  ms = run<4>(NT, T, val, zext, rhs, zout); printf("prefetch depth 4: %.3f ms = %.3f us per step, %.0f GB/s\n", ms, 1e3 * ms / T, gb / (ms * 1e-3));
  ms = run<3>(NT, T, val, zext, rhs, zout); printf("prefetch depth 3: %.3f ms = %.3f us per step, %.0f GB/s\n", ms, 1e3 * ms / T, gb / (ms * 1e-3));
  ms = run<5>(NT, T, val, zext, rhs, zout); printf("prefetch depth 5: %.3f ms = %.3f us per step, %.0f GB/s\n", ms, 1e3 * ms / T, gb / (ms * 1e-3));
  return 0;
}
