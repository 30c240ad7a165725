// Read-streaming probe for MI355X (gfx950): what is the highest HBM read rate a plain kernel reaches on this box,
// as a function of grid size, workgroup size, unroll depth, temporal hint and traversal (grid-stride vs one
// contiguous chunk per workgroup)?  Context for roofline.measured_stream_ceiling.
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/stream_probe scripts/stream_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d2 __attribute__((ext_vector_type(2)));

template <int U, bool NT, bool CHUNK>
__global__ void k_read(long n2, const d2 *__restrict__ a, double *__restrict__ out) {
  double s0 = 0, s1 = 0;
  long i, end, stride;
  if (CHUNK) {  // each workgroup walks its own contiguous range
    const long per = (n2 + gridDim.x - 1) / gridDim.x;
    i = (long)blockIdx.x * per + threadIdx.x;
    end = min(n2, (long)(blockIdx.x + 1) * per);
    stride = blockDim.x;
  } else {
    i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    end = n2;
    stride = (long)gridDim.x * blockDim.x;
  }
  for (; i + (U - 1) * stride < end; i += U * stride) {
    d2 v[U];
#pragma unroll
    for (int k = 0; k < U; k++) v[k] = NT ? __builtin_nontemporal_load(a + i + k * stride) : a[i + k * stride];
#pragma unroll
    for (int k = 0; k < U; k++) { s0 += v[k].x; s1 += v[k].y; }
  }
  for (; i < end; i += stride) { d2 v = a[i]; s0 += v.x; s1 += v.y; }
  if (s0 + s1 == 1.2345e-300) out[blockIdx.x] = s0;
}

template <int U, bool NT, bool CHUNK>
static double run(long n2, const d2 *a, double *out, int grid, int block) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_read<U, NT, CHUNK>), dim3(grid), dim3(block), 0, 0, n2, a, out);
  hipEventRecord(e0);
  for (int r = 0; r < 5; r++) hipLaunchKernelGGL((k_read<U, NT, CHUNK>), dim3(grid), dim3(block), 0, 0, n2, a, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return (double)n2 * 16 * 5 / (ms * 1e-3) / 1e9;
}

int main() {
  const long n2 = (long)6800 * 1000 * 1000 / 16;
  d2 *a; double *out;
  hipMalloc(&a, n2 * 16); hipMalloc(&out, 1 << 20);
  hipMemset(a, 0, n2 * 16);
  const int grids[] = {256 * 2, 256 * 4, 256 * 8, 256 * 16, 256 * 32, 256 * 128};
  const int blocks[] = {256, 512, 1024};
  double best = 0;
  for (int b : blocks)
    for (int g : grids) {
      double r[8] = {run<4, true, false>(n2, a, out, g, b), run<8, true, false>(n2, a, out, g, b), run<16, true, false>(n2, a, out, g, b),
                     run<8, false, false>(n2, a, out, g, b), run<4, true, true>(n2, a, out, g, b), run<8, true, true>(n2, a, out, g, b),
                     run<16, true, true>(n2, a, out, g, b), run<8, false, true>(n2, a, out, g, b)};
      printf("block %4d grid %6d | stride: u4nt %6.0f u8nt %6.0f u16nt %6.0f u8 %6.0f | chunk: u4nt %6.0f u8nt %6.0f u16nt %6.0f u8 %6.0f GB/s\n", b, g,
             r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
      for (double v : r) if (v > best) best = v;
      fflush(stdout);
    }
  printf("best %.0f GB/s\n", best);
  return 0;
}
