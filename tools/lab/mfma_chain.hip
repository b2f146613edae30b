// Lab: issue cadence of v_mfma_f32_32x32x16_bf16 on one wave per SIMD as a function of the number of independent accumulator chains,
// and with a few filler instructions between the MFMAs.   hipcc --offload-arch=gfx950 -O3 mfma_chain.hip -o mfma_chain.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int NACC, int FILL>
__global__ __launch_bounds__(256) void chain(float* out, int iters, long long* cyc) {
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      a[i][j] = (__bf16)(float)((threadIdx.x + i + j) & 7);
      b[i][j] = (__bf16)(float)((threadIdx.x * 3 + i + j) & 3);
    }
  floatx16 acc[NACC];
  for (int n = 0; n < NACC; ++n)
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  int f = threadIdx.x;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 12 / NACC; ++rep) {
#pragma unroll
      for (int n = 0; n < NACC; ++n) {
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(rep + n) & 3], b[(rep * 2 + n) & 3], acc[n], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < FILL; ++q) f = f * 3 + q;      // independent integer VALU fillers
      }
    }
  }
  const long long t1 = clock64();
  float s = (float)f;
  for (int n = 0; n < NACC; ++n)
    for (int r = 0; r < 16; ++r) s += acc[n][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC, int FILL>
void run(const char* name, float* out, long long* cyc, int grid) {
  const int iters = 2000;
  hipLaunchKernelGGL((chain<NACC, FILL>), dim3(grid), dim3(256), 0, 0, out, 10, cyc);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((chain<NACC, FILL>), dim3(grid), dim3(256), 0, 0, out, iters, cyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n = 12.0 * iters;
  printf("| %s | %d | %d | %d | %.1f | %.1f | %.0f |\n", name, NACC, FILL, grid, 1e6 * ms / n, (double)c / n,
         grid * 4.0 * n * 32768.0 / (ms * 1e-3) / 1e12);
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&cyc, 8);
  printf("| kernel | accumulators | fillers per MFMA | workgroups (4 waves each) | ns per MFMA per wave | clock64 ticks per MFMA | TFLOP/s |\n|---|---|---|---|---|---|---|\n");
  run<1, 0>("dependent chain", out, cyc, 256);
  run<2, 0>("2 chains", out, cyc, 256);
  run<3, 0>("3 chains", out, cyc, 256);
  run<4, 0>("4 chains", out, cyc, 256);
  run<3, 2>("3 chains + 2 fillers", out, cyc, 256);
  run<3, 4>("3 chains + 4 fillers", out, cyc, 256);
  run<3, 6>("3 chains + 6 fillers", out, cyc, 256);
  run<1, 0>("dependent chain, 2 waves per SIMD", out, cyc, 512);
  run<3, 0>("3 chains, 2 waves per SIMD", out, cyc, 512);
  return 0;
}
