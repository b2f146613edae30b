// Additive timing lab for the conv tile loop (gfx950): starts from a register-only MFMA chain shaped like
// conv_fwd_kernel's K loop (18 chunks x 16 v_mfma_f32_32x32x2_f32 on ONE accumulator per wave, 4 waves / workgroup)
// and switches the other ingredients on one by one.  Results are not meaningful numerically; timing only.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_lab mfma_lab.hip && ./mfma_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
#define KC 32
#define ALD 65
enum { F_LDSRD = 1, F_BAR = 2, F_LDSWR = 4, F_GLOAD = 8, F_ACC2 = 16, F_STORE = 32 };

template <int F>
__global__ __launch_bounds__(256) void lab_kernel(const float* __restrict__ in, const float* __restrict__ wt, float* out,
                                                  int nchunks, int Kld) {
  __shared__ float As[KC][ALD];
  __shared__ float Bs[KC][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int aq = tid & 7, am0 = tid >> 3, bk0 = tid >> 4, bc = (tid & 15) * 4;
  const int arow = lane >> 5, acol = wm * 32 + (lane & 31), bcol = wn * 32 + (lane & 31);
  // deterministic LDS content
  for (int i = tid; i < KC * ALD; i += 256) (&As[0][0])[i] = 0.001f * i;
  for (int i = tid; i < KC * 64; i += 256) (&Bs[0][0])[i] = 0.002f * i;
  __syncthreads();
  floatx16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
  float4 ra0 = make_float4(1, 2, 3, 4), ra1 = ra0, rb0 = ra0, rb1 = ra0;
  const size_t m0 = (size_t)blockIdx.x * 64;
  const float4* pa0 = reinterpret_cast<const float4*>(in + (m0 + am0) * Kld) + aq;
  const float4* pa1 = reinterpret_cast<const float4*>(in + (m0 + am0 + 32) * Kld) + aq;
  const float4* pb = reinterpret_cast<const float4*>(wt + (size_t)bk0 * 64 + bc);
  float av0 = As[arow][acol], bv0 = Bs[arow][bcol];
  for (int ch = 0; ch < nchunks; ++ch) {
    if (F & F_GLOAD) {
      ra0 = pa0[ch * 8];
      ra1 = pa1[ch * 8];
      rb0 = pb[ch * 512];
      rb1 = pb[ch * 512 + 256];
    }
    float av[16], bv[16];
    if (F & F_LDSRD) {
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        av[kk] = As[2 * kk + arow][acol];
        bv[kk] = Bs[2 * kk + arow][bcol];
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) av[kk] = av0, bv[kk] = bv0;
    }
    __builtin_amdgcn_sched_barrier(0);
    if (F & F_ACC2) {
#pragma unroll
      for (int kk = 0; kk < 16; kk += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bv[kk], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk + 1], bv[kk + 1], acc1, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bv[kk], acc0, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (F & F_BAR) __syncthreads();
    if (F & F_LDSWR) {
      As[aq * 4 + 0][am0] = ra0.x;
      As[aq * 4 + 1][am0] = ra0.y;
      As[aq * 4 + 2][am0] = ra0.z;
      As[aq * 4 + 3][am0] = ra0.w;
      As[aq * 4 + 0][am0 + 32] = ra1.x;
      As[aq * 4 + 1][am0 + 32] = ra1.y;
      As[aq * 4 + 2][am0 + 32] = ra1.z;
      As[aq * 4 + 3][am0 + 32] = ra1.w;
      *reinterpret_cast<float4*>(&Bs[bk0][bc]) = rb0;
      *reinterpret_cast<float4*>(&Bs[bk0 + 16][bc]) = rb1;
    }
    if (F & F_BAR) __syncthreads();
  }
  if (F & F_STORE) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      out[(m0 + wm * 32 + row) * 64 + bcol] = acc0[r] + acc1[r];
    }
  } else {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    if (s == 12345.678f) out[0] = s;
  }
}

template <int F>
static void run(const char* name, int grid, int nchunks, const float* in, const float* wt, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(lab_kernel<F>, dim3(grid), dim3(256), 0, 0, in, wt, out, nchunks, nchunks * KC);
  hipDeviceSynchronize();
  const int reps = 20;
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(lab_kernel<F>, dim3(grid), dim3(256), 0, 0, in, wt, out, nchunks, nchunks * KC);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double us = 1e3 * ms / reps;
  double fl = 2.0 * grid * 64.0 * 64.0 * nchunks * KC;
  printf("%-44s grid %5d  %8.1f us  %7.2f TFLOP/s\n", name, grid, us, fl / us / 1e6);
}

int main() {
  const int nchunks = 18;
  const int maxgrid = 3072;
  float *in, *wt, *out;
  hipMalloc(&in, (size_t)maxgrid * 64 * nchunks * KC * 4);
  hipMalloc(&wt, (size_t)nchunks * KC * 64 * 4);
  hipMalloc(&out, (size_t)maxgrid * 64 * 64 * 4);
  hipMemset(in, 0, (size_t)maxgrid * 64 * nchunks * KC * 4);
  hipMemset(wt, 0, (size_t)nchunks * KC * 64 * 4);
  for (int grid : {768, 1024, 3072}) {
    run<0>("mfma only (1 acc)", grid, nchunks, in, wt, out);
    run<F_ACC2>("mfma only (2 acc)", grid, nchunks, in, wt, out);
    run<F_LDSRD>("+ LDS fragment reads", grid, nchunks, in, wt, out);
    run<F_LDSRD | F_ACC2>("+ LDS fragment reads (2 acc)", grid, nchunks, in, wt, out);
    run<F_LDSRD | F_BAR>("+ LDS reads + barriers", grid, nchunks, in, wt, out);
    run<F_LDSRD | F_BAR | F_LDSWR>("+ LDS reads + barriers + LDS writes", grid, nchunks, in, wt, out);
    run<F_LDSRD | F_BAR | F_LDSWR | F_GLOAD>("+ global loads", grid, nchunks, in, wt, out);
    run<F_LDSRD | F_BAR | F_LDSWR | F_GLOAD | F_STORE>("+ epilogue store", grid, nchunks, in, wt, out);
    run<F_BAR>("mfma + barriers only", grid, nchunks, in, wt, out);
    run<F_LDSWR | F_BAR>("mfma + barriers + LDS writes", grid, nchunks, in, wt, out);
    run<F_GLOAD | F_LDSWR | F_BAR>("mfma + barriers + LDS writes + gloads", grid, nchunks, in, wt, out);
  }
  return 0;
}
