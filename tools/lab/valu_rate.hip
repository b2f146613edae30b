// What does a VALU instruction cost a LONE wave on its SIMD (the BiGRU scans' situation: one wavefront per sequence, ~115 VALU
// instructions per time step, measured ~0.6 us per step)?  Chains of dependent / independent v_fma_f32, v_pk_fma_f32 and v_exp_f32, timed with
// s_memtime (shader clock domain) and s_memrealtime (100 MHz): cycles per instruction and the clock the shader really runs at.
//   hipcc --offload-arch=gfx950 -O2 tools/lab/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));
#define N_IT 2048

template <int MODE>
__global__ void rate_kernel(float* out, unsigned long long* tim, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a1, a2}, p3 = {a3, a0};
  const f2 m = {1.0000001f, 0.9999999f}, c = {1e-7f, -1e-7f};
  __builtin_amdgcn_s_waitcnt(0);
  const unsigned long long r0 = wall_clock64();
  const unsigned long long t0 = clock64();
#pragma unroll 1
  for (int i = 0; i < N_IT / 8; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) {                 // dependent v_fma_f32
        a0 = __builtin_fmaf(a0, 1.0000001f, 1e-7f);
      } else if (MODE == 1) {          // four independent chains of v_fma_f32
        a0 = __builtin_fmaf(a0, 1.0000001f, 1e-7f);
        a1 = __builtin_fmaf(a1, 1.0000001f, 1e-7f);
        a2 = __builtin_fmaf(a2, 1.0000001f, 1e-7f);
        a3 = __builtin_fmaf(a3, 1.0000001f, 1e-7f);
      } else if (MODE == 2) {          // dependent v_pk_fma_f32
        p0 = __builtin_elementwise_fma(p0, m, c);
      } else if (MODE == 3) {          // four independent chains of v_pk_fma_f32
        p0 = __builtin_elementwise_fma(p0, m, c);
        p1 = __builtin_elementwise_fma(p1, m, c);
        p2 = __builtin_elementwise_fma(p2, m, c);
        p3 = __builtin_elementwise_fma(p3, m, c);
      } else if (MODE == 4) {          // dependent v_exp_f32 (+ a v_mul to keep it bounded)
        a0 = __builtin_amdgcn_exp2f(a0 * 1e-3f);
      } else {                         // dependent LDS round trip (ds_write_b32 + ds_read_b32)
        __shared__ float sh[256];
        sh[threadIdx.x] = a0;
        __builtin_amdgcn_wave_barrier();
        a0 = sh[threadIdx.x ^ 1] + 1.f;
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  const unsigned long long t1 = clock64();
  const unsigned long long r1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
  if (threadIdx.x == 0) {
    tim[blockIdx.x * 2] = t1 - t0;
    tim[blockIdx.x * 2 + 1] = r1 - r0;
  }
}

template <int MODE>
static void run(const char* name, int per_iter, int grid, int threads) {
  float* out;
  unsigned long long* tim;
  (void)hipMalloc(&out, (size_t)grid * threads * 4);
  (void)hipMalloc(&tim, (size_t)grid * 16);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(rate_kernel<MODE>, dim3(grid), dim3(threads), 0, 0, out, tim, 0.5f);
  (void)hipDeviceSynchronize();
  unsigned long long h[2];
  (void)hipMemcpy(h, tim, 16, hipMemcpyDeviceToHost);
  const double n = (double)N_IT * per_iter;
  const double ns = h[1] * 10.0;      // 100 MHz
  printf("%-46s grid %5d x %3d: %7.2f s_memtime ticks / instr, %6.2f ns / instr (memtime ticks per us: %.0f)\n", name, grid, threads, h[0] / n, ns / n,
         h[0] / (ns * 1e-3));
  (void)hipFree(out);
  (void)hipFree(tim);
}

int main() {
  for (int pass = 0; pass < 2; ++pass) {
    const int grid = pass == 0 ? 1024 : 8192, th = pass == 0 ? 64 : 256;      // one wave per SIMD / the chip full (8 waves per SIMD)
    printf("== %s\n", pass == 0 ? "ONE wave per SIMD (1024 one-wave workgroups)" : "chip full (8192 x 256 threads)");
    run<0>("dependent v_fma_f32", 1, grid, th);
    run<1>("4 independent v_fma_f32 chains", 4, grid, th);
    run<2>("dependent v_pk_fma_f32", 1, grid, th);
    run<3>("4 independent v_pk_fma_f32 chains", 4, grid, th);
    run<4>("dependent v_mul + v_exp_f32", 2, grid, th);
    run<5>("dependent LDS write + read + add (3 instr)", 3, grid, th);
  }
  return 0;
}
