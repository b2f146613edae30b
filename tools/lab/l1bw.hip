// micro-benchmark: how fast can a CU's waves pull 1 KB MFMA fragments (16 B per lane, coalesced) out of L1 / L2 / LDS?
// usage: l1bw   (prints a table; clocks assumed 2.4 GHz)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int NMFMA>   // MODE 0: buffer_load, 1: global_load, 2: LDS ds_read_b128
__global__ __launch_bounds__(256) void pull(const unsigned char* buf, unsigned mask_kb, int iters, int per_wave, float* sink) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[32768];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 32768 / 16; i += 256) reinterpret_cast<u32x4*>(lds)[i] = u32x4{1u, 2u, 3u, 4u};
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(buf), 0, 0x7fffffff, 0x00020000);
  u32x4 acc = {0, 0, 0, 0};
  floatx16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  unsigned frag = per_wave ? (blockIdx.x * 4 + wave) * 97u : (wave >> 1) * 31u;   // shared: wave pairs read the same stream
  for (int it = 0; it < iters; ++it) {
    u32x4 v[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const unsigned off = ((frag + t) & mask_kb) * 1024u + lane * 16u;
      if (MODE == 0) v[t] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
      else if (MODE == 1) v[t] = *reinterpret_cast<const u32x4*>(buf + off);
      else v[t] = *reinterpret_cast<const u32x4*>(lds + (off & 32767u));
    }
    frag += 3;
    if (NMFMA) {
#pragma unroll
      for (int m = 0; m < NMFMA; ++m)
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v[m % 3]), __builtin_bit_cast(bf16x8, v[(m + 1) % 3]), c, 0, 0, 0);
    } else {
#pragma unroll
      for (int t = 0; t < 3; ++t) acc ^= v[t];
    }
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += c[r];
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w ^ __float_as_uint(s)) == 0x12345u) sink[0] = s;
}

template <int MODE, int NMFMA>
void run(const char* name, const unsigned char* d, unsigned kb, int wgs_per_cu, int per_wave, float* sink) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * wgs_per_cu;
  hipLaunchKernelGGL((pull<MODE, NMFMA>), dim3(grid), dim3(256), 0, 0, d, kb - 1, 100, per_wave, sink);
  hipEventRecord(e0);
  hipLaunchKernelGGL((pull<MODE, NMFMA>), dim3(grid), dim3(256), 0, 0, d, kb - 1, iters, per_wave, sink);
  hipEventRecord(e1);
  hipError_t er = hipEventSynchronize(e1);
  hipError_t er2 = hipGetLastError();
  if (er != hipSuccess || er2 != hipSuccess) printf("  !! %s / %s\n", hipGetErrorString(er), hipGetErrorString(er2));
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)grid * 4 * iters * 3 * 1024;
  const double clk = ms * 1e-3 * 2.4e9;
  printf("%-34s %4u KB  %d WG/CU  %s  %7.3f ms  %6.2f TB/s  %6.1f B/clk/CU", name, kb, wgs_per_cu, per_wave ? "private" : "shared ", ms, bytes / ms * 1e-9,
         bytes / clk / 256);
  if (NMFMA) printf("  MFMA %5.1f%% of peak", 100.0 * ((double)grid * 4 * iters * NMFMA * 32) / (clk * 1024));
  printf("\n");
}

int main() {
  unsigned char* d;
  float* sink;
  hipMalloc(&d, 64 << 20);
  hipMemset(d, 1, 64 << 20);
  hipMalloc(&sink, 64);
  for (int w : {1, 2}) {
    run<0, 0>("buffer_load, stream in L1", d, 8, w, 0, sink);
    run<1, 0>("global_load, stream in L1", d, 8, w, 0, sink);
    run<0, 0>("buffer_load, 256 KB (L2)", d, 256, w, 0, sink);
    run<0, 0>("buffer_load, 2 MB (L2)", d, 2048, w, 0, sink);
    run<0, 0>("buffer_load, 32 MB (MALL)", d, 32768, w, 0, sink);
    run<0, 0>("buffer_load, 256 KB per-wave phase", d, 256, w, 1, sink);
    run<2, 0>("ds_read_b128", d, 32, w, 0, sink);
    run<0, 6>("buffer_load 256 KB + 6 MFMA / 3 KB", d, 256, w, 0, sink);
    run<0, 12>("buffer_load 256 KB + 12 MFMA / 3 KB", d, 256, w, 0, sink);
    run<2, 6>("ds_read + 6 MFMA / 3 KB", d, 32, w, 0, sink);
    run<2, 3>("ds_read + 3 MFMA / 3 KB", d, 32, w, 0, sink);
  }
  return 0;
}
