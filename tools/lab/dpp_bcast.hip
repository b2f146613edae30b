// The BiGRU scans' per-step exchange without LDS?  A wave = two directions x 32 hidden units (lane j of a half = unit j); every lane needs
// all 32 values of its half each step (96 in back-propagation).  Today: ds_write + broadcast ds_read_b128 (8 per step forward, 24 backward).
// Here: v_permlane16_swap_b32 (gfx950) builds U = (row 0 in both rows of the half) and V = (row 1 in both rows), and v_fmac_f32 with the
// DPP modifier row_share:i broadcasts lane i of each 16-lane row to the row inside the multiply-add -- 96 FMAs, no LDS.
// Checks the DPP result against the LDS form (same values, other summation order) and times both as a dependent 64-step scan.
//   hipcc --offload-arch=gfx950 -O3 tools/lab/dpp_bcast.hip -o /tmp/dpp_bcast && /tmp/dpp_bcast
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

// acc += w * (lane I of the row of x this lane sits in): ONE instruction, the broadcast is the multiply-add's DPP operand.  (Written as
// update_dpp + fmaf the compiler vectorizes the products into v_pk_fma_f32 behind 32 separate v_mov_b32_dpp per step.)
template <int I>
__device__ __forceinline__ void fmac_row_share(float& acc, const float x, const float w) {
  asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(x), "v"(w), "n"(I));
}

// acc += w[16 half + i] * (value of unit 16 half + i of this lane's direction), i = 0..15, for the three gate rows
template <int I>
__device__ __forceinline__ void dpp_terms(const float (&w)[96], const float u, const float v, float (&acc)[6]) {
  fmac_row_share<I>(acc[0], u, w[I]);
  fmac_row_share<I>(acc[1], v, w[16 + I]);
  fmac_row_share<I>(acc[2], u, w[32 + I]);
  fmac_row_share<I>(acc[3], v, w[48 + I]);
  fmac_row_share<I>(acc[4], u, w[64 + I]);
  fmac_row_share<I>(acc[5], v, w[80 + I]);
  if constexpr (I + 1 < 16) dpp_terms<I + 1>(w, u, v, acc);
}

template <bool DPP>
__global__ __launch_bounds__(64) void scan_kernel(const float* __restrict__ wg, const float* __restrict__ h0, float* __restrict__ out, int steps) {
  __shared__ __attribute__((aligned(16))) float hs[2][64];
  const int lane = threadIdx.x, d = lane >> 5;
  float w[96];      // three gate rows of this lane: w[32 g + i]
#pragma unroll
  for (int i = 0; i < 96; ++i) w[i] = wg[i * 64 + lane];
  float h = h0[blockIdx.x * 64 + lane];
  float y0 = 0.f, y1 = 0.f, y2 = 0.f;
  for (int s = 0; s < steps; ++s) {
    if (DPP) {
      // U: units 0..15 of the direction in both rows of its half, V: units 16..31
      // (the builtin __builtin_amdgcn_permlane16_swap lost its second result whenever the results fed update_dpp -- this compiler folded
      //  both broadcasts onto the first; the instruction itself, with the wait states the hazard rules ask for around it)
      float u = h, v = h;
      asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(u), "+v"(v));
      float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      dpp_terms<0>(w, u, v, acc);
      y0 = acc[0] + acc[1]; y1 = acc[2] + acc[3]; y2 = acc[4] + acc[5];
    } else {
      hs[s & 1][lane] = h;
      __builtin_amdgcn_wave_barrier();
      const float4* hp = reinterpret_cast<const float4*>(&hs[s & 1][d * 32]);
      float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f, c0 = 0.f, c1 = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float4 hv = hp[k];
        const float* wk = w + 4 * k;
        a0 = __builtin_fmaf(wk[0], hv.x, a0); a1 = __builtin_fmaf(wk[1], hv.y, a1); a0 = __builtin_fmaf(wk[2], hv.z, a0); a1 = __builtin_fmaf(wk[3], hv.w, a1);
        b0 = __builtin_fmaf(wk[32], hv.x, b0); b1 = __builtin_fmaf(wk[33], hv.y, b1); b0 = __builtin_fmaf(wk[34], hv.z, b0); b1 = __builtin_fmaf(wk[35], hv.w, b1);
        c0 = __builtin_fmaf(wk[64], hv.x, c0); c1 = __builtin_fmaf(wk[65], hv.y, c1); c0 = __builtin_fmaf(wk[66], hv.z, c0); c1 = __builtin_fmaf(wk[67], hv.w, c1);
      }
      y0 = a0 + a1; y1 = b0 + b1; y2 = c0 + c1;
      __builtin_amdgcn_wave_barrier();
    }
    h = __builtin_fmaf(y0 + y1 * 0.5f + y2 * 0.25f, 0.05f, 0.1f * h);      // (stays bounded)
  }
  float* o = out + (size_t)blockIdx.x * 256 + lane;
  o[0] = h; o[64] = y0; o[128] = y1; o[192] = y2;
}

int main() {
  const int nseq = 768, steps = 64;
  std::vector<float> hw(96 * 64), hh((size_t)nseq * 64);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
  for (auto& x : hw) x = rnd() * 0.35f;
  for (auto& x : hh) x = rnd();
  float *dw, *dh, *o0, *o1;
  (void)hipMalloc(&dw, hw.size() * 4); (void)hipMalloc(&dh, hh.size() * 4);
  (void)hipMalloc(&o0, (size_t)nseq * 1024); (void)hipMalloc(&o1, (size_t)nseq * 1024);
  (void)hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dh, hh.data(), hh.size() * 4, hipMemcpyHostToDevice);
  for (int st : {1, steps}) {
    hipLaunchKernelGGL(scan_kernel<false>, dim3(nseq), dim3(64), 0, 0, dw, dh, o0, st);
    hipLaunchKernelGGL(scan_kernel<true>, dim3(nseq), dim3(64), 0, 0, dw, dh, o1, st);
    (void)hipDeviceSynchronize();
    std::vector<float> r0((size_t)nseq * 256), r1((size_t)nseq * 256);
    (void)hipMemcpy(r0.data(), o0, r0.size() * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(r1.data(), o1, r1.size() * 4, hipMemcpyDeviceToHost);
    double md = 0, mx = 0;
    for (size_t i = 0; i < r0.size(); ++i) { md = std::max(md, (double)std::fabs(r0[i] - r1[i])); mx = std::max(mx, (double)std::fabs(r0[i])); }
    printf("%2d step(s): max |LDS form - DPP form| = %.3e (max |value| %.3f)%s\n", st, md, mx, md < 1e-5 * (st == 1 ? 1 : 50) ? "  OK" : "  MISMATCH");
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int which = 0; which < 2; ++which)
    for (int ns : {768, 3072}) {
      const int st = ns == 768 ? 64 : 16;
      float *hh2, *oo;
      (void)hipMalloc(&hh2, (size_t)ns * 256); (void)hipMalloc(&oo, (size_t)ns * 1024);
      (void)hipMemset(hh2, 0, (size_t)ns * 256);
      for (int rep = 0; rep < 3; ++rep)
        if (which) hipLaunchKernelGGL(scan_kernel<true>, dim3(ns), dim3(64), 0, 0, dw, hh2, oo, st);
        else hipLaunchKernelGGL(scan_kernel<false>, dim3(ns), dim3(64), 0, 0, dw, hh2, oo, st);
      (void)hipEventRecord(e0, 0);
      for (int rep = 0; rep < 20; ++rep)
        if (which) hipLaunchKernelGGL(scan_kernel<true>, dim3(ns), dim3(64), 0, 0, dw, hh2, oo, st);
        else hipLaunchKernelGGL(scan_kernel<false>, dim3(ns), dim3(64), 0, 0, dw, hh2, oo, st);
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      printf("%s form, %4d sequences x %2d steps: %.1f us per launch\n", which ? "DPP" : "LDS", ns, st, 1e3 * ms / 20);
      (void)hipFree(hh2); (void)hipFree(oo);
    }
  return 0;
}
