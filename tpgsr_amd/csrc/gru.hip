// Fused bidirectional GRU time-step kernels (hidden = 32) for the recurrent residual blocks
// (GruBlock, model/tsrn.py:491-508).  One wavefront owns one sequence: lanes 0-31 run the forward
// direction, lanes 32-63 the reverse direction, lane j of each half owns hidden unit j and keeps its three
// W_hh rows (96 floats) in VGPRs for all T steps.  The hidden state is exchanged through a 256-byte
// wave-private LDS slot (broadcast ds_read_b128), the output is written straight into the NHWC map, so the
// reference's permute/contiguous/view copies and its `.transpose(-1,-2)` (axis = 1) never exist.
// Gate math == nn.GRU:  r = s(gi_r + W_hr h + b_hr), z likewise, n = tanh(gi_n + r*(W_hn h + b_hn)),
// h' = (1-z)*n + z*h, gi = W_i x + b_i precomputed by the MFMA GEMM (tpgsr_conv_fwd).
#include "common.h"

#define GRU_H 32
#define WAVES_PER_BLOCK 1   // one wavefront per workgroup: the per-step barriers degenerate to wave-local ordering

struct SeqGeom {
  long long base;    // pixel index of t = 0
  long long stride;  // pixel stride between time steps
  int T;
  bool active;
};

__device__ __forceinline__ SeqGeom seq_geom(int s, int N, int H, int W, int axis) {
  SeqGeom g;
  int nseq = axis == 0 ? N * H : N * W;
  g.active = s < nseq;
  if (!g.active) s = 0;
  if (axis == 0) {
    g.base = (long long)s * W;
    g.stride = 1;
    g.T = W;
  } else {
    int n = s / W, col = s - n * W;
    g.base = (long long)n * H * W + col;
    g.stride = W;
    g.T = H;
  }
  return g;
}

__device__ __forceinline__ void load_rows(const float* w_hh, int d, int j, float (&wr)[GRU_H], float (&wz)[GRU_H],
                                          float (&wn)[GRU_H]) {
  const float4* pr = reinterpret_cast<const float4*>(w_hh + ((size_t)(d * 96 + 0 * 32 + j)) * GRU_H);
  const float4* pz = reinterpret_cast<const float4*>(w_hh + ((size_t)(d * 96 + 1 * 32 + j)) * GRU_H);
  const float4* pn = reinterpret_cast<const float4*>(w_hh + ((size_t)(d * 96 + 2 * 32 + j)) * GRU_H);
#pragma unroll
  for (int k = 0; k < GRU_H / 4; ++k) {
    float4 a = pr[k], b = pz[k], c = pn[k];
    wr[4 * k] = a.x; wr[4 * k + 1] = a.y; wr[4 * k + 2] = a.z; wr[4 * k + 3] = a.w;
    wz[4 * k] = b.x; wz[4 * k + 1] = b.y; wz[4 * k + 2] = b.z; wz[4 * k + 3] = b.w;
    wn[4 * k] = c.x; wn[4 * k + 1] = c.y; wn[4 * k + 2] = c.z; wn[4 * k + 3] = c.w;
  }
}

__global__ __launch_bounds__(64) void bigru_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ w_hh,
                                                        const float* __restrict__ b_hh, int N, int H, int W, int axis,
                                                        float* __restrict__ h_out) {
  __shared__ __attribute__((aligned(16))) float hs[WAVES_PER_BLOCK][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int d = lane >> 5, j = lane & 31;
  SeqGeom g = seq_geom(blockIdx.x * WAVES_PER_BLOCK + wave, N, H, W, axis);
  float wr[GRU_H], wz[GRU_H], wn[GRU_H];
  load_rows(w_hh, d, j, wr, wz, wn);
  const float br = b_hh[d * 96 + j], bz = b_hh[d * 96 + 32 + j], bn = b_hh[d * 96 + 64 + j];
  float h = 0.f;
  hs[wave][lane] = 0.f;
  __syncthreads();
  const int T = g.T;
  auto pix_of = [&](int step) { return g.base + (long long)(d == 0 ? step : T - 1 - step) * g.stride; };
  float gr = 0.f, gz = 0.f, gn = 0.f;
  if (g.active) {
    const float* p = gi + pix_of(0) * 192 + d * 96 + j;
    gr = p[0]; gz = p[32]; gn = p[64];
  }
  for (int step = 0; step < T; ++step) {
    const long long pix = pix_of(step);
    float ngr = 0.f, ngz = 0.f, ngn = 0.f;
    if (g.active && step + 1 < T) {  // prefetch the next step's input projections
      const float* p = gi + pix_of(step + 1) * 192 + d * 96 + j;
      ngr = p[0]; ngz = p[32]; ngn = p[64];
    }
    // W_hh h: 4 independent FMA chains per gate (the serial recurrence is latency-bound: one wave per SIMD)
    float ar, az, an;
    {
      float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f, z0 = 0.f, z1 = 0.f, z2 = 0.f, z3 = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f, n3 = 0.f;
      const float4* hp = reinterpret_cast<const float4*>(&hs[wave][d * 32]);
#pragma unroll
      for (int k = 0; k < GRU_H / 4; ++k) {
        float4 hv = hp[k];
        r0 = fmaf(wr[4 * k], hv.x, r0); r1 = fmaf(wr[4 * k + 1], hv.y, r1);
        r2 = fmaf(wr[4 * k + 2], hv.z, r2); r3 = fmaf(wr[4 * k + 3], hv.w, r3);
        z0 = fmaf(wz[4 * k], hv.x, z0); z1 = fmaf(wz[4 * k + 1], hv.y, z1);
        z2 = fmaf(wz[4 * k + 2], hv.z, z2); z3 = fmaf(wz[4 * k + 3], hv.w, z3);
        n0 = fmaf(wn[4 * k], hv.x, n0); n1 = fmaf(wn[4 * k + 1], hv.y, n1);
        n2 = fmaf(wn[4 * k + 2], hv.z, n2); n3 = fmaf(wn[4 * k + 3], hv.w, n3);
      }
      ar = br + ((r0 + r1) + (r2 + r3));
      az = bz + ((z0 + z1) + (z2 + z3));
      an = bn + ((n0 + n1) + (n2 + n3));
    }
    float r = sigmoid_f(gr + ar);
    float z = sigmoid_f(gz + az);
    float n = tanh_f(gn + r * an);
    h = (1.f - z) * n + z * h;
    __syncthreads();  // every lane has consumed the old state
    hs[wave][lane] = h;
    if (g.active) h_out[pix * 64 + d * 32 + j] = h;
    __syncthreads();
    gr = ngr; gz = ngz; gn = ngn;
  }
}

extern "C" int tpgsr_bigru_fwd(const float* gi, const float* w_hh, const float* b_hh, int N, int H, int W, int axis,
                               float* h_out, void* stream) {
  TPGSR_CHECK_ARG(gi && w_hh && b_hh && h_out, "tpgsr_bigru_fwd: null pointer");
  TPGSR_CHECK_ARG(N > 0 && H > 0 && W > 0 && (axis == 0 || axis == 1), "tpgsr_bigru_fwd: bad geometry");
  int nseq = axis == 0 ? N * H : N * W;
  hipLaunchKernelGGL(bigru_fwd_kernel, dim3(cdiv(nseq, WAVES_PER_BLOCK)), dim3(64 * WAVES_PER_BLOCK), 0, (hipStream_t)stream, gi, w_hh, b_hh,
                     N, H, W, axis, h_out);
  TPGSR_LAUNCH_CHECK("tpgsr_bigru_fwd");
}

// ------------------------------------------------------------------------------------------------------
// backward through time
//   inputs : gi (saved input projections), h_out (saved states), dh_out (+ optional dh_out2, summed)
//   outputs: dgi [P][192]  = (dr_pre, dz_pre, dn_pre)   -> dW_ih, db_ih, d(input) by GEMM
//            dgh [P][192]  = (dr_pre, dz_pre, dn_pre*r) -> dW_hh, db_hh by GEMM against the shifted states
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void bigru_bwd_kernel(const float* __restrict__ gi, const float* __restrict__ h_out,
                                                        const float* __restrict__ dh_out, const float* __restrict__ dh_out2,
                                                        const float* __restrict__ w_hh, const float* __restrict__ b_hh,
                                                        int N, int H, int W, int axis, float* __restrict__ dgi,
                                                        float* __restrict__ dgh) {
  __shared__ __attribute__((aligned(16))) float hs[WAVES_PER_BLOCK][64];
  __shared__ __attribute__((aligned(16))) float gs[WAVES_PER_BLOCK][3][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int d = lane >> 5, j = lane & 31;
  SeqGeom g = seq_geom(blockIdx.x * WAVES_PER_BLOCK + wave, N, H, W, axis);
  float wr[GRU_H], wz[GRU_H], wn[GRU_H];      // rows j of W_h{r,z,n}      (gate recompute)
  float tr[GRU_H], tz[GRU_H], tn[GRU_H];      // columns j of W_h{r,z,n}   (dh_prev = W_hh^T dgh)
  load_rows(w_hh, d, j, wr, wz, wn);
#pragma unroll
  for (int i = 0; i < GRU_H; ++i) {
    tr[i] = w_hh[((size_t)(d * 96 + 0 * 32 + i)) * GRU_H + j];
    tz[i] = w_hh[((size_t)(d * 96 + 1 * 32 + i)) * GRU_H + j];
    tn[i] = w_hh[((size_t)(d * 96 + 2 * 32 + i)) * GRU_H + j];
  }
  const float br = b_hh[d * 96 + j], bz = b_hh[d * 96 + 32 + j], bn = b_hh[d * 96 + 64 + j];
  const int T = g.T;
  float dh_carry = 0.f;
  // operands of one step: previous state, input projections, incoming gradient -- prefetched one step ahead so the
  // global loads never sit on the serial dependency chain
  auto fetch = [&](int step, float& hprev, float& gr, float& gz, float& gn, float& dho) {
    hprev = gr = gz = gn = dho = 0.f;
    if (!g.active || step < 0) return;
    const int t = d == 0 ? step : T - 1 - step;
    const int tprev = d == 0 ? t - 1 : t + 1;
    const long long pix = g.base + (long long)t * g.stride;
    if (step > 0) hprev = h_out[(g.base + (long long)tprev * g.stride) * 64 + d * 32 + j];
    const float* p = gi + pix * 192 + d * 96 + j;
    gr = p[0]; gz = p[32]; gn = p[64];
    dho = dh_out[pix * 64 + d * 32 + j];
    if (dh_out2) dho += dh_out2[pix * 64 + d * 32 + j];
  };
  float n_hprev, n_gr, n_gz, n_gn, n_dho;
  fetch(T - 1, n_hprev, n_gr, n_gz, n_gn, n_dho);
  for (int step = T - 1; step >= 0; --step) {   // `step` = position in the direction's own forward order
    const int t = d == 0 ? step : T - 1 - step;
    const long long pix = g.base + (long long)t * g.stride;
    const float hprev = n_hprev, gr = n_gr, gz = n_gz, gn = n_gn;
    const float dh = dh_carry + n_dho;
    fetch(step - 1, n_hprev, n_gr, n_gz, n_gn, n_dho);
    hs[wave][lane] = hprev;
    __syncthreads();
    // W_hh h: 4 independent FMA chains per gate (the serial recurrence is latency-bound: one wave per SIMD)
    float ar, az, an;
    {
      float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f, z0 = 0.f, z1 = 0.f, z2 = 0.f, z3 = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f, n3 = 0.f;
      const float4* hp = reinterpret_cast<const float4*>(&hs[wave][d * 32]);
#pragma unroll
      for (int k = 0; k < GRU_H / 4; ++k) {
        float4 hv = hp[k];
        r0 = fmaf(wr[4 * k], hv.x, r0); r1 = fmaf(wr[4 * k + 1], hv.y, r1);
        r2 = fmaf(wr[4 * k + 2], hv.z, r2); r3 = fmaf(wr[4 * k + 3], hv.w, r3);
        z0 = fmaf(wz[4 * k], hv.x, z0); z1 = fmaf(wz[4 * k + 1], hv.y, z1);
        z2 = fmaf(wz[4 * k + 2], hv.z, z2); z3 = fmaf(wz[4 * k + 3], hv.w, z3);
        n0 = fmaf(wn[4 * k], hv.x, n0); n1 = fmaf(wn[4 * k + 1], hv.y, n1);
        n2 = fmaf(wn[4 * k + 2], hv.z, n2); n3 = fmaf(wn[4 * k + 3], hv.w, n3);
      }
      ar = br + ((r0 + r1) + (r2 + r3));
      az = bz + ((z0 + z1) + (z2 + z3));
      an = bn + ((n0 + n1) + (n2 + n3));
    }
    float r = sigmoid_f(gr + ar);
    float z = sigmoid_f(gz + az);
    float n = tanh_f(gn + r * an);
    float dn_pre = dh * (1.f - z) * (1.f - n * n);
    float dz_pre = dh * (hprev - n) * z * (1.f - z);
    float dr_pre = dn_pre * an * r * (1.f - r);
    float dghn = dn_pre * r;
    if (g.active) {
      float* q = dgi + pix * 192 + d * 96 + j;
      q[0] = dr_pre; q[32] = dz_pre; q[64] = dn_pre;
      float* q2 = dgh + pix * 192 + d * 96 + j;
      q2[0] = dr_pre; q2[32] = dz_pre; q2[64] = dghn;
    }
    gs[wave][0][lane] = dr_pre;
    gs[wave][1][lane] = dz_pre;
    gs[wave][2][lane] = dghn;
    __syncthreads();
    const float4* pr = reinterpret_cast<const float4*>(&gs[wave][0][d * 32]);
    const float4* pz = reinterpret_cast<const float4*>(&gs[wave][1][d * 32]);
    const float4* pn = reinterpret_cast<const float4*>(&gs[wave][2][d * 32]);
    float c0 = dh * z, c1 = 0.f, c2 = 0.f, c3 = 0.f;
#pragma unroll
    for (int k = 0; k < GRU_H / 4; ++k) {
      float4 a = pr[k], b = pz[k], c = pn[k];
      c0 = fmaf(tr[4 * k], a.x, c0); c1 = fmaf(tr[4 * k + 1], a.y, c1);
      c2 = fmaf(tr[4 * k + 2], a.z, c2); c3 = fmaf(tr[4 * k + 3], a.w, c3);
      c0 = fmaf(tz[4 * k], b.x, c0); c1 = fmaf(tz[4 * k + 1], b.y, c1);
      c2 = fmaf(tz[4 * k + 2], b.z, c2); c3 = fmaf(tz[4 * k + 3], b.w, c3);
      c0 = fmaf(tn[4 * k], c.x, c0); c1 = fmaf(tn[4 * k + 1], c.y, c1);
      c2 = fmaf(tn[4 * k + 2], c.z, c2); c3 = fmaf(tn[4 * k + 3], c.w, c3);
    }
    const float acc = (c0 + c1) + (c2 + c3);
    dh_carry = acc;
    // next iteration's hs write is ordered behind this iteration's hs reads by the barrier above;
    // its gs write is ordered behind these gs reads by the next hs barrier.
  }
}

extern "C" int tpgsr_bigru_bwd(const float* gi, const float* h_out, const float* dh_out, const float* dh_out2,
                                const float* w_hh, const float* b_hh, int N, int H, int W, int axis, float* dgi,
                                float* dgh, void* stream) {
  TPGSR_CHECK_ARG(gi && h_out && dh_out && w_hh && b_hh && dgi && dgh, "tpgsr_bigru_bwd: null pointer");
  TPGSR_CHECK_ARG(N > 0 && H > 0 && W > 0 && (axis == 0 || axis == 1), "tpgsr_bigru_bwd: bad geometry");
  int nseq = axis == 0 ? N * H : N * W;
  hipLaunchKernelGGL(bigru_bwd_kernel, dim3(cdiv(nseq, WAVES_PER_BLOCK)), dim3(64 * WAVES_PER_BLOCK), 0, (hipStream_t)stream, gi, h_out,
                     dh_out, dh_out2, w_hh, b_hh, N, H, W, axis, dgi, dgh);
  TPGSR_LAUNCH_CHECK("tpgsr_bigru_bwd");
}
