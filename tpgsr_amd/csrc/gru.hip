// Fused bidirectional GRU time-step kernels (hidden = 32) for the recurrent residual blocks
// (GruBlock, model/tsrn.py:491-508).  One wavefront owns one sequence: lanes 0-31 run the forward
// direction, lanes 32-63 the reverse direction, lane j of each half owns hidden unit j and keeps its three
// W_hh rows (96 floats) in VGPRs for all T steps.  The hidden state is exchanged through a 256-byte
// wave-private LDS slot (broadcast ds_read_b128), the output is written straight into the NHWC map, so the
// reference's permute/contiguous/view copies and its `.transpose(-1,-2)` (axis = 1) never exist.
// The forward pass also stores the gate values (r, z, n, W_hn h + b_hn) of every step, so back-propagation through time
// has nothing to recompute: per step it is one LDS exchange of the gate gradients and the 96-term W_hh^T product.
// The dot products run as v_pk_fma_f32 (two fp32 FMAs per lane per issue).
// Gate math == nn.GRU:  r = s(gi_r + W_hr h + b_hr), z likewise, n = tanh(gi_n + r*(W_hn h + b_hn)),
// h' = (1-z)*n + z*h, gi = W_i x + b_i precomputed by the MFMA GEMM (tpgsr_conv_fwd).
#include "gru_common.h"
#include <stdlib.h>

extern "C" __global__ void gru_gate_math_probe_kernel(const float* x, float* sg, float* th, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    sg[i] = gru_sigmoid(x[i]);
    th[i] = gru_tanh(x[i]);
  }
}
/* test hook (tests/test_gru_gate_math_gpu.py): the recurrence's sigmoid / tanh over n values */
extern "C" int tpgsr_gru_gate_math_probe(const float* x, float* sg, float* th, int n, void* stream) {
  TPGSR_CHECK_ARG(x && sg && th && n > 0, "tpgsr_gru_gate_math_probe: bad arguments");
  hipLaunchKernelGGL(gru_gate_math_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, sg, th, n);
  TPGSR_LAUNCH_CHECK("tpgsr_gru_gate_math_probe");
}

// LAB BUILDS ONLY (-DTPGSR_LAB, tools/lab/gru_bwd_probe.py): parts of bigru_bwd_kernel switched off to time the rest -- bit 0 = operands not
// loaded (constants), 1 = nothing stored, 2 = no LDS exchange / W_hh^T product (dh_carry = dh z), 3 = no time steps at all (launch + W_hh
// load).  Results are garbage with any bit set.
#ifdef TPGSR_LAB
__device__ int g_gru_dbg = 0;
extern "C" int tpgsr_gru_debug(int bits) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_gru_dbg), &bits, sizeof(bits)) == hipSuccess ? 0 : TPGSR_ERR_LAUNCH;
}
#define GRU_DBG() __builtin_amdgcn_readfirstlane(g_gru_dbg)
#else
#define GRU_DBG() 0
#endif

// EXACT: T is a multiple of PF (both scan lengths of the 16 x 64 map with the default PF = 8): no per-step bounds tests, the ring's
// refill is switched off per group of PF steps.  Offsets are running 32-bit element indices advanced by a constant per step.
template <int PF, bool EXACT>
__global__ __launch_bounds__(64) void bigru_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ w_hh,
                                                        const float* __restrict__ b_hh, int N, int H, int W, int axis,
                                                        float* __restrict__ h_out, float* __restrict__ gates) {
  __shared__ __attribute__((aligned(16))) float hs[2][64];   // double-buffered by step parity
  const int lane = threadIdx.x & 63;
  const int d = lane >> 5, j = lane & 31;
  const SeqGeom g = seq_geom(blockIdx.x, N, H, W, axis);
  if (!g.active) return;        // (one wave per workgroup: nobody waits for it)
  // row j of W_hr / W_hz interleaved (one packed FMA feeds both gates), row j of W_hn as k-pairs
  f2 wrz[GRU_H], wn2[GRU_H / 2];
  {
    const float* pr = w_hh + ((size_t)(d * 96 + 0 * 32 + j)) * GRU_H;
    const float* pz = w_hh + ((size_t)(d * 96 + 1 * 32 + j)) * GRU_H;
    const float* pn = w_hh + ((size_t)(d * 96 + 2 * 32 + j)) * GRU_H;
#pragma unroll
    for (int k = 0; k < GRU_H; ++k) wrz[k] = mk2(pr[k], pz[k]);
#pragma unroll
    for (int k = 0; k < GRU_H / 2; ++k) wn2[k] = mk2(pn[2 * k], pn[2 * k + 1]);
  }
  const float br = b_hh[d * 96 + j], bz = b_hh[d * 96 + 32 + j], bn = b_hh[d * 96 + 64 + j];
  float h = 0.f;
  hs[0][lane] = 0.f;
  __builtin_amdgcn_wave_barrier();      // LDS operations of one wave execute in order; this only pins the compiler's order
  const int T = g.T;
  const int dpix = d == 0 ? g.stride : -g.stride;                 // pixel step in the direction's own scan order
  int pix = g.base + (d == 0 ? 0 : (T - 1) * g.stride);           // pixel of the current step
  int fpix = pix;                                                 // pixel of the next step to fetch
  // The input projections do not depend on the recurrence: they are fetched PF steps ahead through a small register ring.  One
  // step of look-ahead left the wave waiting on memory in EVERY step whenever a load took longer than a step -- i.e. always, next to
  // the other kernels of the training step (the backward kernel below learnt this first).
  struct StepIn {
    float r, z, n;
  };
  auto fetch = [&]() __attribute__((always_inline)) {
    StepIn s;
    const float* p = gi + fpix * 192 + d * 96 + j;
    s.r = p[0]; s.z = p[32]; s.n = p[64];
    fpix += dpix;
    return s;
  };
  StepIn ring[PF];
#pragma unroll
  for (int i = 0; i < PF; ++i) ring[i] = (EXACT || i < T) ? fetch() : StepIn{0.f, 0.f, 0.f};
  for (int base = 0; base < T; base += PF) {
    const bool more = base + PF < T;               // (EXACT) the next group exists: refill the ring
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int step = base + i;
      if (!EXACT && step >= T) break;             // wave-uniform
      const StepIn c = ring[i];
      if (EXACT ? more : step + PF < T) ring[i] = fetch();
      // W_hh h: six independent packed-FMA chains, 8 deep
      f2 a0 = mk2(0.f, 0.f), a1 = a0, a2 = a0, a3 = a0, n0 = a0, n1 = a0;
      const float4* hp = reinterpret_cast<const float4*>(&hs[i & 1][d * 32]);      // (PF is even: step parity = i parity)
#pragma unroll
      for (int k = 0; k < GRU_H / 4; ++k) {
        const float4 hv = hp[k];
        a0 = pk_fma(wrz[4 * k], mk2(hv.x, hv.x), a0);
        a1 = pk_fma(wrz[4 * k + 1], gru_dup_odd(hv.y), a1);      // (NOT mk2(hv.y, hv.y): gru_common.h)
        a2 = pk_fma(wrz[4 * k + 2], mk2(hv.z, hv.z), a2);
        a3 = pk_fma(wrz[4 * k + 3], gru_dup_odd(hv.w), a3);
        n0 = pk_fma(wn2[2 * k], mk2(hv.x, hv.y), n0);
        n1 = pk_fma(wn2[2 * k + 1], mk2(hv.z, hv.w), n1);
      }
      const f2 rz = (a0 + a1) + (a2 + a3), nn = n0 + n1;
      const float an = bn + (nn.x + nn.y);
      const f2 sg = gru_sigmoid2(mk2(c.r + (br + rz.x), c.z + (bz + rz.y)));      // both gates in lock step (gru_common.h)
      const float r = sg.x, z = sg.y;
      const float n = gru_tanh(__builtin_fmaf(r, an, c.n));
      h = __builtin_fmaf(z, h, (1.f - z) * n);      // (explicit: the same contraction in every kernel that runs this step)
      hs[(i + 1) & 1][lane] = h;
      h_out[pix * 64 + d * 32 + j] = h;
      if (gates) {
        float* q = gates + pix * 256 + d * 128 + j;
        q[0] = r; q[32] = z; q[64] = n; q[96] = an;
      }
      pix += dpix;
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// look-ahead (in time steps) of the operand prefetch rings of both kernels: TPGSR_GRU_PF = 4 | 8 (default) | 12.  Loads and stores
// retire in order on one counter per wave, so a ring slot is only as far ahead as the stores issued before it allow.
static int g_gru_pf = [] { const char* e = getenv("TPGSR_GRU_PF"); const int v = e ? atoi(e) : 8; return (v == 4 || v == 12) ? v : 8; }();
extern "C" void tpgsr_gru_set_prefetch(int steps) { g_gru_pf = (steps == 4 || steps == 12) ? steps : 8; }

extern "C" int tpgsr_bigru_fwd(const float* gi, const float* w_hh, const float* b_hh, int N, int H, int W, int axis,
                               float* h_out, float* gates, void* stream) {
  TPGSR_CHECK_ARG(gi && w_hh && b_hh && h_out, "tpgsr_bigru_fwd: null pointer");
  TPGSR_CHECK_ARG(N > 0 && H > 0 && W > 0 && (axis == 0 || axis == 1), "tpgsr_bigru_fwd: bad geometry");
  TPGSR_CHECK_ARG((long long)N * H * W * 256 < (1ll << 31), "tpgsr_bigru_fwd: map too large for the kernel's 32-bit indices");
  int nseq = axis == 0 ? N * H : N * W;
  const int T = axis == 0 ? W : H;
#define GRU_FWD_CASE(PF)                                                                                                          \
  if (T % PF == 0)                                                                                                                \
    hipLaunchKernelGGL((bigru_fwd_kernel<PF, true>), dim3(nseq), dim3(64), 0, (hipStream_t)stream, gi, w_hh, b_hh, N, H, W, axis, h_out, gates); \
  else                                                                                                                            \
    hipLaunchKernelGGL((bigru_fwd_kernel<PF, false>), dim3(nseq), dim3(64), 0, (hipStream_t)stream, gi, w_hh, b_hh, N, H, W, axis, h_out, gates);
  switch (g_gru_pf) {
    case 4: GRU_FWD_CASE(4) break;
    case 12: GRU_FWD_CASE(12) break;
    default: GRU_FWD_CASE(8) break;
  }
#undef GRU_FWD_CASE
  TPGSR_LAUNCH_CHECK("tpgsr_bigru_fwd");
}

// ------------------------------------------------------------------------------------------------------
// backward through time
//   inputs : gates (r, z, n, an saved by the forward pass), h_out (saved states), dh_out (+ optional dh_out2, summed)
//   outputs: dgi [P][192]  = (dr_pre, dz_pre, dn_pre)   -> dW_ih, db_ih, d(input) by GEMM
//            dgh [P][192]  = (dr_pre, dz_pre, dn_pre*r) -> dW_hh, db_hh by GEMM against the shifted states
// ------------------------------------------------------------------------------------------------------
// COMPACT: `dgh` is [P][64] and receives only what differs from dgi -- the n gate's hidden-side gradient dn_pre * r of both directions
// (the r and z planes of dgh ARE dgi's: the fused GruBlock weight-gradient kernel, gru_wgrad.hip, reads them there)
template <int PF, bool COMPACT, bool EXACT>
__global__ __launch_bounds__(64) void bigru_bwd_kernel(const float* __restrict__ gates, const float* __restrict__ h_out,
                                                        const float* __restrict__ dh_out, const float* __restrict__ dh_out2,
                                                        const float* __restrict__ w_hh, int N, int H, int W, int axis,
                                                        float* __restrict__ dgi, float* __restrict__ dgh) {
  __shared__ __attribute__((aligned(16))) float g_rz[2][2][64];   // [parity][dir][(dr_i, dz_i) pairs]
  __shared__ __attribute__((aligned(16))) float g_n[2][64];       // [parity][dir*32 + i] = dn_pre_i * r_i
  const int lane = threadIdx.x & 63;
  const int d = lane >> 5, j = lane & 31;
  const SeqGeom g = seq_geom(blockIdx.x, N, H, W, axis);
  if (!g.active) return;
  // column j of W_hr / W_hz interleaved, column j of W_hn as row pairs: dh_prev[j] = sum_i W[i][j] * dgate[i]
  f2 trz[GRU_H], tn2[GRU_H / 2];
#pragma unroll
  for (int i = 0; i < GRU_H; ++i)
    trz[i] = mk2(w_hh[((size_t)(d * 96 + 0 * 32 + i)) * GRU_H + j], w_hh[((size_t)(d * 96 + 1 * 32 + i)) * GRU_H + j]);
#pragma unroll
  for (int i = 0; i < GRU_H / 2; ++i)
    tn2[i] = mk2(w_hh[((size_t)(d * 96 + 2 * 32 + 2 * i)) * GRU_H + j], w_hh[((size_t)(d * 96 + 2 * 32 + 2 * i + 1)) * GRU_H + j]);
  const int dbg = GRU_DBG();      // (0 in a release build)
  const int T = g.T;
  if (dbg & 8) {      // (the loads above must stay alive)
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < GRU_H; ++i) acc += trz[i].x + trz[i].y + (i < GRU_H / 2 ? tn2[i].x + tn2[i].y : 0.f);
    if (acc == 12345.678f) dgi[0] = acc;
    return;
  }
  float dh_carry = 0.f;
  // steps run from the direction's LAST step to its first; dpix = pixel step in that order (the previous state sits one step further)
  const int dpix = d == 0 ? -g.stride : g.stride;
  int pix = g.base + (d == 0 ? (T - 1) * g.stride : 0);
  int fpix = pix, fstep = T - 1;
  // operands of one step: they do not depend on the recurrence, so they are fetched PF steps ahead through a small register ring (next
  // to the weight-gradient GEMMs of the side stream a load takes several times its idle latency)
  struct StepIn {
    float hprev, r, z, n, an, dho, dho2;
  };
  auto fetch = [&]() __attribute__((always_inline)) {
    StepIn s;
    s.hprev = 0.f;
    s.dho2 = 0.f;
    if (dbg & 1) {
      s.r = 0.4f; s.z = 0.6f; s.n = 0.1f; s.an = 0.2f; s.dho = 0.3f;
      fpix += dpix;
      --fstep;
      return s;
    }
    if (fstep > 0) s.hprev = h_out[(fpix + dpix) * 64 + d * 32 + j];
    const float* p = gates + fpix * 256 + d * 128 + j;
    s.r = p[0]; s.z = p[32]; s.n = p[64]; s.an = p[96];
    s.dho = dh_out[fpix * 64 + d * 32 + j];
    // (added at the step that consumes it: `s.dho += ...` here made every step wait for ALL its outstanding loads and stores --
    //  one full memory round trip per time step in every launch with a second gradient, the look-ahead ring notwithstanding)
    if (dh_out2) s.dho2 = dh_out2[fpix * 64 + d * 32 + j];
    fpix += dpix;
    --fstep;
    return s;
  };
  StepIn ring[PF];
#pragma unroll
  for (int i = 0; i < PF; ++i) ring[i] = (EXACT || i < T) ? fetch() : StepIn{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int base = T - 1; base >= 0; base -= PF) {
    const bool more = base - PF >= 0;
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int step = base - i;                 // `step` = position in the direction's own forward order
      if (!EXACT && step < 0) break;             // wave-uniform
      const StepIn c = ring[i];
      if (EXACT ? more : step - PF >= 0) ring[i] = fetch();
      const float dh = dh_carry + (c.dho + c.dho2);
      const float dn_pre = dh * (1.f - c.z) * (1.f - c.n * c.n);
      const float dz_pre = dh * (c.hprev - c.n) * c.z * (1.f - c.z);
      const float dr_pre = dn_pre * c.an * c.r * (1.f - c.r);
      const float dghn = dn_pre * c.r;
      const int par = i & 1;                     // (PF is even; only the alternation matters)
      *reinterpret_cast<float2*>(&g_rz[par][d][2 * j]) = make_float2(dr_pre, dz_pre);
      g_n[par][lane] = dghn;
      if (!(dbg & 2)) {
        float* q = dgi + pix * 192 + d * 96 + j;
        q[0] = dr_pre; q[32] = dz_pre; q[64] = dn_pre;
        if (COMPACT) {
          dgh[pix * 64 + d * 32 + j] = dghn;
        } else {
          float* q2 = dgh + pix * 192 + d * 96 + j;
          q2[0] = dr_pre; q2[32] = dz_pre; q2[64] = dghn;
        }
      }
      pix += dpix;
      if (dbg & 4) {
        dh_carry = dh * c.z + dghn;
        continue;
      }
      __builtin_amdgcn_wave_barrier();   // one wave: its LDS operations execute in order; the parity double buffer is kept anyway
      const float4* prz = reinterpret_cast<const float4*>(&g_rz[par][d][0]);
      const float4* pn = reinterpret_cast<const float4*>(&g_n[par][d * 32]);
      f2 c0 = mk2(0.f, 0.f), c1 = c0, c2 = c0, c3 = c0, e0 = c0, e1 = c0;
#pragma unroll
      for (int k = 0; k < GRU_H / 4; ++k) {
        const float4 a = prz[2 * k], b = prz[2 * k + 1], e = pn[k];   // (dr,dz) of units 4k..4k+3; dghn of units 4k..4k+3
        c0 = pk_fma(trz[4 * k], mk2(a.x, a.y), c0);
        c1 = pk_fma(trz[4 * k + 1], mk2(a.z, a.w), c1);
        c2 = pk_fma(trz[4 * k + 2], mk2(b.x, b.y), c2);
        c3 = pk_fma(trz[4 * k + 3], mk2(b.z, b.w), c3);
        e0 = pk_fma(tn2[2 * k], mk2(e.x, e.y), e0);
        e1 = pk_fma(tn2[2 * k + 1], mk2(e.z, e.w), e1);
      }
      const f2 sum = ((c0 + c1) + (c2 + c3)) + (e0 + e1);
      dh_carry = dh * c.z + (sum.x + sum.y);
    }
  }
}

template <bool COMPACT>
static int bigru_bwd_launch(const float* gates, const float* h_out, const float* dh_out, const float* dh_out2, const float* w_hh, int N,
                            int H, int W, int axis, float* dgi, float* dgh, void* stream, const char* who) {
  TPGSR_CHECK_ARG(gates && h_out && dh_out && w_hh && dgi && dgh, "%s: null pointer", who);
  TPGSR_CHECK_ARG(N > 0 && H > 0 && W > 0 && (axis == 0 || axis == 1), "%s: bad geometry", who);
  TPGSR_CHECK_ARG((long long)N * H * W * 256 < (1ll << 31), "%s: map too large for the kernel's 32-bit indices", who);
  int nseq = axis == 0 ? N * H : N * W;
  const int T = axis == 0 ? W : H;
#define GRU_BWD_CASE(PF)                                                                                                          \
  if (T % PF == 0)                                                                                                                \
    hipLaunchKernelGGL((bigru_bwd_kernel<PF, COMPACT, true>), dim3(nseq), dim3(64), 0, (hipStream_t)stream, gates, h_out, dh_out, dh_out2, w_hh, N, H, W, axis, dgi, dgh); \
  else                                                                                                                            \
    hipLaunchKernelGGL((bigru_bwd_kernel<PF, COMPACT, false>), dim3(nseq), dim3(64), 0, (hipStream_t)stream, gates, h_out, dh_out, dh_out2, w_hh, N, H, W, axis, dgi, dgh);
  switch (g_gru_pf) {
    case 4: GRU_BWD_CASE(4) break;
    case 12: GRU_BWD_CASE(12) break;
    default: GRU_BWD_CASE(8) break;
  }
#undef GRU_BWD_CASE
  TPGSR_LAUNCH_CHECK(who);
}

extern "C" int tpgsr_bigru_bwd(const float* gates, const float* h_out, const float* dh_out, const float* dh_out2,
                                const float* w_hh, int N, int H, int W, int axis, float* dgi, float* dgh, void* stream) {
  return bigru_bwd_launch<false>(gates, h_out, dh_out, dh_out2, w_hh, N, H, W, axis, dgi, dgh, stream, "tpgsr_bigru_bwd");
}

/* as tpgsr_bigru_bwd, but the hidden-side gradient is written compactly: dghn [P][64] = dn_pre * r of both directions (its r / z
 * planes equal dgi's) -- 2/3 fewer bytes written here and read by the weight gradients (tpgsr_gru_wgrad) */
extern "C" int tpgsr_bigru_bwd2(const float* gates, const float* h_out, const float* dh_out, const float* dh_out2,
                                 const float* w_hh, int N, int H, int W, int axis, float* dgi, float* dghn, void* stream) {
  return bigru_bwd_launch<true>(gates, h_out, dh_out, dh_out2, w_hh, N, H, W, axis, dgi, dghn, stream, "tpgsr_bigru_bwd2");
}
