// Device code shared by the BiGRU kernels (gru.hip: the scans over precomputed input projections and back-propagation through time;
// gru_proj.hip: the forward scan with the GruBlock's input projection computed in the same launch): sequence geometry, the gate
// functions of the recurrence, packed-FMA helpers.  GruBlock: model/tsrn.py:491-508.
#pragma once
#include "common.h"

#define GRU_H 32
// one wavefront per workgroup: the per-step barrier degenerates to wave-local ordering

struct SeqGeom {
  int base;    // pixel index of t = 0 (32-bit: the launchers bound N H W 256 by 2^31 -- 64-bit multiplies were ~20 instructions of a step)
  int stride;  // pixel stride between time steps
  int T;
  bool active;
};

__device__ __forceinline__ SeqGeom seq_geom(int s, int N, int H, int W, int axis) {
  SeqGeom g;
  int nseq = axis == 0 ? N * H : N * W;
  g.active = s < nseq;
  if (!g.active) s = 0;
  if (axis == 0) {
    g.base = s * W;
    g.stride = 1;
    g.T = W;
  } else {
    int n = s / W, col = s - n * W;
    g.base = n * H * W + col;
    g.stride = W;
    g.T = H;
  }
  return g;
}

// Gate functions of the recurrence.  A time step is ONE dependent instruction stream per wave (a step of the W-axis scan runs with at most
// one wave per SIMD), so its length in instructions is its latency: libm's expf + expm1f + three IEEE divisions were ~110 of the ~190
// instructions of a step.  These keep libm-level accuracy in a third of that:
//   e^x   = v_exp_f32(t) * (1 + ln2 * lo),  t = fl(x log2e), lo = the exact rounding error of t + x * (log2e - fl(log2e))   (6 instructions;
//           the bare v_exp_f32(x * log2e) loses |x| * 6e-8 relative -- common.h's note on what that did to the text-prior gradient)
//   1 / d = v_rcp_f32 + one Newton step (3 instructions, <= 1 ulp)
//   tanh  = x * P(x^2) for |x| < 0.35 (odd Taylor polynomial to x^11: 5e-9 relative), (1 - q) / (1 + q) with q = e^(-2|x|) elsewhere
// Checked against fp64 over the gates' range by tests/test_gru_gate_math_gpu.py: worst case 3.5 ulp (sigmoid) / 4.5 ulp (tanh, where
// 1 - q cancels one bit), against 2 ulp of the libm path; mean error 0.4 ulp either way.
__device__ __forceinline__ float gru_exp(float x) {
  const float t = x * 1.44269504088896341f;
  float lo = __builtin_fmaf(x, 1.44269504088896341f, -t);
  lo = __builtin_fmaf(x, 1.925963033500011e-08f, lo);
  const float e = __builtin_amdgcn_exp2f(t);
  return __builtin_fmaf(e, lo * 0.6931471805599453f, e);
}
__device__ __forceinline__ float gru_rcp(float d) {      // d finite, |d| in [2^-126, 2^126]
  const float r = __builtin_amdgcn_rcpf(d);
  return __builtin_fmaf(__builtin_fmaf(-d, r, 1.f), r, r);
}
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 mk2(float x, float y) {
  f2 v;
  v.x = x;
  v.y = y;
  return v;
}

// (v, v) for a packed multiply-add when v is an ODD component (.y / .w) of a float4 an LDS read has just returned.  mk2(v, v) compiles to
// v_pk_fma_f32 ... op_sel:[0,1,0] on the returned register pair -- the LOW half of the instruction takes the pair's ODD register -- and
// that form read the register as ZERO in lanes 48-63 (while the high half of the same instruction read it correctly) once in ~10^4
// time steps when three scanning waves shared a SIMD: measured term by term in round 6 (profiles/r06_gru_proj_root_cause.md; every other
// operand form of the scans -- even components with op_sel_hi, natural (x, y) pairs, a component copied by v_mov -- never failed in
// 10^6 sequence launches).  The copy costs one v_mov_b32 per use; results are the same bits.
__device__ __forceinline__ f2 gru_dup_odd(float v) {
  float c;
  asm("v_mov_b32 %0, %1" : "=v"(c) : "v"(v));
  return mk2(c, c);
}

// TWO sigmoids in lock step (the r and z gates of a time step), component for component the operations of gru_rcp(1 + gru_exp(min(-x, 80)))
// as packed instructions (v_pk_mul / v_pk_fma / v_pk_add; v_exp and v_rcp back to back).  A lone wave pays ~10 cycles per DEPENDENT VALU
// instruction and 3-5 per independent one (tools/lab/valu_rate.hip): the two sigmoids one after the other -- the compiler even scheduled the
// second one behind the tanh that needs only the first -- were ~250 cycles of a ~1350-cycle step; in lock step they are one chain.
__device__ __forceinline__ f2 gru_sigmoid2(f2 x) {
  const f2 nx = mk2(fminf(-x.x, 80.f), fminf(-x.y, 80.f));
  const f2 l2e = mk2(1.44269504088896341f, 1.44269504088896341f);
  const f2 t = nx * l2e;
  f2 lo = pk_fma(nx, l2e, -t);
  lo = pk_fma(nx, mk2(1.925963033500011e-08f, 1.925963033500011e-08f), lo);
  const f2 e = mk2(__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y));
  const f2 d = mk2(1.f, 1.f) + pk_fma(e, lo * mk2(0.6931471805599453f, 0.6931471805599453f), e);
  const f2 r = mk2(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y));
  return pk_fma(pk_fma(-d, r, mk2(1.f, 1.f)), r, r);
}
__device__ __forceinline__ float gru_sigmoid(float x) { return gru_sigmoid2(mk2(x, x)).x; }
// one sigmoid, the same operations as a component of gru_sigmoid2 (the LSTM cells of lstm_seq.hip / crnn.hip: four gates per unit)
__device__ __forceinline__ float gru_sigmoid1(float x) { return gru_rcp(1.f + gru_exp(fminf(-x, 80.f))); }
// BRANCH-FREE: both forms are computed and one is selected.  Left to itself the compiler sinks them into the two sides of a divergent
// branch (s_cbranch_execz): a scheduling barrier in the middle of the step, with both sides executed by every wave anyway (the lanes of a
// wave are 64 different hidden units).  The empty asm pins both values in front of the select.
__device__ __forceinline__ float gru_tanh(float x) {
  const float q = gru_exp(-2.f * fabsf(x));
  float big = (1.f - q) * gru_rcp(1.f + q);
  const float x2 = x * x;
  float p = __builtin_fmaf(x2, -1382.f / 155925.f, 62.f / 2835.f);
  p = __builtin_fmaf(x2, p, -17.f / 315.f);
  p = __builtin_fmaf(x2, p, 2.f / 15.f);
  p = __builtin_fmaf(x2, p, -1.f / 3.f);
  p = __builtin_fmaf(x2, p, 1.f);
  float small = x * p;
  asm volatile("" : "+v"(small), "+v"(big));
  return fabsf(x) < 0.35f ? small : copysignf(big, x);
}
