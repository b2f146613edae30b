// Shared device code of the implicit-GEMM convolution kernels (conv_mfma.hip: fp32 MFMA; conv_xbf.hip: bf16 MFMA with
// split operands): tile constants, XCD-aware tile order, output-pixel decoding, and the A-operand tile loader -- im2col
// gather by buffer loads with hardware zero fill + the fused producer prologue (BN affine, activation, residual add,
// un-PixelShuffle, channel concat).
#pragma once
#include "common.h"

#define BM 64
#define BN 64
#define KC 32
#define ALD (BM + 1)

// XCD-aware block order (MI355X: block b runs on XCD b % 8, each XCD has a private L2): give every XCD a CONTIGUOUS
// range of logical tiles so neighbouring tiles (which share halo rows / the same pixel chunk) hit the same L2.
// Bijective for any total (cdna_hip_programming.md T1).
__device__ __forceinline__ int xcd_remap(int L, int total) {
  int q = total >> 3, r = total & 7;
  int xcd = L & 7, j = L >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

struct PixelPos {
  int n, oh, ow;
  bool valid;
};

__device__ __forceinline__ PixelPos decode_pixel(const tpgsr_conv_args& a, int m, int M) {
  PixelPos p;
  p.valid = m < M;
  int mm = p.valid ? m : 0;
  int ohw = a.OH * a.OW;
  p.n = mm / ohw;
  int r = mm - p.n * ohw;
  p.oh = r / a.OW;
  p.ow = r - p.oh * a.OW;
  return p;
}

// real (stored) input width: `a.W` is the logical width, zero-dilated by in_dil_w for transposed convs
__device__ __forceinline__ int real_w(const tpgsr_conv_args& a) { return a.in_dil_w > 1 ? (a.W - 1) / a.in_dil_w + 1 : a.W; }
__device__ __forceinline__ int stride_w(const tpgsr_conv_args& a) { return a.stride_w > 1 ? a.stride_w : 1; }

// one float of the A operand: logical input element (n, ih, iw, c) after the fused prologue
__device__ __forceinline__ float load_a_scalar(const tpgsr_conv_args& a, const PixelPos& p, int k, int K) {
  if (!p.valid || k >= K) return 0.f;
  int tap = k / a.Cin;
  int c = k - tap * a.Cin;
  int kh = tap / a.KW, kw = tap - kh * a.KW;
  int ih = p.oh + kh - a.pad_h, iw = p.ow * stride_w(a) + kw - a.pad_w;
  if ((unsigned)ih >= (unsigned)a.H || (unsigned)iw >= (unsigned)a.W) return 0.f;
  int Wr = a.W;
  if (a.in_dil_w > 1) {
    if (iw % a.in_dil_w) return 0.f;
    iw /= a.in_dil_w;
    Wr = real_w(a);
  }
  float v;
  size_t pix = (size_t)(p.n * a.H + ih) * Wr + iw;
  if (a.in_b && c >= a.cin_a) return a.in_b[((size_t)p.n * Wr + iw) * a.in_b_ld + (c - a.cin_a)];
  if (!a.in_ps) {
    v = a.in[pix * a.in_ld + a.in_coff + c];
  } else {
    int C4 = a.Cin >> 2, cs = c >> 2, i = (c >> 1) & 1, j = c & 1;
    v = a.in[((size_t)(p.n * 2 * a.H + 2 * ih + i) * (2 * a.W) + 2 * iw + j) * C4 + cs];
  }
  if (a.in_scale) v = v * a.in_scale[c] + a.in_shift[c];
  v = apply_act(v, a.in_act);
  if (a.in2) v += a.in2[pix * a.in2_ld + c];
  return v;
}

// 4 consecutive columns of a row-major [rows][ld] operand; vec: 16-byte aligned full quads (branch-free)
__device__ __forceinline__ float4 load_row4(const float* base, size_t row, int ld, int col, int ncols, bool rowvalid,
                                            bool vec) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (vec) {
    const bool ok = rowvalid && col < ncols;
    const float* p = base + (ok ? row * (size_t)ld + col : 0);
    float4 t = *reinterpret_cast<const float4*>(p);
    v.x = ok ? t.x : 0.f;
    v.y = ok ? t.y : 0.f;
    v.z = ok ? t.z : 0.f;
    v.w = ok ? t.w : 0.f;
    return v;
  }
  if (!rowvalid || col >= ncols) return v;
  const float* p = base + row * (size_t)ld + col;
  v.x = p[0];
  if (col + 1 < ncols) v.y = p[1];
  if (col + 2 < ncols) v.z = p[2];
  if (col + 3 < ncols) v.w = p[3];
  return v;
}

// ---- buffer-resource loads: out-of-range offsets return 0 in hardware, so padding / ragged edges need no data-side
// select and the loads can stay in flight across the MFMA loop (the prologue is applied when the tile is stored) ----
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, size_t nfloats) {
  size_t bytes = nfloats * 4;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, bytes > 0x7fffffffull ? 0x7fffffff : (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned off_bytes) {
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off_bytes, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
// (sc1: served by the L2 / fabric, never by this CU's vector L1 -- data another workgroup wrote through during this launch)
__device__ __forceinline__ float4 buf_load4_sc1(__amdgpu_buffer_rsrc_t r, unsigned off_bytes) {
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off_bytes, 0, 16);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, unsigned off_bytes) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)off_bytes, 0, 0));
}
#define OOB_OFF 0x7ffffff0u

struct ARaw {
  float4 v, v2;
  bool ok, raw;   // raw: value comes from the concatenated strip (no affine / activation / residual)
};

// issue the loads of one A quad (four consecutive k = one (tap, 4-channel) group; needs Cin % 4 == 0), no dependent
// arithmetic.  LD bits: 1 = per-channel affine, 2 = activation (a.in_act), 4 = residual add (in2), 8 = un-PixelShuffle
// gather, 16 = concatenated strip
// position of one thread's A quad inside the K = (kh, kw, c) index space; advanced incrementally from chunk to chunk
// (the per-chunk integer divisions were ~100 VALU instructions per wave per chunk next to 16 MFMAs)
struct KPos {
  int kh, kw, c;
};
__device__ __forceinline__ KPos kpos_init(const tpgsr_conv_args& a, int kq) {   // kq: quad index, K / 4 ordering
  const int cin4 = a.Cin >> 2;
  int tap = kq / cin4;
  KPos k;
  k.c = (kq - tap * cin4) * 4;
  k.kh = tap / a.KW;
  k.kw = tap - k.kh * a.KW;
  return k;
}
__device__ __forceinline__ void kpos_advance(const tpgsr_conv_args& a, KPos& k, int dk) {
  k.c += dk;
  while (k.c >= a.Cin) {
    k.c -= a.Cin;
    if (++k.kw == a.KW) {
      k.kw = 0;
      ++k.kh;
    }
  }
}

// branch-free form of kpos_advance for a fixed step (all quantities wave-uniform): the bf16 kernels keep loads in flight
// across iterations, and control flow inside the loop makes the compiler's s_waitcnt placement conservative
struct KStep {
  int dr, dqw, dqh;   // dk = (dqh * KW + dqw) * Cin + dr
};
__device__ __forceinline__ KStep kstep_init(const tpgsr_conv_args& a, int dk) {
  KStep s;
  const int dq = dk / a.Cin;
  s.dr = dk - dq * a.Cin;
  s.dqh = dq / a.KW;
  s.dqw = dq - s.dqh * a.KW;
  return s;
}
__device__ __forceinline__ void kpos_advance(const tpgsr_conv_args& a, KPos& k, const KStep& s) {
  k.c += s.dr;
  const bool cc = k.c >= a.Cin;
  k.c -= cc ? a.Cin : 0;
  k.kw += s.dqw + (cc ? 1 : 0);          // < 2 KW
  const bool cw = k.kw >= a.KW;
  k.kw -= cw ? a.KW : 0;
  k.kh += s.dqh + (cw ? 1 : 0);
}

template <int LD>
__device__ __forceinline__ ARaw load_a_raw(const tpgsr_conv_args& a, __amdgpu_buffer_rsrc_t rin, __amdgpu_buffer_rsrc_t rin2,
                                           const PixelPos& p, const KPos& kp) {
  ARaw r;
  const int kh = kp.kh, kw = kp.kw, c = kp.c;
  int ih = p.oh + kh - a.pad_h, iw = p.ow * stride_w(a) + kw - a.pad_w;
  r.ok = p.valid && kh < a.KH && c < a.Cin && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
  int Wr = a.W;
  if (a.in_dil_w > 1) {
    r.ok = r.ok && (iw % a.in_dil_w) == 0;
    iw /= a.in_dil_w;
    Wr = real_w(a);
  }
  const unsigned pix = (unsigned)((p.n * a.H + ih) * Wr + iw);
  if ((LD & 16) && c >= a.cin_a) {   // concatenated second source: an [N][W][Cb] strip broadcast over H, no prologue
    r.v = buf_load4(rin2, r.ok ? (((unsigned)p.n * (unsigned)Wr + (unsigned)iw) * (unsigned)a.in_b_ld + (unsigned)(c - a.cin_a)) * 4u : OOB_OFF);
    r.v2 = make_float4(0.f, 0.f, 0.f, 0.f);
    r.raw = true;
    return r;
  }
  r.raw = false;
  if (!(LD & 8)) {
    r.v = buf_load4(rin, r.ok ? (pix * (unsigned)a.in_ld + (unsigned)(a.in_coff + c)) * 4u : OOB_OFF);
  } else {
    unsigned C4 = (unsigned)a.Cin >> 2, cs = (unsigned)c >> 2, W2 = 2u * (unsigned)a.W;
    unsigned b = r.ok ? ((((unsigned)p.n * 2u * a.H + 2u * ih) * W2 + 2u * iw) * C4 + cs) * 4u : OOB_OFF;
    r.v.x = buf_load1(rin, b);
    r.v.y = buf_load1(rin, r.ok ? b + C4 * 4u : OOB_OFF);
    r.v.z = buf_load1(rin, r.ok ? b + W2 * C4 * 4u : OOB_OFF);
    r.v.w = buf_load1(rin, r.ok ? b + (W2 * C4 + C4) * 4u : OOB_OFF);
  }
  if (LD & 4) r.v2 = buf_load4(rin2, r.ok ? (pix * (unsigned)a.in2_ld + (unsigned)c) * 4u : OOB_OFF);
  return r;
}

// apply the fused prologue to a landed quad (called right before the LDS store)
// LD bit 32 (the whole-CU halo kernel only): the residual operand is scaled per channel, a = in * s + t + in2 * s2 -- the BatchNorm-backward
// apply dy = c0 dz + c1 y + c2 (in = dz, in2 = y) folded into the loader of the data-gradient convolution that consumes dy
template <int LD>
__device__ __forceinline__ float4 finish_a(const tpgsr_conv_args& a, const ARaw& r, const float4& s, const float4& t,
                                           const float4& s2 = make_float4(1.f, 1.f, 1.f, 1.f)) {
  float4 v = r.v;
  if ((LD & 16) && r.raw) return v;   // hardware zero fill already handled padding
  if (LD & 1) {
    v.x = v.x * s.x + t.x;
    v.y = v.y * s.y + t.y;
    v.z = v.z * s.z + t.z;
    v.w = v.w * s.w + t.w;
  }
  if (LD & 2) {
    v.x = apply_act(v.x, a.in_act);
    v.y = apply_act(v.y, a.in_act);
    v.z = apply_act(v.z, a.in_act);
    v.w = apply_act(v.w, a.in_act);
  }
  if (LD & 32) {
    v.x = __builtin_fmaf(r.v2.x, s2.x, v.x);
    v.y = __builtin_fmaf(r.v2.y, s2.y, v.y);
    v.z = __builtin_fmaf(r.v2.z, s2.z, v.z);
    v.w = __builtin_fmaf(r.v2.w, s2.w, v.w);
  } else if (LD & 4) {
    v.x += r.v2.x;
    v.y += r.v2.y;
    v.z += r.v2.z;
    v.w += r.v2.w;
  }
  if (LD & 3) {  // affine / activation turn the hardware zero fill into f(0): re-zero padding explicitly
    v.x = r.ok ? v.x : 0.f;
    v.y = r.ok ? v.y : 0.f;
    v.z = r.ok ? v.z : 0.f;
    v.w = r.ok ? v.w : 0.f;
  }
  return v;
}

