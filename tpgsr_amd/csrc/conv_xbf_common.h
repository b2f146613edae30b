// Shared device code of the split-operand bf16 MFMA convolution kernels (conv_xbf.hip: tile loop, halo kernels, weight gradients;
// conv_panel.hip: the row-panel kernel of the 1x1 convolutions): operand splitting, the term-pair MFMA sequence and the forward epilogue.
#pragma once
#include "conv_loader.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

#define XW_ROW 192                // bytes per LDS row of the wgrad images (64 bf16 + 64 B pad): 4 consecutive rows -> disjoint 64-B bank windows
#define XW_PLANE (32 * XW_ROW)

// x = h[0] + h[1] + h[2] exactly (T = 3); h[0] = RNE bf16 (T = 1)
template <int T>
__device__ __forceinline__ void split_bf(float x, __bf16 (&h)[T]) {
  h[0] = (__bf16)x;
  if (T > 1) {
    float r = x - (float)h[0];
    h[1] = (__bf16)r;
    if (T > 2) h[2] = (__bf16)(r - (float)h[1]);
  }
}
template <int T>
__device__ __forceinline__ void split4(const float4& v, uint2 (&out)[T]) {
  __bf16 a[T], b[T], c[T], d[T];
  split_bf<T>(v.x, a);
  split_bf<T>(v.y, b);
  split_bf<T>(v.z, c);
  split_bf<T>(v.w, d);
#pragma unroll
  for (int t = 0; t < T; ++t) {
    bf16x4 q;
    q[0] = a[t]; q[1] = b[t]; q[2] = c[t]; q[3] = d[t];
    out[t] = __builtin_bit_cast(uint2, q);
  }
}

// acc += sum over the term pairs (i, j) with i + j <= T + 1 of a[i] * b[j], smallest magnitudes first
template <int T>
__device__ __forceinline__ floatx16 mfma_terms(const bf16x8 (&a)[T], const bf16x8 (&b)[T], floatx16 acc) {
  if (T == 3) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
  }
  if (T == 2) {   // two-term split: a1 b1 + a1 b2 + a2 b1, what is dropped is <= 3 * 2^-18 |a b| per product
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
  }
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
}

// ---- epilogue shared by the forward kernels (as conv_fwd_kernel): bias, activation, (pixel-shuffled) store, BN partial
// statistics.  The calling threads are the 256 of the four MFMA waves (tid 0..255); `red` = 4 * 64 WNB floats of LDS nobody
// reads any more; contains one __syncthreads() when a.bn_partial is set. ----
// (-DXBF_NT_STORE=1: nontemporal output stores.  Measured: the launch alone 31.1 vs 31.8 us, the C3 step 10.17 vs 9.83 ms -- the next
//  layer finds less of its input in L2 -- so it stays off)
#ifndef XBF_NT_STORE
#define XBF_NT_STORE 0
#endif
// one element of the BatchNorm-backward epilogue (tpgsr_conv_args.bnb_y): dz = the activation backward of the gradient `raw` this launch
// produced, s += dz, ss += dz * xhat.  ONE definition with explicit fused multiply-adds for every copy of the epilogue (the
// whole-CU halo kernel has its own store path): under -ffp-contract=fast two copies of `ss += dz * (y - mu) * rs` may round differently.
__device__ __forceinline__ float xbf_bnb_elem(float raw, float y, float sc, float sh, float mu, float rs, int act, float& s, float& ss) {
  const float dz = act ? raw * act_grad(__builtin_fmaf(y, sc, sh), act) : raw;
  s += dz;
  ss = __builtin_fmaf(dz * (y - mu), rs, ss);
  return dz;
}

template <int WMB, int WNB>
__device__ __forceinline__ void xbf_store_tile(const tpgsr_conv_args& a, floatx16 (&acc)[WMB][WNB], int M, int m0, int n0, int wm,
                                               int wn, int lane, float* red) {
  constexpr int BNT = 64 * WNB;
  const int ohw = a.OH * a.OW;
#pragma unroll
  for (int j = 0; j < WNB; ++j) {
    const int cloc = wn * 32 * WNB + 32 * j + (lane & 31);
    const int n = n0 + cloc;
    const bool nvalid = n < a.Cout;
    const float bias = (a.bias && nvalid) ? a.bias[n] : 0.f;
    float s = 0.f, ss = 0.f;
    // BatchNorm-backward statistics of the gradient this launch produces (tpgsr_conv_args.bnb_y): s = sum dz, ss = sum dz * xhat;
    // bnb_store_dz: the activation backward on the way out
    const bool bnb = a.bnb_y != nullptr;
    float b_mu = 0.f, b_rs = 0.f, b_sc = 1.f, b_sh = 0.f;
    if (bnb && nvalid) {
      if (a.bn_partial) {
        b_mu = a.bnb_mean[n];
        b_rs = a.bnb_rstd[n];
      }
      if (a.bnb_act && a.bnb_scale) {
        b_sc = a.bnb_scale[n];
        b_sh = a.bnb_shift[n];
      }
    }
#pragma unroll
    for (int i = 0; i < WMB; ++i) {
      float yv[16];
      if (bnb) {   // all sixteen loads of the BatchNorm input in flight before the first store (the stores may alias for the compiler)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * 32 * WMB + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          yv[r] = (m < M && nvalid) ? a.bnb_y[(size_t)m * a.Cout + n] : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        int m = m0 + wm * 32 * WMB + 32 * i + row;
        if (m < M && nvalid) {
          float raw = acc[i][j][r];
          if (bnb) {
            const float dz = xbf_bnb_elem(raw, yv[r], b_sc, b_sh, b_mu, b_rs, a.bnb_act, s, ss);
            if (a.bnb_store_dz) raw = dz;
          } else {
            s += raw;
            ss = __builtin_fmaf(raw, raw, ss);      // (explicit: every copy of this epilogue must round the same way)
          }
          float v = apply_act(raw + bias, a.out_act);
          if (!a.out_ps) {
            if (XBF_NT_STORE) __builtin_nontemporal_store(v, &a.out[(size_t)m * a.out_ld + a.out_coff + n]);
            else a.out[(size_t)m * a.out_ld + a.out_coff + n] = v;
          } else {
            int nn = m / ohw;
            int rem = m - nn * ohw;
            int oh = rem / a.OW, ow = rem - oh * a.OW;
            int cs = n >> 2, pi = (n >> 1) & 1, pj = n & 1;
            a.out[((size_t)(nn * 2 * a.OH + 2 * oh + pi) * (2 * a.OW) + 2 * ow + pj) * (a.Cout >> 2) + cs] = v;
          }
        }
      }
    }
    if (a.bn_partial) {
      s += __shfl_xor(s, 32);
      ss += __shfl_xor(ss, 32);
      if (lane < 32) {
        red[(wm * 2 + 0) * BNT + cloc] = s;
        red[(wm * 2 + 1) * BNT + cloc] = ss;
      }
    }
  }
}
// ---- the same epilogue with WIDE stores.  xbf_store_tile above issues one four-byte store per accumulator register (a lane owns one
// column of sixteen rows): 16 per 32 x 32 block, and it is the ISSUE of those stores that bounds it (1 us per block measured in the
// whole-CU halo kernel's trace, profiles/r04l_*).  Here every wave transposes its block through 4 KB of LDS of its own -- `stage`,
// 1024 floats per wave, wave-local: the LDS operations of one wave execute in order, no barrier -- and stores 4 x 16 bytes per lane.
// Same values and the same BatchNorm partial sums, bit for bit (same expressions, same summation order).  Falls back to
// xbf_store_tile for what it does not cover: pixel-shuffle stores, the BatchNorm-backward epilogue, unaligned rows.
template <int WMB, int WNB>
__device__ __forceinline__ void xbf_store_tile_wide(const tpgsr_conv_args& a, floatx16 (&acc)[WMB][WNB], int M, int m0, int n0, int wm,
                                                    int wn, int lane, float* red, float* stage) {
  const bool fast = stage != nullptr && !a.out_ps && !a.bnb_y && (a.Cout & 3) == 0 && (a.out_ld & 3) == 0 && (a.out_coff & 3) == 0 &&
                    ((uintptr_t)a.out & 15) == 0;
  if (!fast) {
    xbf_store_tile<WMB, WNB>(a, acc, M, m0, n0, wm, wn, lane, red);
    return;
  }
  constexpr int BNT = 64 * WNB;
  const int quad = lane & 7;
#pragma unroll
  for (int j = 0; j < WNB; ++j) {
    const int cloc = wn * 32 * WNB + 32 * j + (lane & 31);
    const int n = n0 + cloc;
    const bool nvalid = n < a.Cout;
    const float bias = (a.bias && nvalid) ? a.bias[n] : 0.f;
    const int nq = n0 + wn * 32 * WNB + 32 * j + quad * 4;
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int i = 0; i < WMB; ++i) {
      const int mrow0 = m0 + wm * 32 * WMB + 32 * i;
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mrow0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const float raw = acc[i][j][r];
        if (m < M && nvalid) {
          s += raw;
          ss = __builtin_fmaf(raw, raw, ss);
        }
        v[r] = raw + bias;
      }
      if (a.out_act != TPGSR_ACT_NONE) {       // (uniform)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = apply_act(v[r], a.out_act);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = v[r];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int row = 8 * k + (lane >> 3);
        const int m = mrow0 + row;
        const float4 q4 = *reinterpret_cast<const float4*>(stage + row * 32 + quad * 4);
        if (m < M && nq < a.Cout) *reinterpret_cast<float4*>(a.out + (size_t)m * a.out_ld + a.out_coff + nq) = q4;
      }
    }
    if (a.bn_partial) {
      s += __shfl_xor(s, 32);
      ss += __shfl_xor(ss, 32);
      if (lane < 32) {
        red[(wm * 2 + 0) * BNT + cloc] = s;
        red[(wm * 2 + 1) * BNT + cloc] = ss;
      }
    }
  }
}

// after a barrier: statistics per 64-pixel row block (the layout bn_finalize expects) out of the wave rows' partials
// (write_through: the rows leave as agent-scope relaxed atomic stores = `sc1` write-through -- for a launch whose LAST workgroup reads
//  every row back, tpgsr_conv_args.fin_mode: per-XCD L2s are not coherent with each other)
template <int WMB, int WNB>
__device__ __forceinline__ void xbf_bn_flush(const tpgsr_conv_args& a, int M, int n0, int mblk, int tid, const float* red,
                                             const bool write_through = false) {
  constexpr int BNT = 64 * WNB;
  // WMB = 1: the two wave rows together are the one 64-pixel block; WMB = 2: each wave row is a block of its own
  for (int e = tid; e < BNT * WMB; e += 256) {
    const int blk = e / BNT, c = e - blk * BNT;
    const long long rb64 = (long long)mblk * WMB + blk;
    if (n0 + c < a.Cout && rb64 * 64 < M) {
      float* dst = a.bn_partial + (size_t)rb64 * 2 * a.Cout;
      float v0, v1;
      if (WMB == 1) {
        v0 = red[0 * BNT + c] + red[2 * BNT + c];
        v1 = red[1 * BNT + c] + red[3 * BNT + c];
      } else {
        v0 = red[(blk * 2 + 0) * BNT + c];
        v1 = red[(blk * 2 + 1) * BNT + c];
      }
      if (write_through) {
        __hip_atomic_store(dst + n0 + c, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + a.Cout + n0 + c, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        dst[n0 + c] = v0;
        dst[a.Cout + n0 + c] = v1;
      }
    }
  }
}

// BatchNorm finalize by the last workgroup of a launch (tpgsr_conv_args.fin_mode): called by ALL `nthreads` threads (a multiple of 64, >= 256)
// of the workgroup that drew the last ticket, behind a barrier; `scr` = 33 KB of LDS nobody reads any more.  64 channels at a time:
// thread (row lane rl = tid / 16, channel quad q = tid % 16) sums rows rl, rl + RL, ... of its four channels in fp64 from 16-byte
// L1-bypassing loads (the rows were written through by other workgroups, possibly on other XCDs), the RL partial sums per channel
// are added in lane order.  The order of the additions depends on (nblk, nthreads) only: every run gives the same bits.
__device__ __forceinline__ void xbf_fin_last(const tpgsr_conv_args& a, const int M, const int tid, const int nthreads, double* scr) {
  const int C = a.Cout, nblk = (M + 63) >> 6;
  const int RL = nthreads >> 4, rl = tid >> 4, q = tid & 15;
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.bn_partial, (size_t)nblk * 2 * C);
  for (int c0 = 0; c0 < C; c0 += 64) {
    double s[4] = {0.0, 0.0, 0.0, 0.0}, ss[4] = {0.0, 0.0, 0.0, 0.0};
    const int cq = c0 + q * 4;
    if (cq < C) {
      for (int b = rl; b < nblk; b += RL) {
        const float4 u = buf_load4_sc1(rs, ((unsigned)(b * 2 + 0) * (unsigned)C + (unsigned)cq) * 4u);
        const float4 v = buf_load4_sc1(rs, ((unsigned)(b * 2 + 1) * (unsigned)C + (unsigned)cq) * 4u);
        s[0] += (double)u.x; s[1] += (double)u.y; s[2] += (double)u.z; s[3] += (double)u.w;
        ss[0] += (double)v.x; ss[1] += (double)v.y; ss[2] += (double)v.z; ss[3] += (double)v.w;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      scr[(rl * 64 + q * 4 + i) * 2 + 0] = s[i];
      scr[(rl * 64 + q * 4 + i) * 2 + 1] = ss[i];
    }
    __syncthreads();
    if (tid < 64 && c0 + tid < C) {
      const int c = c0 + tid;
      double S = 0.0, SS = 0.0;
      for (int r = 0; r < RL; ++r) {
        S += scr[(r * 64 + tid) * 2 + 0];
        SS += scr[(r * 64 + tid) * 2 + 1];
      }
      const double count = (double)a.fin_count;
      if (a.fin_mode == 1) {          // == bn_finalize_kernel (elementwise.hip)
        const double mean_raw = S / count;
        double var = SS / count - mean_raw * mean_raw;
        if (var < 0.0) var = 0.0;
        const double mean = mean_raw + (a.fin_bias ? (double)a.fin_bias[c] : 0.0);
        const double rstd = 1.0 / sqrt(var + (double)a.fin_eps);
        const double g = (double)a.fin_gamma[c];
        a.fin_scale[c] = (float)(g * rstd);
        a.fin_shift[c] = (float)((double)a.fin_beta[c] - mean * g * rstd);
        if (a.fin_mean) a.fin_mean[c] = (float)mean;
        if (a.fin_rstd) a.fin_rstd[c] = (float)rstd;
        if (a.fin_rm) {
          const double unbiased = a.fin_count > 1 ? var * count / (count - 1.0) : var;
          const double mom = (double)a.fin_momentum;
          a.fin_rm[c] = (float)((1.0 - mom) * (double)a.fin_rm[c] + mom * mean);
          a.fin_rv[c] = (float)((1.0 - mom) * (double)a.fin_rv[c] + mom * unbiased);
        }
      } else {                        // == bn_bwd_finalize_kernel
        if (a.fin_shift) a.fin_shift[c] = a.fin_accumulate ? a.fin_shift[c] + (float)SS : (float)SS;       // dgamma
        if (a.fin_mean) a.fin_mean[c] = a.fin_accumulate ? a.fin_mean[c] + (float)S : (float)S;          // dbeta
        const double rstd = (double)a.bnb_rstd[c], mu = (double)a.bnb_mean[c], g = (double)a.fin_gamma[c];
        const double mdz = S / count, mdzx = SS / count, k0 = g * rstd;
        a.fin_scale[c] = (float)k0;
        a.fin_scale[C + c] = (float)(-k0 * mdzx * rstd);
        a.fin_scale[2 * C + c] = (float)(-k0 * (mdz - mu * rstd * mdzx));
      }
    }
    __syncthreads();
  }
}
// stage: 1024 floats of LDS per wave behind `red`'s 4 x 64 WNB floats (nullptr: four-byte stores)
template <int WMB, int WNB>
__device__ __forceinline__ void xbf_epilogue(const tpgsr_conv_args& a, floatx16 (&acc)[WMB][WNB], int M, int m0, int n0, int mblk,
                                             int wm, int wn, int lane, int tid, float* red, float* stage = nullptr) {
  xbf_store_tile_wide<WMB, WNB>(a, acc, M, m0, n0, wm, wn, lane, red, stage);
  if (a.bn_partial) {
    __syncthreads();
    xbf_bn_flush<WMB, WNB>(a, M, n0, mblk, tid, red);
  }
}

