// Implicit-GEMM convolution / linear layers on the gfx950 bf16 matrix cores with SPLIT fp32 operands.
//
// gfx950 has no reduced-precision fast path for fp32 inputs (no xf32): v_mfma_f32_32x32x2_f32 runs at the fp32 vector rate,
// 1/16 of v_mfma_f32_32x32x16_bf16.  An fp32 value is EXACTLY the sum of three bf16 values (24 significand bits = 3 x 8):
//     x = x1 + x2 + x3,   x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)          (both subtractions are exact)
// so  a*b = a1b1 + (a1b2 + a2b1) + (a1b3 + a2b2 + a3b1) + O(2^-26 |ab|): six bf16 MFMAs (exact products, fp32 accumulate)
// reproduce the fp32 product to better than one fp32 rounding, at 16/6 = 2.7x the fp32-MFMA rate.  TERMS = 3 is that
// fp32-equivalent mode (the default numerics of this library: every parity test holds at the fp32 tolerances);
// TERMS = 1 is plain bf16 operands with fp32 accumulation (BASELINE.json's "bf16" configurations; opt-in).
//
//   fwd / dgrad :  out[m][n]  = sum_k A[m][k] * W[k][n]      tile 64 pixels x 64 channels, 4 waves x one 32x32 accumulator
//   wgrad       :  part[k][n] = sum_m A[m][k] * dy[m][n]     tile 64 k x 64 channels, split over m
//
// A is gathered by the same loader as the fp32 kernels (conv_loader.h: im2col by buffer loads with hardware zero fill, the
// producer's BN affine / activation / residual / un-PixelShuffle / concat applied in registers), split into bf16 terms when
// the tile is stored to LDS.  W comes pre-split from tpgsr_split_bf_program (once per step, next to the pack program):
// planes [TERMS][n][Kp] with k contiguous, so a weight tile is TERMS 16-byte loads per thread.
//   fwd LDS image: per term [64 rows][32 k] bf16, rows padded to 80 B -> ds_read_b128 fragments (8 consecutive k), conflict-free.
//   wgrad LDS image: per term [32 pixels][64 k or n] bf16 in the natural pixel-major order (ds_write_b64 of 4 channels),
//   rows padded to 192 B; the contraction runs over pixels, so fragments are fetched with the transposing
//   ds_read_b64_tr_b16 (4 consecutive pixels of one channel per lane and instruction).
#include "conv_loader.h"
#include <stdlib.h>
#include <stdio.h>
#include <type_traits>
#include <utility>
#include <vector>
#include <mutex>

#include "conv_xbf_common.h"

// ------------------------------------------------------------------------------------------------------
// forward / data-gradient.  Workgroup = 4 waves in a 2 x 2 arrangement; every wave owns WMB x WNB accumulator blocks of
// 32 x 32, so the workgroup tile is (64 WMB) pixels x (64 WNB) channels.  K chunks of 32 (two MFMA k-blocks of 16).
//   A (activations, split at run time): through a DOUBLE-buffered LDS image with ONE barrier per chunk -- while the matrix pipe
//     works on chunk c out of buffer c & 1, the same wave splits chunk c + 1 (in registers: its global loads were issued one
//     iteration earlier) into the other buffer and issues the loads of chunk c + 2.  Image per term: [rows][32 k] bf16 =
//     64-byte rows, 16-byte slot s of row r stored at slot s ^ ((r >> 2) & 3): conflict-free ds_read_b128 fragments with no
//     padding.
//   W (weights, split once per step by tpgsr_split_bf_program): NEVER touches LDS.  The planes are stored in MFMA fragment
//     order [term][n / 32][k / 16][lane][8], so a wave's B operand of one k-block is ONE fully coalesced 1 KB load straight into
//     the registers the MFMA reads.
//   Why the blocks per wave matter (x3 mode, per 32x32x16 MFMA = 32 cycles of one SIMD, i.e. one MFMA per 8 clk per CU at
//     peak): a 1 x 1 wave tile needs 3 A + 3 W fragments (6 KB) per 6 MFMAs = 512 B of LDS reads AND 512 B of L1 reads per
//     MFMA -- 50 % of the LDS pipe (128 B/clk) and 100 % of the vector L1 (64 B/clk) at matrix peak, which is where the
//     1 x 1 kernel sat (31 % matrix-pipe busy, LDS 46 %: rocprofv3 PMC, profiles/r02_pmc_conv.md).  2 x 1 halves the W
//     bytes per MFMA, 2 x 2 halves both.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int xa_off(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4); }

// register budget: 1 x 1 wave tiles must keep three workgroups per CU resident (the trunk convolutions launch exactly
// three per CU), i.e. <= 168 VGPRs + AGPRs
// SK = 1 (split-K, tpgsr_conv_args.sk_splits; 1 x 1 wave tiles only): the grid is sk_splits x tiles, workgroup (z, tile) walks chunks
// [z cps, (z + 1) cps) and leaves its raw accumulators in sk_part; conv_splitk_reduce_kernel below adds them and runs the epilogue
template <int LD, int T, int WMB, int WNB, int SK = 0>
__global__ __launch_bounds__(256, (WMB * WNB == 1 ? 3 : WMB * WNB == 2 ? 2 : 1)) void conv_fwd_xbf_kernel(tpgsr_conv_args a, int M, int K) {
  constexpr int BMT = 64 * WMB, BNT = 64 * WNB;
  constexpr int A_PLANE = BMT * 64;               // bytes per term
  constexpr int BUF = T * A_PLANE;
  constexpr int NQ = BMT / 32;                    // A quads (4 consecutive k of one pixel) per thread and chunk
  constexpr bool TWO = (T == 3) && (WMB * WNB <= 2);   // separate accumulator for the correction terms (registers permitting)
  constexpr int EPI = (4 * 64 * WNB + 4 * 1024) * 4;      // the epilogue's scratch: BatchNorm partials + 4 KB of staging per wave
  __shared__ __attribute__((aligned(16))) unsigned char xsm[2 * BUF > EPI ? 2 * BUF : EPI];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int nbn = (a.Cout + BNT - 1) / BNT;
  const int ntile = SK ? (int)gridDim.x / a.sk_splits : (int)gridDim.x;
  const int skz = SK ? (int)blockIdx.x / ntile : 0;
  const int tile = xcd_remap(SK ? (int)blockIdx.x - skz * ntile : (int)blockIdx.x, ntile);
  const int mblk = tile / nbn;
  const int m0 = mblk * BMT, n0 = (tile - mblk * nbn) * BNT;
  const int nchunks_all = a.kp / KC;
  const int cps = SK ? (nchunks_all + a.sk_splits - 1) / a.sk_splits : nchunks_all;     // chunks per split
  const int c0 = skz * cps;                                                              // first chunk of this workgroup
  const int nchunks = SK ? min(cps, nchunks_all - c0) : nchunks_all;

  // A staging: quad (tid & 7) = 16 of a row's 64 bytes -> half of slot (tid & 7) >> 1; pixels (tid >> 3) + 32 i
  const int aq = tid & 7;
  const int am0 = tid >> 3;
  PixelPos px[NQ];
  int wofs[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    px[i] = decode_pixel(a, m0 + am0 + 32 * i, M);
    wofs[i] = xa_off(am0 + 32 * i, aq >> 1) + (aq & 1) * 8;
  }

  const int Wr_ = real_w(a);
  const size_t in_floats = a.in_ps ? (size_t)a.N * a.H * a.W * a.Cin : (size_t)a.N * a.H * Wr_ * a.in_ld;
  const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(a.in, in_floats);
  const __amdgpu_buffer_rsrc_t rs_in2 = (LD & 16) ? make_rsrc(a.in_b, (size_t)a.N * Wr_ * a.in_b_ld)
                                                  : make_rsrc(a.in2 ? a.in2 : a.in, (size_t)a.N * a.H * Wr_ * a.in2_ld);
  // W fragments: plane t, 32-column block nb, k-block kb16 at ((t * NB32 + nb) * KB16 + kb16) * 1024 bytes (+ lane * 16)
  const int wrows = a.wt_ld > 0 ? a.wt_ld : a.Cout;
  const int NB32 = (wrows + 31) >> 5, KB16 = a.kp >> 4;
  const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(reinterpret_cast<const float*>(a.wt_bf), (size_t)T * NB32 * 32 * a.kp / 2);
  const int ncol0 = n0 + wn * 32 * WNB;             // first column of this wave
  const unsigned plane_w = (unsigned)NB32 * KB16 * 1024u;
  unsigned woff[WNB];
#pragma unroll
  for (int j = 0; j < WNB; ++j)   // a column block entirely past Cout: zeros (hardware zero fill of the out-of-range offset)
    woff[j] = ncol0 + 32 * j < a.Cout ? ((unsigned)((a.wt_coff + ncol0 + 32 * j) >> 5) * KB16) * 1024u + lane * 16u : OOB_OFF;

  ARaw qa[NQ];
  float4 qs = make_float4(1.f, 1.f, 1.f, 1.f), qt = make_float4(0.f, 0.f, 0.f, 0.f);
  // K order of the pre-split weights: natural (tap, ci), or -- a.wt_bf_cin > 0 -- channel blocks of 32 outermost:
  // k' = ((ci / 32) * KH KW + tap) * 32 + ci % 32 (the order the halo kernel below consumes; chunk = one tap of one block)
  const bool kperm = a.wt_bf_cin > 0;
  KPos kp_ = kperm ? KPos{0, 0, aq * 4} : kpos_init(a, aq);
  if (SK && c0 > 0) {       // start at chunk c0
    if (kperm) {
      const int taps = a.KH * a.KW, cb = c0 / taps, tp = c0 - cb * taps;
      kp_ = KPos{tp / a.KW, tp - (tp / a.KW) * a.KW, cb * 32 + aq * 4};
    } else {
      kpos_advance(a, kp_, c0 * KC);
    }
  }
  const KStep kstep = kstep_init(a, KC);
  auto load_chunk = [&]() {          // past the last chunk: kh >= KH, every load is the hardware-zero-filled out-of-range one
#pragma unroll
    for (int i = 0; i < NQ; ++i) qa[i] = load_a_raw<LD>(a, rs_in, rs_in2, px[i], kp_);
    if (LD & 1) {
      qs = *reinterpret_cast<const float4*>(a.in_scale + ((kp_.kh < a.KH && kp_.c < a.Cin) ? kp_.c : 0));
      qt = *reinterpret_cast<const float4*>(a.in_shift + ((kp_.kh < a.KH && kp_.c < a.Cin) ? kp_.c : 0));
    }
    if (kperm) {
      const bool cw = ++kp_.kw >= a.KW;
      kp_.kw = cw ? 0 : kp_.kw;
      kp_.kh += cw ? 1 : 0;
      const bool chh = kp_.kh >= a.KH;
      kp_.kh = chh ? 0 : kp_.kh;
      kp_.c += chh ? 32 : 0;
    } else {
      kpos_advance(a, kp_, kstep);
    }
  };
  auto store_chunk = [&](unsigned char* buf) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const float4 v = finish_a<LD>(a, qa[i], qs, qt);
      uint2 h[T];
      split4<T>(v, h);
#pragma unroll
      for (int t = 0; t < T; ++t) *reinterpret_cast<uint2*>(buf + t * A_PLANE + wofs[i]) = h[t];
    }
  };

  floatx16 acc[WMB][WNB], accl[TWO ? WMB : 1][TWO ? WNB : 1];
#pragma unroll
  for (int i = 0; i < WMB; ++i)
#pragma unroll
    for (int j = 0; j < WNB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[i][j][r] = 0.f;
        if (TWO) accl[i][j][r] = 0.f;
      }

  const int g = lane >> 5;
  int aoff[WMB][2];                                  // the two k-blocks of a chunk
#pragma unroll
  for (int i = 0; i < WMB; ++i) {
    const int arow = wm * 32 * WMB + 32 * i + (lane & 31);
    aoff[i][0] = xa_off(arow, g);
    aoff[i][1] = xa_off(arow, 2 + g);
  }
  // Software pipeline, one chunk deep in REGISTERS on top of the one-chunk-deep LDS image: iteration c issues the fragment
  // loads of chunk c (W from global, A from LDS buffer c & 1) into register set c & 1 and runs the MFMAs of chunk c - 1 out
  // of the other set, so neither the L2 latency of W nor the LDS latency of A sits between a barrier and the matrix pipe.
  // The loop body is straight-line code (loads past the end are harmless: zero-filled / unused), which is what lets the
  // compiler count outstanding loads exactly instead of draining them at every join.
  bf16x8 av[2][2][WMB][T];   // [set][k-block][m-block][term]
  u32x4 bw[2][2][WNB][T];    // [set][k-block][n-block][term]
  auto fetch = [&](const int ch, auto set_tag) {
    constexpr int S = decltype(set_tag)::value;
    const unsigned char* cur = xsm + (ch & 1) * BUF;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int j = 0; j < WNB; ++j)
#pragma unroll
        for (int t = 0; t < T; ++t)
          bw[S][kb][j][t] = __builtin_amdgcn_raw_buffer_load_b128(
              rs_w, woff[j] == OOB_OFF ? (int)OOB_OFF : (int)(woff[j] + t * plane_w + (unsigned)((ch + c0) * 2 + kb) * 1024u), 0, 0);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int i = 0; i < WMB; ++i)
#pragma unroll
        for (int t = 0; t < T; ++t) av[S][kb][i][t] = *reinterpret_cast<const bf16x8*>(cur + t * A_PLANE + aoff[i][kb]);
  };
  auto multiply = [&](auto set_tag) {
    constexpr int S = decltype(set_tag)::value;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int j = 0; j < WNB; ++j) {
        bf16x8 bv[T];
#pragma unroll
        for (int t = 0; t < T; ++t) bv[t] = __builtin_bit_cast(bf16x8, bw[S][kb][j][t]);
#pragma unroll
        for (int i = 0; i < WMB; ++i) {
          const bf16x8(&a_)[T] = av[S][kb][i];
          if (T == 3) {
            floatx16& lo = TWO ? accl[TWO ? i : 0][TWO ? j : 0] : acc[i][j];
            lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[0], bv[2], lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[2], bv[0], lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[1], bv[1], lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[0], bv[1], lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[1], bv[0], lo, 0, 0, 0);
          }
          if (T == 2) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[0], bv[1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[1], bv[0], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[0], bv[0], acc[i][j], 0, 0, 0);
        }
      }
  };
  // iteration c: fetch(c) -> set c & 1; MFMAs of chunk c - 1; split chunk c + 1 into the other LDS buffer; issue the loads of c + 2
  auto iteration = [&](const int ch, auto set_tag) {
    constexpr int S = decltype(set_tag)::value;
    fetch(ch, set_tag);
    multiply(std::integral_constant<int, S ^ 1>{});
    store_chunk(xsm + ((ch + 1) & 1) * BUF);
    load_chunk();
    __syncthreads();                                 // (waits for this wave's LDS reads and writes first)
  };
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;
  load_chunk();
  store_chunk(xsm);
  load_chunk();
  __syncthreads();
  fetch(0, Set0{});                                  // iteration 0 has no MFMAs
  store_chunk(xsm + BUF);
  load_chunk();
  __syncthreads();
  int ch = 1;
  for (; ch + 1 < nchunks; ch += 2) {
    iteration(ch, Set1{});
    iteration(ch + 1, Set0{});
  }
  if (ch < nchunks) {
    iteration(ch, Set1{});
    multiply(Set1{});
  } else {
    multiply(Set0{});
  }
  if (TWO) {
#pragma unroll
    for (int i = 0; i < WMB; ++i)
#pragma unroll
      for (int j = 0; j < WNB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] += accl[TWO ? i : 0][TWO ? j : 0][r];
  }

  if constexpr (SK != 0) {     // raw accumulators, fragment order: [split][tile][thread][16]
    float4* dst = reinterpret_cast<float4*>(a.sk_part) + ((size_t)(skz * ntile + tile) * 256 + tid) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = make_float4(acc[0][0][4 * q], acc[0][0][4 * q + 1], acc[0][0][4 * q + 2], acc[0][0][4 * q + 3]);
    return;
  }
  __syncthreads();      // every wave is through with the A image: it becomes the epilogue's scratch
  xbf_epilogue<WMB, WNB>(a, acc, M, m0, n0, mblk, wm, wn, lane, tid, reinterpret_cast<float*>(xsm),
                         reinterpret_cast<float*>(xsm) + 4 * 64 * WNB + wave * 1024);
}

// second launch of a split-K convolution: workgroup = output tile; the S partial accumulators of every thread are added in split order
// (deterministic) and handed to the tile loop's own epilogue
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(tpgsr_conv_args a, int M) {
  constexpr int EPI = (4 * 64 + 4 * 1024) * 4;
  __shared__ __attribute__((aligned(16))) unsigned char xsm[EPI];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int nbn = (a.Cout + 63) / 64;
  const int ntile = (int)gridDim.x, tile = (int)blockIdx.x;
  const int mblk = tile / nbn;
  const int m0 = mblk * 64, n0 = (tile - mblk * nbn) * 64;
  floatx16 acc[1][1];
  const float4* src = reinterpret_cast<const float4*>(a.sk_part) + ((size_t)tile * 256 + tid) * 4;
  // all S x 4 loads in flight before the first addition (the planner's S <= 8; a loop over z waited for a round trip per split:
  // 8 us for 64 KB); added in split order as before
  float4 w[8][4];
#pragma unroll
  for (int z = 0; z < 8; ++z)
    if (z < a.sk_splits) {
#pragma unroll
      for (int q = 0; q < 4; ++q) w[z][q] = src[(size_t)z * ntile * 256 * 4 + q];
    }
  float4 v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = w[0][q];
#pragma unroll
  for (int z = 1; z < 8; ++z)
    if (z < a.sk_splits) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[q].x += w[z][q].x; v[q].y += w[z][q].y; v[q].z += w[z][q].z; v[q].w += w[z][q].w;
      }
    }
  for (int z = 8; z < a.sk_splits; ++z) {          // (a caller's own S > 8)
    const float4* sz = src + (size_t)z * ntile * 256 * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 u = sz[q];
      v[q].x += u.x; v[q].y += u.y; v[q].z += u.z; v[q].w += u.w;
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    acc[0][0][4 * q] = v[q].x; acc[0][0][4 * q + 1] = v[q].y; acc[0][0][4 * q + 2] = v[q].z; acc[0][0][4 * q + 3] = v[q].w;
  }
  xbf_epilogue<1, 1>(a, acc, M, m0, n0, mblk, wm, wn, lane, tid, reinterpret_cast<float*>(xsm), reinterpret_cast<float*>(xsm) + 4 * 64 + wave * 1024);
}

// loader variants instantiated for the bf16 path (the same set as the fp32 kernel)
#define XBF_LD_CASES(X) X(0) X(1) X(2) X(3) X(4) X(5) X(7) X(8) X(17)

// ------------------------------------------------------------------------------------------------------
// forward / data-gradient of KH x KW > 1 x 1 convolutions: HALO kernel.
//
// What bounds the tile loop above on the trunk / recognizer shapes is not a pipe but memory latency and the A side's
// instruction count: every tap re-loads the same input pixels from L2 / HBM (9x for a 3x3) and re-splits them into bf16 terms,
// two loads per thread in flight for ~1 us each (measured: 1.0 us per K chunk at 3 workgroups / CU, plain-bf16 and x3 alike).
// Here the K loop runs channel block (32) outermost, taps innermost, and the workgroup keeps the HALO of its 64 output pixels
// -- every input pixel any tap touches, one 32-channel block at a time -- in LDS, split once:
//   virtual padded index   q(n, r, s) = (n Hp + r) Wp + s,  Hp = OH + KH - 1, Wp = OW + KW - 1  (stride 1, zero padding virtual)
//   output pixel m = (n, oh, ow) reads, for tap (kh, kw), entry  q(n, oh, ow) + kh Wp + kw      -- a constant offset per tap,
//   so a tile of 64 consecutive m needs the contiguous range [q(m0), q(m_last) + (KH - 1) Wp + KW - 1]: L entries of
//   32 channels x T bf16 terms (198 for a 3x3 on a 64-wide map; tiles may span rows and images, the padding rows between
//   images are part of the index space).
// Roles (8 waves): waves 4..7 PRODUCE -- load halo block c + 1 (one 16-byte load per entry quad, fused prologue, split, store
// to LDS buffer (c + 1) & 1) -- while waves 0..3 run the KH KW x 2 MFMA steps of block c out of buffer c & 1, one barrier per
// channel block.  Separate waves because the vector-memory counter is in-order per wave: a wave that streams W fragments
// every step cannot also keep a 1-us halo load in flight.  Per 3x3 block: 7 loads + 7 splits per producer thread against 108
// MFMAs per consumer wave (the tile loop: 18 + 18 against 108), and the input is read from L2 / HBM about 3x instead of 9x.
// W as above: fragment-ordered planes in k' order straight into registers, one step ahead; A fragments: ds_read_b128 at
// entry (q(m) - q(m0)) + tap offset, 64-byte entries with the 16-byte slot XOR-swizzled by (entry >> 2) & 3.
// ------------------------------------------------------------------------------------------------------
// diagnostic time line (tpgsr_halo_trace): wall-clock stamps (100 MHz) of the first 8 workgroups, [workgroup][8 wave rows][256 slots];
// slot 4 j + k of item j -- producers: k = 0 loads issued, 1 split + stored, 2 past the barrier; consumers: k = 0 at the
// barrier, 1 past it, 2 MFMAs issued, 3 tile stored (last item of a tile)
__device__ unsigned long long* g_halo_trace = nullptr;
#define HALO_STAMP(slot)                                                                                     \
  do {                                                                                                       \
    if (trace && lane == 0 && (slot) < 256)                                                                  \
      __builtin_nontemporal_store((unsigned long long)wall_clock64(), trace + (blockIdx.x * 8 + wave) * 256 + (slot));                \
  } while (0)

// persistent tile t -> logical tile (m-block major).  Default: every XCD a contiguous range of tiles (neighbouring pixel tiles share
// halo rows in its L2).  Column-major per XCD (reserved0 bit 0, chosen by the launcher when the split weight planes exceed an L2):
// block b runs on XCD b % 8 and the grid is a multiple of 8, so tile t = 8 j + x is always handled on XCD x -- give XCD x the column
// tiles x nper .. (x + 1) nper - 1 (nbn = 8 nper) of ALL pixel tiles: its slice of W (1/8 of the planes) stays in L2 instead of every
// XCD streaming all of W once per pixel tile (conv5: 178 MB of memory-side reads per launch for 35 MB of operands).  For nbn = 1, 2, 4
// the plain order t = mblk nbn + nblk already pins column tile t % nbn to XCD t % 8.  Placement only affects speed.
__device__ __forceinline__ int halo_tile(int t, int ntiles, int nbn, int colmajor) {
  if (!colmajor) return xcd_remap(t, ntiles);
  if (nbn < 8) return t;
  const int nper = nbn >> 3, x = t & 7, j = t >> 3;
  return (j / nper) * nbn + x * nper + j % nper;
}

template <int LD, int T, int NE>   // NE: halo entries per producer thread (capacity 32 NE entries)
__global__ __launch_bounds__(512, 4) void conv_halo_xbf_kernel(tpgsr_conv_args a, int M, int Lcap) {
  // PERSISTENT: the grid is what the chip holds at once (two workgroups per CU for the 3x3 x3 shapes); workgroup b runs
  // tiles b, b + grid, ... and the producer / consumer pipeline below simply continues across tiles -- while the consumers
  // finish tile i (last channel block, epilogue stores) the producers are already loading and splitting tile i + 1.
  // 8 waves at <= 128 registers: two workgroups per CU.  (Tried and measured slower: 4 consumers + 2 producers at 168 registers
  // with two accumulators per wave -- the second 6-wave workgroup no longer fits next to the first on a CU, 42.8 vs 30.6 us
  // on the trunk convolution.)
  extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];   // [2 buffers][T][Lcap entries][64 B], then 2 x 1 KB `red`
  const int PLANE = Lcap * 64, BUF = T * PLANE;
  float* red_base = reinterpret_cast<float*>(hsm + 2 * BUF);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  typedef __attribute__((address_space(1))) unsigned long long* gptr_t;     // (a global, not a flat, store: a pending FLAT
  const gptr_t trace = blockIdx.x < 8 ? (gptr_t)g_halo_trace : (gptr_t) nullptr;   //  access makes every later s_waitcnt a full drain)
  const int nbn = (a.Cout + 63) >> 6;
  const int ntiles = ((M + 63) >> 6) * nbn;
  const int taps = a.KH * a.KW, NC = a.Cin >> 5;
  const int Hp = a.OH + a.KH - 1, Wp = a.OW + a.KW - 1, ohw = a.OH * a.OW;
  auto qbase = [&](int m) __attribute__((always_inline)) {
    const int n = m / ohw, r = m - n * ohw, oh = r / a.OW;
    return (n * Hp + oh) * Wp + (r - oh * a.OW);
  };

  if (wave >= 4) {
    // ------------------------------- producers -------------------------------
    const int pt = tid - 256, aq = pt & 7, er = pt >> 3;      // quad aq of entries er, er + 32, ...
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(a.in, (LD & 8) ? (size_t)a.N * a.H * a.W * a.Cin : (size_t)a.N * a.H * a.W * a.in_ld);
    const __amdgpu_buffer_rsrc_t rs_in2 = make_rsrc(a.in2 ? a.in2 : a.in, (size_t)a.N * a.H * a.W * a.in2_ld);
    int hpix[NE];          // of the tile being LOADED; input pixel of entry 32 i + er: >= 0, -1 = padding (stored as zeros), -2 = not part of the halo
    int hcol[(LD & 8) ? NE : 1];   // un-PixelShuffle gather (LD & 8): hpix holds the image row n H + ih, hcol the column iw
    auto decode_tile = [&](const int t) __attribute__((always_inline)) {
      const int mblk = halo_tile(t, ntiles, nbn, a.reserved0 & 1) / nbn;
      const int m0 = mblk * 64;
      const int q0 = qbase(m0);
      const int L = qbase(min(m0 + 63, M - 1)) - q0 + (a.KH - 1) * Wp + a.KW;     // <= Lcap (host bound)
      // entry 32 i + er <-> virtual padded position q0 + er + 32 i: decode the first, then walk in steps of 32 (Wp >= 8, checked
      // by the launcher: at most four row wraps per step)
      const int q = q0 + er;
      int n = q / (Hp * Wp);
      const int rem = q - n * (Hp * Wp);
      int r = rem / Wp, sx = rem - r * Wp;
#pragma unroll
      for (int i = 0; i < NE; ++i) {
        const int ih = r - a.pad_h, iw = sx - a.pad_w;
        const bool in = n < a.N && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
        if (LD & 8) {
          hpix[i] = 32 * i + er < L ? (in ? n * a.H + ih : -1) : -2;
          hcol[i] = iw;
        } else {
          hpix[i] = 32 * i + er < L ? (in ? (n * a.H + ih) * a.W + iw : -1) : -2;
        }
        sx += 32;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const bool c1 = sx >= Wp;
          sx -= c1 ? Wp : 0;
          r += c1 ? 1 : 0;
          const bool c2 = r >= Hp;
          r -= c2 ? Hp : 0;
          n += c2 ? 1 : 0;
        }
      }
    };
    // the loads of item j + 1 (next channel block, or block 0 of the next tile) are issued BEFORE item j is split and stored,
    // so they are in flight while this wave sits in barrier j waiting for the consumers
    constexpr bool DB = !(LD & 4);       // (the residual-add loader carries two quads per entry: one register set only)
    ARaw hr[DB ? 2 : 1][NE];             // .raw carries "this entry is part of the halo" (store mask)
    float4 qs[2], qt[2];
    qs[0] = qs[1] = make_float4(1.f, 1.f, 1.f, 1.f);
    qt[0] = qt[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_item = [&](auto set_tag, const int cc) __attribute__((always_inline)) {
      constexpr int S = decltype(set_tag)::value;
      const int c = cc * 32 + aq * 4;
#pragma unroll
      for (int i = 0; i < NE; ++i) {
        const bool ok = hpix[i] >= 0;
        hr[S][i].ok = ok;
        hr[S][i].raw = hpix[i] > -2;
        if (LD & 8) {   // logical channels c..c+3 = (cs = c / 4, i, j) of the stored [N][2H][2W][Cin / 4] tensor
          const unsigned C4 = (unsigned)a.Cin >> 2, W2 = 2u * (unsigned)a.W;
          const unsigned b = ok ? ((2u * (unsigned)hpix[i] * W2 + 2u * (unsigned)hcol[i]) * C4 + ((unsigned)c >> 2)) * 4u : OOB_OFF;
          hr[S][i].v.x = buf_load1(rs_in, b);
          hr[S][i].v.y = buf_load1(rs_in, ok ? b + C4 * 4u : OOB_OFF);
          hr[S][i].v.z = buf_load1(rs_in, ok ? b + W2 * C4 * 4u : OOB_OFF);
          hr[S][i].v.w = buf_load1(rs_in, ok ? b + (W2 * C4 + C4) * 4u : OOB_OFF);
        } else {
          hr[S][i].v = buf_load4(rs_in, ok ? ((unsigned)hpix[i] * (unsigned)a.in_ld + (unsigned)(a.in_coff + c)) * 4u : OOB_OFF);
        }
        if (LD & 4) hr[S][i].v2 = buf_load4(rs_in2, ok ? ((unsigned)hpix[i] * (unsigned)a.in2_ld + (unsigned)c) * 4u : OOB_OFF);
      }
      if (LD & 1) {
        qs[S] = *reinterpret_cast<const float4*>(a.in_scale + c);
        qt[S] = *reinterpret_cast<const float4*>(a.in_shift + c);
      }
    };
    auto store_item = [&](auto set_tag, const int j) __attribute__((always_inline)) {
      constexpr int S = decltype(set_tag)::value;
      unsigned char* buf = hsm + (j & 1) * BUF;
#pragma unroll
      for (int i = 0; i < NE; ++i) {
        const bool part = hr[S][i].raw;
        hr[S][i].raw = false;
        const float4 v = finish_a<LD>(a, hr[S][i], qs[S], qt[S]);
        uint2 h[T];
        split4<T>(v, h);
        const int e = 32 * i + er;
        const int off = e * 64 + (((aq >> 1) ^ ((e >> 2) & 3)) << 4) + (aq & 1) * 8;
        if (part) {
#pragma unroll
          for (int t = 0; t < T; ++t) *reinterpret_cast<uint2*>(buf + t * PLANE + off) = h[t];
        }
      }
    };
    int t = blockIdx.x, cc = 0, j = 0;
    using P0 = std::integral_constant<int, 0>;
    using PN = std::integral_constant<int, DB ? 1 : 0>;
    decode_tile(t);
    if (DB) load_item(P0{}, 0);
    // one item: (if double-buffered) issue the next item's loads, split + store this one, barrier; returns false after the last
    auto item = [&](auto cur_tag, auto nxt_tag) __attribute__((always_inline)) -> bool {
      int ncc = cc + 1, nt = t;
      if (ncc == NC) {
        ncc = 0;
        nt = t + gridDim.x;
      }
      const bool more = nt < ntiles;
      if (!DB) load_item(cur_tag, cc);
      if (more && ncc == 0) decode_tile(nt);       // (this item's entries are already captured in its register set)
      if (DB && more) load_item(nxt_tag, ncc);
      HALO_STAMP(4 * j);
      store_item(cur_tag, j);
      HALO_STAMP(4 * j + 1);
      __syncthreads();      // barrier j: item j is in LDS, and the consumers are done with item j - 1
      HALO_STAMP(4 * j + 2);
      ++j;
      t = nt;
      cc = ncc;
      return more;
    };
    while (true) {
      if (!item(P0{}, PN{})) break;
      if (!item(PN{}, P0{})) break;
    }
    __syncthreads();        // the final barrier (the consumers' last statistics flush)
    return;
  }

  // ------------------------------- consumers -------------------------------
  const int wm = wave & 1, wn = wave >> 1;
  const int wrows = a.wt_ld > 0 ? a.wt_ld : a.Cout;
  const int NB32 = (wrows + 31) >> 5, KB16 = a.kp >> 4;
  const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(reinterpret_cast<const float*>(a.wt_bf), (size_t)T * NB32 * 32 * a.kp / 2);
  const unsigned plane_w = (unsigned)NB32 * KB16 * 1024u;
  const int g = lane >> 5;
  unsigned woff = 0;
  int ebase = 0;

  floatx16 acc[1][1];       // ONE accumulator (correction terms first, then a1 b1, every step): the register budget is 128
  // W fragments run one TAP (two 16-k steps = 12 MFMAs = 384 cycles) ahead of the matrix pipe, in two register sets indexed
  // by tap parity; A fragments are fetched as soon as the MFMAs reading their registers have been issued (one step ahead).
  bf16x8 av[2][T];          // [k-block][term]
  u32x4 bw[2][2][T];        // [set][k-block][term]
  int gtap = 0;             // tap counter of the tile = (channel block, tap) in k' order; its W k-blocks are 2 gtap, 2 gtap + 1
  auto fetch_w = [&](auto set_tag, auto kb_tag, const int tapidx) __attribute__((always_inline)) {
    constexpr int S = decltype(set_tag)::value, KB = decltype(kb_tag)::value;
#pragma unroll
    for (int t = 0; t < T; ++t)
      bw[S][KB][t] = __builtin_amdgcn_raw_buffer_load_b128(
          rs_w, woff == OOB_OFF ? (int)OOB_OFF : (int)(woff + t * plane_w + (unsigned)(2 * tapidx + KB) * 1024u), 0, 0);
  };
  auto fetch_a = [&](auto kb_tag, const unsigned char* buf, const int tapoff) __attribute__((always_inline)) {
    constexpr int KB = decltype(kb_tag)::value;
    const int e = ebase + tapoff;
    const int off = e * 64 + (((KB * 2 + g) ^ ((e >> 2) & 3)) << 4);
#pragma unroll
    for (int t = 0; t < T; ++t) av[KB][t] = *reinterpret_cast<const bf16x8*>(buf + t * PLANE + off);
  };
  auto multiply = [&](auto set_tag, auto kb_tag) __attribute__((always_inline)) {
    constexpr int S = decltype(set_tag)::value, KB = decltype(kb_tag)::value;
    bf16x8 bv[T];
#pragma unroll
    for (int t = 0; t < T; ++t) bv[t] = __builtin_bit_cast(bf16x8, bw[S][KB][t]);
    acc[0][0] = mfma_terms<T>(av[KB], bv, acc[0][0]);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  int kw = 0, tapoff = 0;
  auto next_tap = [&]() __attribute__((always_inline)) {
    const bool cw = ++kw == a.KW;
    tapoff += cw ? Wp - a.KW + 1 : 1;
    kw = cw ? 0 : kw;
  };
  // one tap out of W set S; NEXT: another tap of this channel block follows (prefetch its A fragments).
  // The sched_barriers pin the software pipeline: without them the machine scheduler sinks the prefetches down to their first
  // use (measured: s_waitcnt vmcnt(0) at the loop head, 115 cycles per MFMA and wave instead of 32 -- every tap waited out L2).
  auto tap = [&](auto set_tag, auto next_tag, const unsigned char* buf) __attribute__((always_inline)) {
    constexpr int S = decltype(set_tag)::value;
    constexpr bool NEXT = decltype(next_tag)::value != 0;
    fetch_w(std::integral_constant<int, S ^ 1>{}, I0{}, gtap + 1);
    fetch_w(std::integral_constant<int, S ^ 1>{}, I1{}, gtap + 1);
    if (NEXT) next_tap();
    __builtin_amdgcn_sched_barrier(0);
    multiply(set_tag, I0{});
    __builtin_amdgcn_sched_barrier(0);
    if (NEXT) fetch_a(I0{}, buf, tapoff);
    __builtin_amdgcn_sched_barrier(0);
    multiply(set_tag, I1{});
    __builtin_amdgcn_sched_barrier(0);
    if (NEXT) fetch_a(I1{}, buf, tapoff);
    __builtin_amdgcn_sched_barrier(0);
    ++gtap;
  };
  // last tap of a block with an ODD tap count: the next block starts on set 0 again, which this tap is still reading --
  // its W prefetch goes into each half of set 0 as soon as the MFMAs reading that half have been issued
  auto tap_last_odd = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    multiply(I0{}, I0{});
    __builtin_amdgcn_sched_barrier(0);
    fetch_w(I0{}, I0{}, gtap + 1);
    multiply(I0{}, I1{});
    __builtin_amdgcn_sched_barrier(0);
    fetch_w(I0{}, I1{}, gtap + 1);
    __builtin_amdgcn_sched_barrier(0);
    ++gtap;
  };
  int j = 0, pend_mblk = -1, pend_n0 = 0, ntile_done = 0;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int tile = halo_tile(t, ntiles, nbn, a.reserved0 & 1);
    const int mblk = tile / nbn;
    const int m0 = mblk * 64, n0 = (tile - mblk * nbn) * 64;
    const int q0 = qbase(m0);
    const int ncol0 = n0 + wn * 32;
    woff = ncol0 < a.Cout ? ((unsigned)((a.wt_coff + ncol0) >> 5) * KB16) * 1024u + lane * 16u : OOB_OFF;
    ebase = qbase(min(m0 + wm * 32 + (lane & 31), M - 1)) - q0;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
    gtap = 0;
    fetch_w(I0{}, I0{}, 0);
    fetch_w(I0{}, I1{}, 0);
    for (int cc = 0; cc < NC; ++cc, ++j) {
      HALO_STAMP(4 * j);
      __syncthreads();        // barrier j
      HALO_STAMP(4 * j + 1);
      if (pend_mblk >= 0) {   // the previous tile's BN statistics: its wave rows' partials are behind a barrier now
        xbf_bn_flush<1, 1>(a, M, pend_n0, pend_mblk, tid, red_base + ((ntile_done - 1) & 1) * 256);
        pend_mblk = -1;
      }
      const unsigned char* buf = hsm + (j & 1) * BUF;
      kw = 0;
      tapoff = 0;
      fetch_a(I0{}, buf, 0);
      fetch_a(I1{}, buf, 0);
      int tp = 0;
      for (; tp + 2 < taps; tp += 2) {     // straight-line body: the compiler counts the outstanding W loads exactly
        tap(I0{}, I1{}, buf);
        tap(I1{}, I1{}, buf);
      }
      if (taps - tp == 2) {
        tap(I0{}, I1{}, buf);
        tap(I1{}, I0{}, buf);
      } else {
        tap_last_odd();
      }
      HALO_STAMP(4 * j + 2);
    }
    xbf_store_tile<1, 1>(a, acc, M, m0, n0, wm, wn, lane, red_base + (ntile_done & 1) * 256);
    HALO_STAMP(4 * (j - 1) + 3);
    if (a.bn_partial) {
      pend_mblk = mblk;
      pend_n0 = n0;
    }
    ++ntile_done;
  }
  __syncthreads();            // the final barrier
  if (pend_mblk >= 0) xbf_bn_flush<1, 1>(a, M, pend_n0, pend_mblk, tid, red_base + ((ntile_done - 1) & 1) * 256);
}

extern "C" int tpgsr_halo_trace(unsigned long long* buf) {   // buf: 8 * 8 * 256 uint64 of device memory, or nullptr to switch off
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_halo_trace), &buf, sizeof(buf)) != hipSuccess) {
    tpgsr_set_error("tpgsr_halo_trace: hipMemcpyToSymbol failed");
    return TPGSR_ERR_LAUNCH;
  }
  return 0;
}

// upper bound of the halo length of any 64-pixel tile (see the kernel's header)
static int halo_capacity(const tpgsr_conv_args* a) {
  const int Wp = a->OW + a->KW - 1, ohw = a->OH * a->OW;
  const int row_wraps = (a->OW % 64 == 0) ? 0 : 63 / a->OW + 1;
  const int img_wraps = (ohw % 64 == 0) ? 0 : 63 / ohw + 1;
  return 63 + row_wraps * (a->KW - 1) + img_wraps * (a->KH - 1) * Wp + (a->KH - 1) * Wp + a->KW;
}

extern "C" int tpgsr_halo_capacity(const tpgsr_conv_args* a) { return a ? halo_capacity(a) : -1; }   // (host-only; tests/test_halo_host_cpu.py)

static long long g_nmajor_min_bytes = 3ll << 20;
/* weight-plane size above which the halo forward kernel walks its tiles column-major per XCD (-1: never; 0: whenever the column-tile
 * count allows); default 3 MB */
extern "C" void tpgsr_halo_set_colmajor_min_bytes(long long v) { g_nmajor_min_bytes = v; }

/* smallest tap count the halo kernel takes (default 2; 1 sends 1x1 convolutions with Cin % 32 == 0 through it as well --
 * TPGSR_XBF_HALO_MINTAPS, experiment switch) */
static int g_halo_force_ne9 = 0;
/* tests: let the halo forward kernel take halos of 225..288 entries (its 9-entries-per-thread variant) */
extern "C" void tpgsr_halo_set_ne9(int on) { g_halo_force_ne9 = on ? 1 : 0; }
static int g_halo_min_taps = [] { const char* e = getenv("TPGSR_XBF_HALO_MINTAPS"); return e && e[0] == '1' ? 1 : 2; }();
extern "C" void tpgsr_halo_set_min_taps(int v) { g_halo_min_taps = v < 1 ? 1 : v; }

#define XBF_HALO_LD_CASES(X) X(0) X(1) X(2) X(3) X(4) X(5) X(7)

// returns 1 when launched, 0 when the shape is not one of the halo kernel's, < 0 on error
// does the two-workgroup halo kernel take this launch?  0: no; else the halo capacity (`small`: the 7-entries-per-thread variant)
static int halo_takes(const tpgsr_conv_args* a, int ld, bool* small_out) {
  static const bool on = [] { const char* e = getenv("TPGSR_XBF_HALO"); return !(e && e[0] == '0'); }();
  const int T = a->terms;
  if (!on || a->KH * a->KW < g_halo_min_taps || (a->wt_bf_cin != a->Cin && !(a->KH * a->KW == 1 && a->wt_bf_cin == 0)) || (a->Cin & 31) || a->stride_w > 1 || a->in_dil_w > 1 ||
      a->in_b || ((ld & ~7) && ld != 8) || ld == 6 || a->OW + a->KW - 1 < 8)
    return 0;
  const int Lcap = halo_capacity(a);
  const size_t lds = (size_t)2 * T * Lcap * 64 + 2048;      // two halo buffers + two 1 KB statistics scratch areas
  // two workgroups per CU or not at all: with one, nothing covers a workgroup's barriers and epilogues (the 16x50 recognizer
  // conv, 278 halo entries = 107 KB in x3 mode, measured 54 us here against 47 us on the tile loop)
  if (Lcap > 32 * 9 || lds > 80 * 1024) return 0;
  const bool small = Lcap <= 32 * 7;
  // the 9-entries-per-thread variant (halos of 225..288 entries: the recognizer's 16 x 50 maps).  In x3 arithmetic its two 107 KB
  // buffers never fit (rejected above); in x2 / bf16 they do (73 KB) and the launch was measured at 203 us against ~45 us on the tile
  // loop (128 -> 64 data gradient at batch 48, profiles/r03j_kernel_stats_c3_x2.md) -- off unless TPGSR_XBF_HALO_NE9=1
  static const bool ne9 = [] { const char* e = getenv("TPGSR_XBF_HALO_NE9"); return e && e[0] == '1'; }();
  if (!small && !ne9 && !g_halo_force_ne9) return 0;
  switch (ld) {
#define XBF_HALO_OK(B) case B:
    XBF_HALO_LD_CASES(XBF_HALO_OK)
#undef XBF_HALO_OK
    case 8: break;
    default: return 0;
  }
  if (small_out) *small_out = small;
  return Lcap;
}

static int conv_halo_xbf_launch(const tpgsr_conv_args* a, long long M, int ld, hipStream_t st) {
  const int T = a->terms;
  bool small = false;
  const int Lcap = halo_takes(a, ld, &small);
  if (Lcap <= 0) return 0;
  const size_t lds = (size_t)2 * T * Lcap * 64 + 2048;      // two halo buffers + two 1 KB statistics scratch areas
  const void* fn = nullptr;
#define XBF_HALO_CASE(B)                                                                                                      \
  case B:                                                                                                                     \
    fn = T == 1 ? (small ? (const void*)conv_halo_xbf_kernel<B, 1, 7> : (const void*)conv_halo_xbf_kernel<B, 1, 9>)           \
       : T == 2 ? (small ? (const void*)conv_halo_xbf_kernel<B, 2, 7> : (const void*)conv_halo_xbf_kernel<B, 2, 9>)           \
                : (small ? (const void*)conv_halo_xbf_kernel<B, 3, 7> : (const void*)conv_halo_xbf_kernel<B, 3, 9>);          \
    break;
  switch (ld) {
    XBF_HALO_LD_CASES(XBF_HALO_CASE)
    XBF_HALO_CASE(8)        // plain un-PixelShuffle gather (the data gradient of the upsample block's convolution)
    default: return 0;
  }
#undef XBF_HALO_CASE
  if (lds > 64 * 1024) {   // opt-in to > 64 KB of dynamic LDS, per (kernel, device): raised to the largest size seen so far
    static std::mutex mu;
    static std::vector<std::pair<std::pair<const void*, int>, size_t>> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
      tpgsr_set_error("tpgsr_conv_fwd: hipGetDevice failed");
      return TPGSR_ERR_LAUNCH;
    }
    std::lock_guard<std::mutex> lock(mu);
    size_t* cur = nullptr;
    for (auto& d : done)
      if (d.first.first == fn && d.first.second == dev) cur = &d.second;
    if (!cur || *cur < lds) {
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        tpgsr_set_error("tpgsr_conv_fwd: LDS opt-in (%zu bytes) for the halo kernel failed", lds);
        return TPGSR_ERR_LAUNCH;
      }
      if (cur) *cur = lds; else done.push_back({{fn, dev}, lds});
    }
  }
  // persistent grid: as many workgroups as the chip holds at once (occupancy x CUs), cached per (kernel, LDS size, device)
  struct Occ { const void* fn; size_t lds; int dev, wgs; };
  static std::mutex omu;
  static std::vector<Occ> occ;
  int resident = 0;
  {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
      tpgsr_set_error("tpgsr_conv_fwd: hipGetDevice failed");
      return TPGSR_ERR_LAUNCH;
    }
    std::lock_guard<std::mutex> lock(omu);
    for (auto& o : occ)
      if (o.fn == fn && o.lds == lds && o.dev == dev) resident = o.wgs;
    if (!resident) {
      int per_cu = 0, cus = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 512, lds) != hipSuccess || per_cu < 1 ||
          hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) {
        tpgsr_set_error("tpgsr_conv_fwd: occupancy query for the halo kernel failed (LDS %zu bytes)", lds);
        return TPGSR_ERR_LAUNCH;
      }
      resident = per_cu * cus;
      occ.push_back({fn, lds, dev, resident});
      static const bool dbg = [] { const char* e = getenv("TPGSR_XBF_DEBUG"); return e && e[0] == '1'; }();
      if (dbg)
        fprintf(stderr, "[tpgsr] halo conv: Cin %d Cout %d %dx%d ld %d T %d Lcap %d lds %zu -> %d workgroups/CU x %d CUs\n", a->Cin, a->Cout,
                a->KH, a->KW, ld, T, Lcap, lds, per_cu, cus);
    }
  }
  const long long ntiles = (long long)cdiv(M, 64) * cdiv(a->Cout, 64);
  dim3 grid((unsigned)(ntiles < resident ? ntiles : resident));
  int Mi = (int)M, Lc = Lcap;
  tpgsr_conv_args args = *a;
  // column-major tile order per XCD when the split weight planes would not stay in a 4 MB L2 next to the activations
  const int nbn = cdiv(a->Cout, 64);
  const long long w_bytes = (long long)T * a->kp * cdiv(a->Cout, 32) * 32 * 2;
  args.reserved0 = (g_nmajor_min_bytes >= 0 && w_bytes > g_nmajor_min_bytes && (nbn % 8 == 0 || 8 % nbn == 0) && nbn > 1 && (grid.x % 8 == 0 || grid.x == ntiles)) ? 1 : 0;
  void* params[] = {&args, &Mi, &Lc};
  if (hipLaunchKernel(fn, grid, dim3(512), params, lds, st) != hipSuccess) {
    tpgsr_set_error("tpgsr_conv_fwd(halo): launch failed: %s", hipGetErrorString(hipGetLastError()));
    return TPGSR_ERR_LAUNCH;
  }
  return 1;
}

// wave-tile choice of the tile loop (WMB x 1 blocks of 32 x 32 per wave; workgroup tile 64 WMB x 64).  TPGSR_XBF_TILE=11|21 forces one.
static int xbf_fwd_tile(long long M, int Cout) {
  static const int force = [] { const char* e = getenv("TPGSR_XBF_TILE"); return e ? atoi(e) : 0; }();
  if (force == 11 || force == 21) return force;
  return 11;
}

extern "C" int tpgsr_conv_panel_xbf_launch(const tpgsr_conv_args* a, long long M, int ld, hipStream_t st);
extern "C" int tpgsr_conv_halo3_xbf_launch(const tpgsr_conv_args* a, long long M, int ld, hipStream_t st);   // conv_halo3.hip

// ---- split-K (tpgsr_conv_args.sk_splits) ----
// on by default (C3 x2 interleaved on one box: 5.391 / 5.380 -> 5.363 / 5.340 ms per step, family replay 4.29 -> 4.15 ms; the STN head's
// 96-pixel convolutions 32 -> 13 us, InfoGen's 512 -> 128 47 -> 31, the BiLSTM projections' data gradients 34 / 30 -> 29 / 20);
// TPGSR_XBF_SPLITK=0 / tpgsr_splitk_set_enabled(0): every launch unsplit
static int g_sk_on = [] { const char* e = getenv("TPGSR_XBF_SPLITK"); return (e && e[0] == '0') ? 0 : 1; }();
extern "C" void tpgsr_splitk_set_enabled(int on) { g_sk_on = on ? 1 : 0; }
extern "C" int tpgsr_conv_halo3_would_take(const tpgsr_conv_args* a, long long M, int ld);    // conv_halo3.hip
extern "C" int tpgsr_conv_panel_would_take(const tpgsr_conv_args* a, long long M, int ld);    // conv_panel.hip

// launches with fewer tiles than ~2/3 of the CUs and >= 24 K chunks: S workgroups per tile so that ~640 are resident, >= 6 chunks each
static int splitk_choice(long long M, int Cout, int kp) {
  static const int target = [] { const char* e = getenv("TPGSR_XBF_SPLITK_TARGET"); return e ? atoi(e) : 640; }();      // resident workgroups aimed at
  static const int min_cps = [] { const char* e = getenv("TPGSR_XBF_SPLITK_MIN_CHUNKS"); return e ? atoi(e) : 6; }();   // chunks per split at least
  static const int max_tiles = [] { const char* e = getenv("TPGSR_XBF_SPLITK_MAX_TILES"); return e ? atoi(e) : 256; }();
  static const int min_k = [] { const char* e = getenv("TPGSR_XBF_SPLITK_MIN_K"); return e ? atoi(e) : 24; }();          // K chunks of the launch at least
  const long long ntiles = cdiv(M, 64) * cdiv(Cout, 64);
  const int nchunks = kp / KC;
  if (ntiles > max_tiles || nchunks < min_k) return 0;
  int S = (int)(target / ntiles);
  S = S < 2 ? 2 : S > 8 ? 8 : S;
  int cps = (nchunks + S - 1) / S;
  if (cps < min_cps) cps = min_cps;
  S = (nchunks + cps - 1) / cps;          // no empty split
  return S > 1 ? S : 0;
}

extern "C" int tpgsr_conv_splitk_plan(const tpgsr_conv_args* a, long long* bytes) {
  if (bytes) *bytes = 0;
  if (!a || !g_sk_on || !(a->terms > 0 && a->terms <= 3 && a->wt_bf && (a->Cin & 3) == 0 && (a->wt_coff & 31) == 0) || a->bn_row_tiles > 1 || a->in2_scale)
    return 0;
  const int ld = (a->in_scale ? 1 : 0) | (a->in_act ? 2 : 0) | (a->in2 ? 4 : 0) | (a->in_ps ? 8 : 0) | (a->in_b ? 16 : 0);
  switch (ld) {
#define XBF_SK_OK(B) case B:
    XBF_LD_CASES(XBF_SK_OK)
#undef XBF_SK_OK
    break;
    default: return 0;
  }
  const long long M = (long long)a->N * a->OH * a->OW;
  // (a launch the two-workgroup halo kernel would take is split all the same -- conv6, 2 x 2 over 1248 pixels: 36.7 -> 31.3 us in x3,
  //  28.9 -> 23.5 in x2 -- unless TPGSR_XBF_SPLITK_OVER_HALO=0; an explicit sk_splits wins in the launcher)
  static const int over_halo = [] { const char* e = getenv("TPGSR_XBF_SPLITK_OVER_HALO"); return (e && e[0] == '0') ? 0 : 1; }();
  if (tpgsr_conv_halo3_would_take(a, M, ld) || (!over_halo && halo_takes(a, ld, nullptr) > 0) || tpgsr_conv_panel_would_take(a, M, ld)) return 0;
  const int S = splitk_choice(M, a->Cout, a->kp);
  if (S > 1 && bytes) *bytes = (long long)S * cdiv(M, 64) * cdiv(a->Cout, 64) * 256 * 16 * 4;
  return S;
}

extern "C" int tpgsr_conv_fwd_xbf_launch(const tpgsr_conv_args* a, long long M, int K, int ld, hipStream_t st) {
  const int T = a->terms;
  TPGSR_CHECK_ARG(a->wt_bf_cin == 0 || (a->wt_bf_cin == a->Cin && (a->Cin & 31) == 0),
                  "tpgsr_conv_fwd: weights were split in channel-block order for Cin %d, the convolution has Cin %d", a->wt_bf_cin, a->Cin);
  if (a->sk_splits > 1) {       // split-K: S workgroups per tile + the reduce / epilogue launch (the caller asked tpgsr_conv_splitk_plan)
    TPGSR_CHECK_ARG(a->sk_part && a->sk_splits <= 64 && a->sk_splits <= a->kp / KC && ((uintptr_t)a->sk_part & 15) == 0,
                    "tpgsr_conv_fwd: sk_splits %d needs sk_part (16-byte aligned) and at most one split per K chunk (%d)", a->sk_splits, a->kp / KC);
    const unsigned ntile = (unsigned)(cdiv(M, 64) * cdiv(a->Cout, 64));
    dim3 gsk(ntile * (unsigned)a->sk_splits);
#define XBF_SK_CASE(B)                                                                                                   \
  case B:                                                                                                                \
    if (T == 1) hipLaunchKernelGGL((conv_fwd_xbf_kernel<B, 1, 1, 1, 1>), gsk, dim3(256), 0, st, *a, (int)M, K);          \
    else if (T == 2) hipLaunchKernelGGL((conv_fwd_xbf_kernel<B, 2, 1, 1, 1>), gsk, dim3(256), 0, st, *a, (int)M, K);     \
    else hipLaunchKernelGGL((conv_fwd_xbf_kernel<B, 3, 1, 1, 1>), gsk, dim3(256), 0, st, *a, (int)M, K);                 \
    break;
    switch (ld) {
      XBF_LD_CASES(XBF_SK_CASE)
      default:
        tpgsr_set_error("tpgsr_conv_fwd: unsupported loader combination %d", ld);
        return TPGSR_ERR_ARG;
    }
#undef XBF_SK_CASE
    hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3(ntile), dim3(256), 0, st, *a, (int)M);
    TPGSR_LAUNCH_CHECK("tpgsr_conv_fwd(bf16 MFMA, split-K)");
  }
  const int h3 = tpgsr_conv_halo3_xbf_launch(a, M, ld, st);      // whole-CU kernel: three tiles per workgroup, one round of the chip
  if (h3 < 0) return h3;
  if (h3 > 0) TPGSR_LAUNCH_CHECK("tpgsr_conv_fwd(bf16 MFMA, whole-CU halo)");
  TPGSR_CHECK_ARG(!(ld & 32), "tpgsr_conv_fwd: a scaled residual operand (in2_scale) exists in the whole-CU halo kernel's loader only, which does not "
                  "take this launch (ask tpgsr_conv_in2_scale_ok first)");
  TPGSR_CHECK_ARG(a->bn_row_tiles <= 1, "tpgsr_conv_fwd: bn_row_tiles %d is the whole-CU halo kernel's, which does not take this launch "
                  "(ask tpgsr_conv_bn_row_tiles first)", a->bn_row_tiles);
  const int h = conv_halo_xbf_launch(a, M, ld, st);
  if (h < 0) return h;
  if (h > 0) TPGSR_LAUNCH_CHECK("tpgsr_conv_fwd(bf16 MFMA, halo)");
  const int pn = tpgsr_conv_panel_xbf_launch(a, M, ld, st);       // conv_panel.hip: 1x1 convolutions with a short K over many pixels
  if (pn < 0) return pn;
  if (pn > 0) TPGSR_LAUNCH_CHECK("tpgsr_conv_fwd(bf16 MFMA, panel)");
  const int cfg = xbf_fwd_tile(M, a->Cout);
  const int wmb = cfg / 10;
  dim3 grid(cdiv(M, 64 * wmb) * cdiv(a->Cout, 64));
#define XBF_FWD_T(B, TT)                                                                                                    \
  switch (cfg) {                                                                                                            \
    case 21: hipLaunchKernelGGL((conv_fwd_xbf_kernel<B, TT, 2, 1>), grid, dim3(256), 0, st, *a, (int)M, K); break;          \
    default: hipLaunchKernelGGL((conv_fwd_xbf_kernel<B, TT, 1, 1>), grid, dim3(256), 0, st, *a, (int)M, K); break;          \
  }
#define XBF_FWD_CASE(B)                  \
  case B:                                \
    if (T == 1) { XBF_FWD_T(B, 1) }      \
    else if (T == 2) { XBF_FWD_T(B, 2) } \
    else { XBF_FWD_T(B, 3) }             \
    break;
  switch (ld) {
    XBF_LD_CASES(XBF_FWD_CASE)
    default:
      tpgsr_set_error("tpgsr_conv_fwd: unsupported loader combination %d", ld);
      return TPGSR_ERR_ARG;
  }
#undef XBF_FWD_CASE
#undef XBF_FWD_T
  TPGSR_LAUNCH_CHECK("tpgsr_conv_fwd(bf16 MFMA)");
}

// ------------------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------------------
#define WK 64
#define WM 32

// transposing fragment fetch: 8 consecutive pixels (contraction index) of channel `col0 + (lane & 31)` from a pixel-major
// [32 pixels][64 channels] bf16 image.  One ds_read_b64_tr_b16 hands lane l the column (l & 15) of the [4 pixels][16 channels]
// block its 16-lane group addresses (lane q of the group points at pixel row q>>2, channels 4*(q&3)..+3).
__device__ __forceinline__ bf16x8 frag_tr(const unsigned char* plane, int lane, int col0, int mb) {
  const int G = lane >> 4, q = lane & 15;
  const int mbase = mb * 16 + (G >> 1) * 8;
  const int c0 = col0 + (G & 1) * 16 + (q & 3) * 4;
  const unsigned char* p = plane + (mbase + (q >> 2)) * XW_ROW + c0 * 2;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + 4 * XW_ROW));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

extern "C" void tpgsr_wgrad_plan_host(long long M, int K, int Cout, int* Z, int* MB);   // conv_mfma.hip
extern "C" int tpgsr_loader_bits(const tpgsr_conv_args* a);

// (bx of gx: this workgroup's index within ITS launch -- the whole grid, or one item's share of a batched launch)
template <int LD, int T>
__device__ __forceinline__ void conv_wgrad_xbf_body(const tpgsr_wgrad_args& w, const int M, const int K, const int MB, const unsigned bx,
                                                    const unsigned gx, unsigned char* Am, unsigned char* Ym) {
  const tpgsr_conv_args& a = w.c;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wk = wave & 1, wn = wave >> 1;
  const int nkb = (K + WK - 1) / WK, nnb = (a.Cout + BN - 1) / BN;
  const int tile = xcd_remap(bx, gx);       // k-block fastest: the k-blocks of one pixel split share an L2
  const int kblk = tile % nkb, nblk = (tile / nkb) % nnb, zblk = tile / (nkb * nnb);
  const int k0 = kblk * WK, n0 = nblk * BN;
  const int mbeg = zblk * MB;
  const int mend = min(M, mbeg + MB);

  // A staging: quad (tid&15) of this block's 64 k rows (fixed for the whole kernel), pixels (tid>>4), +16
  const int aq = tid & 15;
  const int ap0 = tid >> 4;
  const KPos kp = kpos_init(a, (k0 >> 2) + aq);
  const int ac = kp.c;
  const int yc = (tid & 15) * 4;

  ARaw qa0, qa1;
  float4 ry0, ry1;
  float4 qs = make_float4(1.f, 1.f, 1.f, 1.f), qt = make_float4(0.f, 0.f, 0.f, 0.f);
  const int Wr_ = real_w(a);
  const size_t in_floats = a.in_ps ? (size_t)a.N * a.H * a.W * a.Cin : (size_t)a.N * a.H * Wr_ * a.in_ld;
  const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(a.in, in_floats);
  const __amdgpu_buffer_rsrc_t rs_in2 = (LD & 16) ? make_rsrc(a.in_b, (size_t)a.N * Wr_ * a.in_b_ld)
                                                  : make_rsrc(a.in2 ? a.in2 : a.in, (size_t)a.N * a.H * Wr_ * a.in2_ld);
  const __amdgpu_buffer_rsrc_t rs_dy = make_rsrc(w.dy, w.dy_ps ? (size_t)M * a.Cout : (size_t)M * w.dy_ld);
  if ((LD & 1) && kp.kh < a.KH) {
    qs = *reinterpret_cast<const float4*>(a.in_scale + ac);
    qt = *reinterpret_cast<const float4*>(a.in_shift + ac);
  }
  const bool cok = n0 + yc < a.Cout;
  auto load_dy = [&](const PixelPos& p, int m) -> float4 {
    if (!w.dy_ps) return buf_load4(rs_dy, (p.valid && cok) ? ((unsigned)m * (unsigned)w.dy_ld + (unsigned)(w.dy_coff + n0 + yc)) * 4u : OOB_OFF);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!p.valid || !cok) return v;
    // logical channels col..col+3 = (cs = col/4, i, j) of a [N][2OH][2OW][Cout/4] tensor
    const int C4 = a.Cout >> 2, cs = (n0 + yc) >> 2;
    const size_t W2 = 2 * (size_t)a.OW;
    const float* b = w.dy + ((size_t)(p.n * 2 * a.OH + 2 * p.oh) * W2 + 2 * p.ow) * C4 + cs;
    v.x = b[0];
    v.y = b[C4];
    v.z = b[W2 * C4];
    v.w = b[W2 * C4 + C4];
    return v;
  };
  auto load_chunk = [&](int mc) {
    const int ma = mc + ap0, mb = mc + ap0 + 16;
    const PixelPos p0 = decode_pixel(a, ma, mend);
    const PixelPos p1 = decode_pixel(a, mb, mend);
    qa0 = load_a_raw<LD>(a, rs_in, rs_in2, p0, kp);
    qa1 = load_a_raw<LD>(a, rs_in, rs_in2, p1, kp);
    ry0 = load_dy(p0, ma);
    ry1 = load_dy(p1, mb);
  };
  float4 dbq = make_float4(0.f, 0.f, 0.f, 0.f);   // column sums of dy over this thread's pixel rows (bias gradient)
  const bool want_db = (w.dbpart != nullptr) && kblk == 0;
  auto store_chunk = [&]() {
    const float4 v0 = finish_a<LD>(a, qa0, qs, qt);
    const float4 v1 = finish_a<LD>(a, qa1, qs, qt);
    uint2 h0[T], h1[T], y0[T], y1[T];
    split4<T>(v0, h0);
    split4<T>(v1, h1);
    split4<T>(ry0, y0);
    split4<T>(ry1, y1);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      *reinterpret_cast<uint2*>(Am + t * XW_PLANE + ap0 * XW_ROW + aq * 8) = h0[t];
      *reinterpret_cast<uint2*>(Am + t * XW_PLANE + (ap0 + 16) * XW_ROW + aq * 8) = h1[t];
      *reinterpret_cast<uint2*>(Ym + t * XW_PLANE + ap0 * XW_ROW + aq * 8) = y0[t];
      *reinterpret_cast<uint2*>(Ym + t * XW_PLANE + (ap0 + 16) * XW_ROW + aq * 8) = y1[t];
    }
    if (want_db) {
      dbq.x += ry0.x + ry1.x;
      dbq.y += ry0.y + ry1.y;
      dbq.z += ry0.z + ry1.z;
      dbq.w += ry0.w + ry1.w;
    }
  };

  floatx16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  if (mbeg < mend) {
    load_chunk(mbeg);
    store_chunk();
  }
  __syncthreads();
  for (int mc = mbeg; mc < mend; mc += WM) {
    const bool more = mc + WM < mend;
    if (more) load_chunk(mc + WM);
#pragma unroll
    for (int mb = 0; mb < WM / 16; ++mb) {
      bf16x8 av[T], bv[T];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        av[t] = frag_tr(Am + t * XW_PLANE, lane, wk * 32, mb);
        bv[t] = frag_tr(Ym + t * XW_PLANE, lane, wn * 32, mb);
      }
      acc = mfma_terms<T>(av, bv, acc);
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    if (more) {
      store_chunk();
      __syncthreads();
    }
  }
  const int bcol = wn * 32 + (lane & 31);
  const int n = n0 + bcol;
  float* dst = w.part + (size_t)zblk * K * a.Cout;
  if (n < a.Cout) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int k = k0 + wk * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (k < K) dst[(size_t)k * a.Cout + n] = acc[r];
    }
  }
  if (want_db) {   // combine the 16 pixel-row lanes of every channel quad in a fixed order
    float* red = reinterpret_cast<float*>(Am);        // [16 row lanes][64 channels]; MFMA reads are behind the last barrier
    *reinterpret_cast<float4*>(red + ap0 * 64 + yc) = dbq;
    __syncthreads();
    if (tid < BN && n0 + tid < a.Cout) {
      float sdb = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) sdb += red[r * 64 + tid];
      w.dbpart[(size_t)zblk * a.Cout + n0 + tid] = sdb;
    }
  }
}

template <int LD, int T>
__global__ __launch_bounds__(256) void conv_wgrad_xbf_kernel(tpgsr_wgrad_args w, int M, int K, int MB) {
  __shared__ __attribute__((aligned(16))) unsigned char Am[T * XW_PLANE];
  __shared__ __attribute__((aligned(16))) unsigned char Ym[T * XW_PLANE];
  conv_wgrad_xbf_body<LD, T>(w, M, K, MB, blockIdx.x, gridDim.x, Am, Ym);
}

// THREE k-blocks per workgroup (round 6).  The tile loop above spends ~600 instructions per wave on a 32-pixel chunk -- pixel decoding by
// integer division, address arithmetic, the split of BOTH operands, LDS staging -- for SIX MFMAs: its launches are instruction-bound at
// 0.10-0.13 of their class peak (SQ counters, profiles/r06_sq_family.md: matrix pipe 13 % busy, 32 % of the wave cycles issuing).  Here a
// workgroup owns 192 k-rows x 64 columns (three taps of a 64-channel layer: one kernel row of the trunk's 3x3): dy is loaded, split and
// staged ONCE for three times the MFMAs, every wave keeps three independent accumulator chains, and the output pixel of a thread is
// advanced from chunk to chunk by additions (no division inside the loop).  Same slabs (part[z][k][n]), same per-split pixel ranges, same
// summation order per output element as the one-block kernel.  Plain / affine loaders, dense dy, T <= 2.
template <int LD, int T>
__global__ __launch_bounds__(256) void conv_wgrad_xbf3_kernel(tpgsr_wgrad_args w, int M, int K, int MB) {
  __shared__ __attribute__((aligned(16))) unsigned char Am[3][T * XW_PLANE];
  __shared__ __attribute__((aligned(16))) unsigned char Ym[T * XW_PLANE];
  const tpgsr_conv_args& a = w.c;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wk = wave & 1, wn = wave >> 1;
  const int nkb = K / (3 * WK), nnb = (a.Cout + BN - 1) / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);       // k-block fastest: the k-blocks of one pixel split share an L2
  const int kblk = tile % nkb, nblk = (tile / nkb) % nnb, zblk = tile / (nkb * nnb);
  const int k0 = kblk * 3 * WK, n0 = nblk * BN;
  const int mbeg = zblk * MB;
  const int mend = min(M, mbeg + MB);
  const int aq = tid & 15, ap0 = tid >> 4, yc = (tid & 15) * 4;

  // this thread's three A quads: (tap, 4 channels) of k-block j, fixed for the whole kernel
  int dih[3], diw[3], coff[3];
  float4 qs[3], qt[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const KPos kp = kpos_init(a, ((k0 + j * WK) >> 2) + aq);
    dih[j] = kp.kh - a.pad_h;
    diw[j] = kp.kw - a.pad_w;
    coff[j] = a.in_coff + kp.c;
    qs[j] = make_float4(1.f, 1.f, 1.f, 1.f);
    qt[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (LD & 1) {
      qs[j] = *reinterpret_cast<const float4*>(a.in_scale + kp.c);
      qt[j] = *reinterpret_cast<const float4*>(a.in_shift + kp.c);
    }
  }
  const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(a.in, (size_t)a.N * a.H * a.W * a.in_ld);
  const __amdgpu_buffer_rsrc_t rs_dy = make_rsrc(w.dy, (size_t)M * w.dy_ld);
  const bool cok = n0 + yc < a.Cout;

  // output pixels of this thread's two rows (ap0, ap0 + 16) of the current chunk, decoded once and advanced by WM per chunk
  int pm[2], pn[2], poh[2], pow_[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int m = mbeg + ap0 + 16 * r, ohw = a.OH * a.OW;
    pm[r] = m;
    pn[r] = m / ohw;
    const int rem = m - pn[r] * ohw;
    poh[r] = rem / a.OW;
    pow_[r] = rem - poh[r] * a.OW;
  }
  float4 qa[2][3], ry[2];
  bool qok[2][3];
  auto load_chunk = [&]() __attribute__((always_inline)) {      // the chunk the position state points at; then advance the state
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const bool valid = pm[r] < mend;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int ih = poh[r] + dih[j], iw = pow_[r] + diw[j];
        const bool ok = valid && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
        qok[r][j] = ok;
        qa[r][j] = buf_load4(rs_in, ok ? ((unsigned)((pn[r] * a.H + ih) * a.W + iw) * (unsigned)a.in_ld + (unsigned)coff[j]) * 4u : OOB_OFF);
      }
      ry[r] = buf_load4(rs_dy, (valid && cok) ? ((unsigned)pm[r] * (unsigned)w.dy_ld + (unsigned)(w.dy_coff + n0 + yc)) * 4u : OOB_OFF);
      pm[r] += WM;
      pow_[r] += WM;
      while (pow_[r] >= a.OW) {        // (WM / OW + 1 iterations at most)
        pow_[r] -= a.OW;
        if (++poh[r] == a.OH) {
          poh[r] = 0;
          ++pn[r];
        }
      }
    }
  };
  float4 dbq = make_float4(0.f, 0.f, 0.f, 0.f);   // column sums of dy over this thread's pixel rows (bias gradient)
  const bool want_db = (w.dbpart != nullptr) && kblk == 0;
  auto store_chunk = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int row = ap0 + 16 * r;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        float4 v = qa[r][j];
        if (LD & 1) {      // (the affine maps the hardware zero fill to t: padding is re-zeroed)
          v.x = qok[r][j] ? v.x * qs[j].x + qt[j].x : 0.f;
          v.y = qok[r][j] ? v.y * qs[j].y + qt[j].y : 0.f;
          v.z = qok[r][j] ? v.z * qs[j].z + qt[j].z : 0.f;
          v.w = qok[r][j] ? v.w * qs[j].w + qt[j].w : 0.f;
        }
        uint2 h[T];
        split4<T>(v, h);
#pragma unroll
        for (int t = 0; t < T; ++t) *reinterpret_cast<uint2*>(Am[j] + t * XW_PLANE + row * XW_ROW + aq * 8) = h[t];
      }
      uint2 y[T];
      split4<T>(ry[r], y);
#pragma unroll
      for (int t = 0; t < T; ++t) *reinterpret_cast<uint2*>(Ym + t * XW_PLANE + row * XW_ROW + aq * 8) = y[t];
    }
    if (want_db) {      // (the pair first, as the one-block kernel: the same partial sums bit for bit)
      dbq.x += ry[0].x + ry[1].x;
      dbq.y += ry[0].y + ry[1].y;
      dbq.z += ry[0].z + ry[1].z;
      dbq.w += ry[0].w + ry[1].w;
    }
  };

  floatx16 acc[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

  if (mbeg < mend) {
    load_chunk();
    store_chunk();
  }
  __syncthreads();
  for (int mc = mbeg; mc < mend; mc += WM) {
    const bool more = mc + WM < mend;
    if (more) load_chunk();
#pragma unroll
    for (int mb = 0; mb < WM / 16; ++mb) {
      bf16x8 bv[T];
#pragma unroll
      for (int t = 0; t < T; ++t) bv[t] = frag_tr(Ym + t * XW_PLANE, lane, wn * 32, mb);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        bf16x8 av[T];
#pragma unroll
        for (int t = 0; t < T; ++t) av[t] = frag_tr(Am[j] + t * XW_PLANE, lane, wk * 32, mb);
        acc[j] = mfma_terms<T>(av, bv, acc[j]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    if (more) {
      store_chunk();
      __syncthreads();
    }
  }
  const int n = n0 + wn * 32 + (lane & 31);
  float* dst = w.part + (size_t)zblk * K * a.Cout;
  if (n < a.Cout) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = k0 + j * WK + wk * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        dst[(size_t)k * a.Cout + n] = acc[j][r];
      }
  }
  if (want_db) {   // combine the 16 pixel-row lanes of every channel quad in a fixed order
    float* red = reinterpret_cast<float*>(Am[0]);        // [16 row lanes][64 channels]; MFMA reads are behind the last barrier
    *reinterpret_cast<float4*>(red + ap0 * 64 + yc) = dbq;
    __syncthreads();
    if (tid < BN && n0 + tid < a.Cout) {
      float sdb = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) sdb += red[r * 64 + tid];
      w.dbpart[(size_t)zblk * a.Cout + n0 + tid] = sdb;
    }
  }
}

static int g_wg3_on = [] { const char* e = getenv("TPGSR_XBF_WGRAD3"); return (e && e[0] == '0') ? 0 : 1; }();
/* experiment / test switch: 0 sends every tile-loop weight gradient back to the one-block kernel */
extern "C" void tpgsr_wgrad3_set_enabled(int on) { g_wg3_on = on ? 1 : 0; }

// Several INDEPENDENT weight-gradient GEMMs in one launch (a device-resident item table, like the pack / reduce programs).  The two
// BiLSTM layers of the text-prior generator have ten of them -- 2 directions x (hidden side, input side) + the embedding, per layer --
// over M = N T = 1248 rows each: a launch alone is 25-65 us of start-up, four splits of ten 32-row chunks and a slab write, and the ten
// in a row were 335 us at the END of the training step, where the weight-gradient stream is the only one still running.  Together
// they are ~2000 workgroups that fill the chip once.
template <int LD, int T>
__global__ __launch_bounds__(256) void conv_wgrad_xbf_batch_kernel(const tpgsr_wgrad_batch_item* __restrict__ items, int n) {
  __shared__ __attribute__((aligned(16))) unsigned char Am[T * XW_PLANE];
  __shared__ __attribute__((aligned(16))) unsigned char Ym[T * XW_PLANE];
  __shared__ int s_d;
  if (threadIdx.x == 0) {
    int lo = 0, hi = n - 1;  // last item whose blk0 <= blockIdx.x
    while (lo < hi) {
      int mid = (lo + hi + 1) >> 1;
      if (items[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    s_d = lo;
  }
  __syncthreads();
  const tpgsr_wgrad_batch_item* it = items + s_d;
  const tpgsr_wgrad_args w = it->w;
  conv_wgrad_xbf_body<LD, T>(w, it->M, it->K, it->MB, blockIdx.x - (unsigned)it->blk0, (unsigned)it->nblk, Am, Ym);
}

/* host side of a batched launch: what tpgsr_conv_wgrad would launch for *w -- filled into *item (blk0 is the caller's prefix sum of nblk).
 * Returns the loader variant (>= 0) when the launch is the tile-loop split-bf16 kernel's (1x1 geometry, terms > 0, 16-byte rows), which is
 * what a batch is made of; TPGSR_ERR_ARG otherwise (the caller launches that one on its own). */
extern "C" int tpgsr_conv_wgrad_batch_prepare(const tpgsr_wgrad_args* w, tpgsr_wgrad_batch_item* item) {
  TPGSR_CHECK_ARG(w && item && w->dy && w->part, "tpgsr_conv_wgrad_batch_prepare: null pointer");
  const tpgsr_conv_args* a = &w->c;
  const long long M = (long long)a->N * a->OH * a->OW;
  const int K = a->KH * a->KW * a->Cin;
  const bool vecY = !w->dy_ps && (w->dy_ld & 3) == 0 && (w->dy_coff & 3) == 0 && w->dy_ld >= ((a->Cout + 3) & ~3) + w->dy_coff &&
                    ((uintptr_t)w->dy & 15) == 0;
  if (!(a->terms >= 1 && a->terms <= 3 && a->KH * a->KW == 1 && (a->Cin & 3) == 0 && vecY && M < (1ll << 31) &&
        M * w->dy_ld * 4 <= 0x7fffffffll && a->wt_bf_cin == 0 && !w->dy_bf)) {
    tpgsr_set_error("tpgsr_conv_wgrad_batch_prepare: not a tile-loop split-bf16 launch");
    return TPGSR_ERR_ARG;
  }
  int Z, MB;
  tpgsr_wgrad_plan_host(M, K, a->Cout, &Z, &MB);
  if (w->zsplits > 0) {
    Z = w->zsplits;
    MB = cdiv(cdiv(M, 64), Z) * 64;
  }
  item->w = *w;
  item->M = (int)M;
  item->K = K;
  item->MB = MB;
  item->blk0 = 0;
  item->nblk = cdiv(K, WK) * cdiv(a->Cout, BN) * Z;
  return tpgsr_loader_bits(a);
}

extern "C" int tpgsr_conv_wgrad_batch(const tpgsr_wgrad_batch_item* items_dev, int n, int total_blocks, int ld, int terms, void* stream) {
  TPGSR_CHECK_ARG(items_dev && n > 0 && total_blocks > 0 && terms >= 1 && terms <= 3, "tpgsr_conv_wgrad_batch: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(total_blocks);
#define XBF_WGB_CASE(B)                                                                                                 \
  case B:                                                                                                               \
    if (terms == 1) hipLaunchKernelGGL((conv_wgrad_xbf_batch_kernel<B, 1>), grid, dim3(256), 0, st, items_dev, n);      \
    else if (terms == 2) hipLaunchKernelGGL((conv_wgrad_xbf_batch_kernel<B, 2>), grid, dim3(256), 0, st, items_dev, n); \
    else hipLaunchKernelGGL((conv_wgrad_xbf_batch_kernel<B, 3>), grid, dim3(256), 0, st, items_dev, n);                 \
    break;
  switch (ld) {
    XBF_WGB_CASE(0) XBF_WGB_CASE(1) XBF_WGB_CASE(3)
    default:
      tpgsr_set_error("tpgsr_conv_wgrad_batch: unsupported loader combination %d", ld);
      return TPGSR_ERR_ARG;
  }
#undef XBF_WGB_CASE
  TPGSR_LAUNCH_CHECK("tpgsr_conv_wgrad_batch");
}

extern "C" int tpgsr_conv_wgrad_xbf_launch(const tpgsr_wgrad_args* w, long long M, int K, int Z, int MB, int ld, hipStream_t st) {
  const tpgsr_conv_args* a = &w->c;
  const int T = a->terms;
  // three k-blocks per workgroup where the shape allows: K a multiple of 192, the plain / affine loader, dense dy, a plain stride-1 geometry
  if (g_wg3_on && T >= 1 && T <= 2 && K % (3 * WK) == 0 && (ld == 0 || ld == 1) && !w->dy_ps && a->in_dil_w <= 1 && a->stride_w <= 1 &&
      !a->in_ps && (a->Cin & 3) == 0 && M * (long long)w->dy_ld * 4 <= 0x7fffffffll) {
    dim3 grid3((K / (3 * WK)) * cdiv(a->Cout, BN) * Z);
    if (ld == 0) {
      if (T == 1) hipLaunchKernelGGL((conv_wgrad_xbf3_kernel<0, 1>), grid3, dim3(256), 0, st, *w, (int)M, K, MB);
      else hipLaunchKernelGGL((conv_wgrad_xbf3_kernel<0, 2>), grid3, dim3(256), 0, st, *w, (int)M, K, MB);
    } else {
      if (T == 1) hipLaunchKernelGGL((conv_wgrad_xbf3_kernel<1, 1>), grid3, dim3(256), 0, st, *w, (int)M, K, MB);
      else hipLaunchKernelGGL((conv_wgrad_xbf3_kernel<1, 2>), grid3, dim3(256), 0, st, *w, (int)M, K, MB);
    }
    TPGSR_LAUNCH_CHECK("tpgsr_conv_wgrad(bf16 MFMA, three k-blocks)");
  }
  dim3 grid(cdiv(K, WK) * cdiv(a->Cout, BN) * Z);
#define XBF_WG_CASE(B)                                                                                       \
  case B:                                                                                                    \
    if (T == 1) hipLaunchKernelGGL((conv_wgrad_xbf_kernel<B, 1>), grid, dim3(256), 0, st, *w, (int)M, K, MB); \
    else if (T == 2) hipLaunchKernelGGL((conv_wgrad_xbf_kernel<B, 2>), grid, dim3(256), 0, st, *w, (int)M, K, MB); \
    else hipLaunchKernelGGL((conv_wgrad_xbf_kernel<B, 3>), grid, dim3(256), 0, st, *w, (int)M, K, MB);        \
    break;
  switch (ld) {
    XBF_WG_CASE(0) XBF_WG_CASE(1) XBF_WG_CASE(2) XBF_WG_CASE(3) XBF_WG_CASE(4) XBF_WG_CASE(5) XBF_WG_CASE(7) XBF_WG_CASE(17)
    default:
      tpgsr_set_error("tpgsr_conv_wgrad: unsupported loader combination %d", ld);
      return TPGSR_ERR_ARG;
  }
#undef XBF_WG_CASE
  TPGSR_LAUNCH_CHECK("tpgsr_conv_wgrad(bf16 MFMA)");
}

// ------------------------------------------------------------------------------------------------------
// weight gradient of KH x KW > 1 x 1 convolutions: HALO kernel (the forward kernel's mirror image).
//   part[z][k = tap Cin + ci][n] = sum over the pixels m of split z of  A[m][k] dy[m][n]
// The tile loop above re-loads and re-splits A once per tap AND dy once per 64-row k-block (9x each for a 3x3), with 12 MFMAs
// between two barriers.  Here a workgroup owns (pixel split z, channel block of 32, 64 output channels) and walks its pixels in
// tiles of 64: the producers keep the halo of the tile (as in the forward kernel: every input pixel any tap touches, split once)
// in LDS, the consumers contract over the tile's pixels for EVERY tap out of that one image --
//   A^T fragments (32 channels x 16 pixels): ds_read_b64_tr_b16 from the pixel-major halo image at (entry of the pixel) + tap offset,
//   dy fragments (16 pixels x 32 channels): pre-split ONCE per launch by dy_split_kernel into MFMA fragment order
//   [term][m / 16][n / 32][lane][8] and loaded straight into registers (like the weights of the forward kernel),
// and keep the [taps x 32 x 64] result block in registers across all tiles of the split (no atomics; the splits are summed by
// tpgsr_wgrad_reduce as before).  12 waves = 8 consumers (2 column halves x 4 tap groups of <= 3 taps: 48 accumulator
// registers) + 4 producers, one workgroup per CU at <= 168 registers.
// Bias gradient: column sums of dy = ones[32 x 16] x dy fragments on the matrix pipe (tap group 0 of channel block 0).
// ------------------------------------------------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(256) void dy_split_kernel(const float* __restrict__ dy, int dy_ld, int dy_coff, int dy_ps, int M, int Cout,
                                                       int OH, int OW, unsigned short* __restrict__ out, int MB16, int NB32) {
  const int lane = threadIdx.x & 63;
  const long long b = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= (long long)MB16 * NB32) return;
  const int mb = (int)(b / NB32), nb = (int)(b - (long long)mb * NB32);
  const int n = nb * 32 + (lane & 31), p0 = mb * 16 + (lane >> 5) * 8;
  __bf16 h[8][T];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int m = p0 + j;
    float v = 0.f;
    if (m < M && n < Cout) {
      if (!dy_ps) {
        v = dy[(size_t)m * dy_ld + dy_coff + n];
      } else {   // logical channel n = (cs, i, jj) of a [N][2 OH][2 OW][Cout / 4] tensor
        const int ohw = OH * OW, nn = m / ohw, rem = m - nn * ohw, oh = rem / OW, ow = rem - oh * OW;
        v = dy[((size_t)(nn * 2 * OH + 2 * oh + ((n >> 1) & 1)) * (2 * OW) + 2 * ow + (n & 1)) * (Cout >> 2) + (n >> 2)];
      }
    }
    split_bf<T>(v, h[j]);
  }
  const size_t plane = (size_t)MB16 * NB32 * 512;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = h[j][t];
    *reinterpret_cast<bf16x8*>(out + t * plane + ((size_t)b * 64 + lane) * 8) = v;
  }
}

// Division of a 31-bit numerator by a launch constant as multiply-high + shift (Granlund-Montgomery, round-up form): the halo kernels turn a
// tile index into (image, row, column) with four to six integer divisions per tile and thread, ~30 VALU instructions each on this ISA --
// tools/lab/wgh_probe.py: with every load, store and MFMA of the halo weight-gradient kernel switched off, a tile still took 1.65 us.
struct FastDiv {
  unsigned mul, sh;
};
static FastDiv fastdiv_make(unsigned d) {
  FastDiv f;
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  f.sh = l;
  f.mul = (unsigned)((((1ull << l) - d) << 32) / d + 1);
  return f;
}
__device__ __forceinline__ int fastdiv(int n, const FastDiv f) {      // 0 <= n < 2^31
  const unsigned t = __umulhi((unsigned)n, f.mul);
  return (int)((t + (unsigned)n) >> f.sh);
}
struct HaloDivs {
  FastDiv ohw, ow, hpwp, wp;
};

// LAB BUILDS ONLY (-DTPGSR_LAB, tools/lab/wgh_probe.py): bit 0 = consumers skip their fragment reads + MFMAs, bit 1 = producers issue no
// global loads after the first tile, bit 2 = producers skip the prologue / split / LDS stores.  Results are garbage with any bit set, so
// the switch, its export and its run-time branches do not exist in a release build.
#ifdef TPGSR_LAB
__device__ int g_wgh_dbg = 0;
extern "C" int tpgsr_wgh_debug(int bits) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_wgh_dbg), &bits, sizeof(bits)) == hipSuccess ? 0 : TPGSR_ERR_LAUNCH;
}
#define WGH_DBG() __builtin_amdgcn_readfirstlane(g_wgh_dbg)
#else
#define WGH_DBG() 0
#endif

template <int LD, int T, int NE>
__global__ __launch_bounds__(768, 3) void conv_wgrad_halo_kernel(tpgsr_wgrad_args w, int M, int Lcap, int Z, HaloDivs dv) {
  extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];   // [2 buffers][T][Lcap entries][64 B], then [2][64] entry table
  const tpgsr_conv_args& a = w.c;
  const int dbg = WGH_DBG();
  const int PLANE = Lcap * 64, BUF = T * PLANE;
  int* etab = reinterpret_cast<int*>(hsm + 2 * BUF);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nnb = (a.Cout + 63) >> 6, NC = a.Cin >> 5;
  const int blk = xcd_remap(blockIdx.x, gridDim.x);     // column tile fastest: the workgroups sharing a halo sit on one XCD
  const int nbk = blk % nnb, cc = (blk / nnb) % NC, z = blk / (nnb * NC);
  const int n0 = nbk * 64;
  const int tiles = (M + 63) >> 6, tpz = (tiles + Z - 1) / Z;
  const int t_beg = z * tpz, t_end = min(tiles, t_beg + tpz);
  const int taps = a.KH * a.KW;
  const int Hp = a.OH + a.KH - 1, Wp = a.OW + a.KW - 1, ohw = a.OH * a.OW;
  const int K = taps * a.Cin;
  auto qbase = [&](int m) __attribute__((always_inline)) {
    const int n = fastdiv(m, dv.ohw), r = m - n * ohw, oh = fastdiv(r, dv.ow);
    return (n * Hp + oh) * Wp + (r - oh * a.OW);
  };

  if (wave >= 8) {
    // ------------------------------- producers (as in the forward kernel; an item = one pixel tile) -------------------------------
    const int pt = tid - 512, aq = pt & 7, er = pt >> 3;
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(a.in, (size_t)a.N * a.H * a.W * a.in_ld);
    const __amdgpu_buffer_rsrc_t rs_in2 = make_rsrc(a.in2 ? a.in2 : a.in, (size_t)a.N * a.H * a.W * a.in2_ld);
    const int c = cc * 32 + aq * 4;
    float4 qs = make_float4(1.f, 1.f, 1.f, 1.f), qt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (LD & 1) {
      qs = *reinterpret_cast<const float4*>(a.in_scale + c);
      qt = *reinterpret_cast<const float4*>(a.in_shift + c);
    }
    int hpix[NE];
    int q0 = 0;
    auto decode_tile = [&](const int t) __attribute__((always_inline)) {
      const int m0 = t * 64;
      q0 = qbase(m0);
      const int L = qbase(min(m0 + 63, M - 1)) - q0 + (a.KH - 1) * Wp + a.KW;
      // this thread's NE CONSECUTIVE entries er * NE ..: walking the padded index space one entry at a time is one compare per entry and
      // the pixel index runs along (the stride-32 walk needed four wrap steps and two multiplies per entry: with every load, store and
      // MFMA of this kernel switched off a tile still took 1.45 us, profiles/r04ae_wgrad_halo_probe.md)
      const int q = q0 + er * NE;
      int n = fastdiv(q, dv.hpwp);
      const int rem = q - n * (Hp * Wp);
      int r = fastdiv(rem, dv.wp), sx = rem - r * Wp;
      int rowbase = (n * a.H + r - a.pad_h) * a.W - a.pad_w;      // pixel index of (n, r - pad_h, 0 - pad_w + sx) = rowbase + sx
      const int img_step = (Hp - a.H) * a.W;
#pragma unroll
      for (int i = 0; i < NE; ++i) {
        const int ih = r - a.pad_h, iw = sx - a.pad_w;
        const bool in = n < a.N && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
        hpix[i] = er * NE + i < L ? (in ? rowbase + sx : -1) : -2;
        const bool c1 = ++sx >= Wp;
        sx = c1 ? 0 : sx;
        r += c1 ? 1 : 0;
        rowbase += c1 ? a.W : 0;
        const bool c2 = r >= Hp;
        r = c2 ? 0 : r;
        n += c2 ? 1 : 0;
        rowbase -= c2 ? img_step : 0;
      }
    };
    constexpr bool DB = !(LD & 4);
    ARaw hr[DB ? 2 : 1][NE];
    int tab[2] = {0, 0};      // this thread's table entry (threads 0..63): entry of tile pixel pt, per register set
    auto load_item = [&](auto set_tag, const int t) __attribute__((always_inline)) {
      constexpr int S = decltype(set_tag)::value;
#pragma unroll
      for (int i = 0; i < NE; ++i) {
        const bool ok = hpix[i] >= 0;
        hr[S][i].ok = ok;
        hr[S][i].raw = hpix[i] > -2;
        hr[S][i].v = buf_load4(rs_in, ok ? ((unsigned)hpix[i] * (unsigned)a.in_ld + (unsigned)(a.in_coff + c)) * 4u : OOB_OFF);
        if (LD & 4) hr[S][i].v2 = buf_load4(rs_in2, ok ? ((unsigned)hpix[i] * (unsigned)a.in2_ld + (unsigned)c) * 4u : OOB_OFF);
      }
      // pixels past the end of the problem point at the last real pixel's entry: finite data, multiplied by dy = 0
      tab[S] = pt < 64 ? qbase(min(t * 64 + pt, M - 1)) - q0 : 0;
    };
    auto store_item = [&](auto set_tag, const int j) __attribute__((always_inline)) {
      constexpr int S = decltype(set_tag)::value;
      unsigned char* buf = hsm + (j & 1) * BUF;
#pragma unroll
      for (int i = 0; i < NE; ++i) {
        const bool part = hr[S][i].raw;
        hr[S][i].raw = false;
        const float4 v = finish_a<LD>(a, hr[S][i], qs, qt);
        uint2 h[T];
        split4<T>(v, h);
        const int e = er * NE + i;
        const int off = e * 64 + (((aq >> 1) ^ ((e >> 2) & 3)) << 4) + (aq & 1) * 8;
        if (part) {
#pragma unroll
          for (int t = 0; t < T; ++t) *reinterpret_cast<uint2*>(buf + t * PLANE + off) = h[t];
        }
      }
      if (pt < 64) etab[(j & 1) * 64 + pt] = tab[S];
    };
    using P0 = std::integral_constant<int, 0>;
    using PN = std::integral_constant<int, DB ? 1 : 0>;
    int t = t_beg, j = 0;
    if (t < t_end) {
      decode_tile(t);
      if (DB) load_item(P0{}, t);
      auto item = [&](auto cur_tag, auto nxt_tag) __attribute__((always_inline)) -> bool {
        const bool more = t + 1 < t_end;
        if (!DB) load_item(cur_tag, t);
        if (more) decode_tile(t + 1);
        if (DB && more && !(dbg & 2)) load_item(nxt_tag, t + 1);
        if (!(dbg & 4)) store_item(cur_tag, j);
        __syncthreads();      // barrier j: tile j is in LDS, and the consumers are done with tile j - 1
        ++j;
        ++t;
        return more;
      };
      while (true) {
        if (!item(P0{}, PN{})) break;
        if (!item(PN{}, P0{})) break;
      }
    }
    return;
  }

  // ------------------------------- consumers -------------------------------
  const int wn = wave & 1, tg = wave >> 1;
  const int TPG = (taps + 3) >> 2;                      // taps per group (<= 3, launcher)
  const int tap0 = tg * TPG, ntap = max(0, min(taps, tap0 + TPG) - tap0);
  int tapoff[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int tp = min(tap0 + i, taps - 1);
    tapoff[i] = (tp / a.KW) * Wp + tp % a.KW;
  }
  const int NB32 = (a.Cout + 31) >> 5, MB16 = (M + 15) >> 4;
  const int nb32 = (n0 >> 5) + wn;
  const bool ncol = nb32 < NB32;
  const size_t plane_y = (size_t)MB16 * NB32 * 1024;    // bytes per term
  const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(reinterpret_cast<const float*>(w.dy_bf), (size_t)T * MB16 * NB32 * 256);
  const bool want_db = w.dbpart != nullptr && cc == 0 && tg == 0;
  // transposing fragment fetch (frag_tr above, on the halo image): 16-lane group G, lane q of the group addresses pixel row
  // (G >> 1) * 8 + (q >> 2) (+4 for the upper half) of the 16-pixel step, channels (G & 1) * 16 + (q & 3) * 4 ..+3
  const int G = lane >> 4, q = lane & 15;
  const int prow = (G >> 1) * 8 + (q >> 2);
  const int c0 = (G & 1) * 16 + (q & 3) * 4;
  auto a_off = [&](const int e) __attribute__((always_inline)) { return e * 64 + (((c0 >> 3) ^ ((e >> 2) & 3)) << 4) + (c0 & 4) * 2; };

  floatx16 acc[3], accdb;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = acc[2][r] = accdb[r] = 0.f;
  u32x4 by[2][T];           // dy fragments of a 16-pixel step, two register sets (one step ahead)
  bf16x8 av[2][T];          // A^T fragments of one (step, tap), two sets (one tap ahead)
  auto fetch_y = [&](auto set_tag, const int mb16) __attribute__((always_inline)) {
    constexpr int S = decltype(set_tag)::value;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const size_t off = t * plane_y + ((size_t)mb16 * NB32 + nb32) * 1024 + lane * 16;
      by[S][t] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, (ncol && mb16 < MB16) ? (int)off : (int)OOB_OFF, 0, 0);
    }
  };
  auto fetch_a = [&](auto set_tag, const unsigned char* buf, const int e_lo, const int e_hi) __attribute__((always_inline)) {
    constexpr int S = decltype(set_tag)::value;
    const int o_lo = a_off(e_lo), o_hi = a_off(e_hi);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(buf + t * PLANE + o_lo));
      s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(buf + t * PLANE + o_hi));
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      av[S][t] = __builtin_bit_cast(bf16x8, v);
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;

  int j = 0;
  if (t_beg < t_end) fetch_y(I0{}, t_beg * 4);
  for (int t = t_beg; t < t_end; ++t, ++j) {
    __syncthreads();          // barrier j
    const unsigned char* buf = hsm + (j & 1) * BUF;
    int e8[4][2];             // entries of this lane's pixel rows: step s, lower / upper half
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      e8[s][0] = etab[(j & 1) * 64 + s * 16 + prow];
      e8[s][1] = etab[(j & 1) * 64 + s * 16 + prow + 4];
    }
    // one 16-pixel step: dy fragments of the next step on their way, then the group's taps; sets alternate by step / tap parity
    auto step = [&](auto ys_tag, const int s, const int mb16) __attribute__((always_inline)) {
      constexpr int YS = decltype(ys_tag)::value;
      fetch_y(std::integral_constant<int, YS ^ 1>{}, mb16 + 1);     // (next tile's first step included; past the end: zeros)
      if (ntap > 0) fetch_a(I0{}, buf, e8[s][0] + tapoff[0], e8[s][1] + tapoff[0]);
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 bv[T];
#pragma unroll
      for (int t2 = 0; t2 < T; ++t2) bv[t2] = __builtin_bit_cast(bf16x8, by[YS][t2]);
      if (want_db) {
#pragma unroll
        for (int t2 = T - 1; t2 >= 0; --t2) accdb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, bv[t2], accdb, 0, 0, 0);
      }
      if (ntap > 0) {
        if (ntap > 1) fetch_a(I1{}, buf, e8[s][0] + tapoff[1], e8[s][1] + tapoff[1]);
        __builtin_amdgcn_sched_barrier(0);
        acc[0] = mfma_terms<T>(av[0], bv, acc[0]);
        __builtin_amdgcn_sched_barrier(0);
        if (ntap > 1) {
          if (ntap > 2) fetch_a(I0{}, buf, e8[s][0] + tapoff[2], e8[s][1] + tapoff[2]);
          __builtin_amdgcn_sched_barrier(0);
          acc[1] = mfma_terms<T>(av[1], bv, acc[1]);
          __builtin_amdgcn_sched_barrier(0);
          if (ntap > 2) {
            acc[2] = mfma_terms<T>(av[0], bv, acc[2]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    };
    if (!(dbg & 1)) {
      step(I0{}, 0, t * 4 + 0);
      step(I1{}, 1, t * 4 + 1);
      step(I0{}, 2, t * 4 + 2);
      step(I1{}, 3, t * 4 + 3);
    }
  }

  // ---- the split's slab: rows k = tap Cin + cc 32 + row, columns n0 + wn 32 + (lane & 31) ----
  const int n = n0 + wn * 32 + (lane & 31);
  float* dst = w.part + (size_t)z * K * a.Cout;
  if (n < a.Cout) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < ntap) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k = (tap0 + i) * a.Cin + cc * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          dst[(size_t)k * a.Cout + n] = acc[i][r];
        }
      }
    if (want_db && lane < 32) w.dbpart[(size_t)z * a.Cout + n] = accdb[0];
  }
}

// geometry test shared by the plan and the launcher
static bool wgrad_halo_shape_ok(const tpgsr_conv_args* a, int* Lcap_out) {
  const int taps = a->KH * a->KW;
  // only where it pays (TPGSR_XBF_WGRAD_HALO_MINWORK overrides the Cin x Cout threshold): measured at batch 48, the recognizer's
  // 128..512-channel convolutions and the 64->256 upsample convolution gain 15-30 % over the tile loop (conv5 225 -> 157 us,
  // upsample 156 -> 121 us) while the 64->64 trunk and 64->128 convolutions lose (42 -> 45 us: two channel blocks x 128 pixel
  // splits, every workgroup writes a slab for six tiles of work, plus the dy pre-split) and the whole C3 step came out 1 % slower
  // with them on -- profiles/r02b_wgrad_halo.md
  static const long long minwork = [] { const char* e = getenv("TPGSR_XBF_WGRAD_HALO_MINWORK"); return e ? atoll(e) : 16384ll; }();
  if ((long long)a->Cin * a->Cout < minwork) return false;
  if (taps < 2 || taps > 12 || (a->Cin & 31) || a->stride_w > 1 || a->in_dil_w > 1 || a->in_ps || a->in_b || a->OW + a->KW - 1 < 8) return false;
  const int Lcap = halo_capacity(a);
  if (Lcap > 32 * 9) return false;
  *Lcap_out = Lcap;
  return true;
}

extern "C" int tpgsr_wgrad_halo_plan(const tpgsr_conv_args* a, int* zsplits, long long* dy_bf_bytes) {
  static const bool on = [] { const char* e = getenv("TPGSR_XBF_WGRAD_HALO"); return !(e && e[0] == '0'); }();
  int Lcap = 0;
  if (!on || !a || a->terms <= 0 || !wgrad_halo_shape_ok(a, &Lcap)) return 0;
  const long long M = (long long)a->N * a->OH * a->OW;
  const int tiles = (int)cdiv(M, 64);
  int cus = 256, dev = 0;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (cus < 1) cus = 256;
  const int groups = (a->Cin >> 5) * cdiv(a->Cout, 64);
  int Z = cus / groups;                                   // one workgroup per CU
  if (Z < 1) Z = 1;
  if (Z > tiles) Z = tiles;
  const int tpz = cdiv(tiles, Z);
  Z = cdiv(tiles, tpz);
  if (zsplits) *zsplits = Z;
  if (dy_bf_bytes) *dy_bf_bytes = 3ll * cdiv(M, 16) * cdiv(a->Cout, 32) * 1024;
  return 1;
}

// returns 1 when launched, 0 when not this kernel's case, < 0 on error
extern "C" int tpgsr_conv_wgrad_halo_launch(const tpgsr_wgrad_args* w, long long M, int ld, hipStream_t st) {
  const tpgsr_conv_args* a = &w->c;
  const int T = a->terms;
  int Lcap = 0;
  if (w->zsplits <= 0 || !w->dy_bf || T <= 0 || (ld & ~7) || ld == 6 || !wgrad_halo_shape_ok(a, &Lcap)) return 0;
  static const bool on = [] { const char* e = getenv("TPGSR_XBF_WGRAD_HALO"); return !(e && e[0] == '0'); }();
  const size_t lds = (size_t)2 * T * Lcap * 64 + 512;
  if (!on || lds > 150 * 1024) return 0;
  const int MB16 = (int)cdiv(M, 16), NB32 = cdiv(a->Cout, 32);
  {
    dim3 g((unsigned)cdiv((long long)MB16 * NB32, 4));
    if (T == 1)
      hipLaunchKernelGGL(dy_split_kernel<1>, g, dim3(256), 0, st, w->dy, w->dy_ld, w->dy_coff, w->dy_ps, (int)M, a->Cout, a->OH, a->OW,
                         (unsigned short*)w->dy_bf, MB16, NB32);
    else if (T == 2)
      hipLaunchKernelGGL(dy_split_kernel<2>, g, dim3(256), 0, st, w->dy, w->dy_ld, w->dy_coff, w->dy_ps, (int)M, a->Cout, a->OH, a->OW,
                         (unsigned short*)w->dy_bf, MB16, NB32);
    else
      hipLaunchKernelGGL(dy_split_kernel<3>, g, dim3(256), 0, st, w->dy, w->dy_ld, w->dy_coff, w->dy_ps, (int)M, a->Cout, a->OH, a->OW,
                         (unsigned short*)w->dy_bf, MB16, NB32);
  }
  const bool small = Lcap <= 32 * 7;
  const void* fn = nullptr;
#define XBF_WGH_CASE(B)                                                                                                          \
  case B:                                                                                                                        \
    fn = T == 1 ? (small ? (const void*)conv_wgrad_halo_kernel<B, 1, 7> : (const void*)conv_wgrad_halo_kernel<B, 1, 9>)          \
       : T == 2 ? (small ? (const void*)conv_wgrad_halo_kernel<B, 2, 7> : (const void*)conv_wgrad_halo_kernel<B, 2, 9>)          \
                : (small ? (const void*)conv_wgrad_halo_kernel<B, 3, 7> : (const void*)conv_wgrad_halo_kernel<B, 3, 9>);         \
    break;
  switch (ld) {
    XBF_HALO_LD_CASES(XBF_WGH_CASE)
    default: return 0;
  }
#undef XBF_WGH_CASE
  if (lds > 64 * 1024) {
    static std::mutex mu;
    static std::vector<std::pair<std::pair<const void*, int>, size_t>> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
      tpgsr_set_error("tpgsr_conv_wgrad: hipGetDevice failed");
      return TPGSR_ERR_LAUNCH;
    }
    std::lock_guard<std::mutex> lock(mu);
    size_t* cur = nullptr;
    for (auto& d : done)
      if (d.first.first == fn && d.first.second == dev) cur = &d.second;
    if (!cur || *cur < lds) {
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        tpgsr_set_error("tpgsr_conv_wgrad: LDS opt-in (%zu bytes) for the halo kernel failed", lds);
        return TPGSR_ERR_LAUNCH;
      }
      if (cur) *cur = lds; else done.push_back({{fn, dev}, lds});
    }
  }
  const int Z = w->zsplits;
  dim3 grid((unsigned)((a->Cin >> 5) * cdiv(a->Cout, 64) * Z));
  int Mi = (int)M, Lc = Lcap, Zi = Z;
  tpgsr_wgrad_args args = *w;
  const int Hp_ = a->OH + a->KH - 1, Wp_ = a->OW + a->KW - 1;
  HaloDivs dv = {fastdiv_make((unsigned)(a->OH * a->OW)), fastdiv_make((unsigned)a->OW), fastdiv_make((unsigned)(Hp_ * Wp_)), fastdiv_make((unsigned)Wp_)};
  void* params[] = {&args, &Mi, &Lc, &Zi, &dv};
  if (hipLaunchKernel(fn, grid, dim3(768), params, lds, st) != hipSuccess) {
    tpgsr_set_error("tpgsr_conv_wgrad(halo): launch failed: %s", hipGetErrorString(hipGetLastError()));
    return TPGSR_ERR_LAUNCH;
  }
  return 1;
}

// ------------------------------------------------------------------------------------------------------
// operand splitting: fp32 [K][ld] (k-major, as packed for the fp32 kernels) -> bf16 planes in MFMA FRAGMENT ORDER
//   dst[((t * NB32 + n / 32) * KB16 + k / 16) * 64 + ((k >> 3) & 1) * 32 + (n & 31)][k & 7],  NB32 = ceil(N / 32), KB16 = Kp / 16,
// Kp = K rounded up to 32, zero padded in k and n: the B operand of one (32-column, 16-k) block is 1 KB contiguous, lane-major.
// One launch for all operands of a network (descriptor table, 64x64 tiles transposed through LDS so both sides stay coalesced).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split_bf_program_kernel(const tpgsr_split_desc* __restrict__ descs, int ndesc) {
  __shared__ float tile[64][65];
  __shared__ int s_d;
  if (threadIdx.x == 0) {
    int lo = 0, hi = ndesc - 1;  // last descriptor whose blk0 <= blockIdx.x
    while (lo < hi) {
      int mid = (lo + hi + 1) >> 1;
      if (descs[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    s_d = lo;
  }
  __syncthreads();
  const tpgsr_split_desc d = descs[s_d];
  const int b = (int)blockIdx.x - d.blk0;
  const int nkb = d.kp / 64 + ((d.kp & 63) ? 1 : 0);
  const int kb = b % nkb, nbk = b / nkb;
  const int k0 = kb * 64, n0 = nbk * 64;
  {  // load: coalesced along n
    const int n = threadIdx.x & 63, kr = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      int k = k0 + kr * 16 + i;                      // destination k (k' order when d.cin > 0)
      if (d.cin > 0 && k < d.K) {
        const int taps = d.K / d.cin, cc = k / (taps * 32), rem = k - cc * taps * 32;
        k = (rem >> 5) * d.cin + cc * 32 + (rem & 31);
      }
      tile[kr * 16 + i][n] = (k < d.K && n0 + n < d.N) ? d.src[(size_t)k * d.ld + n0 + n] : 0.f;
    }
  }
  __syncthreads();
  const int NB32 = (d.N + 31) >> 5, KB16 = d.kp >> 4;
  unsigned short* dst = reinterpret_cast<unsigned short*>(d.dst);
  const size_t plane = (size_t)NB32 * KB16 * 512;            // bf16 elements per term
  for (int i = 0; i < 2; ++i) {
    const int p = threadIdx.x + 256 * i;                      // 64 columns x 8 groups of 8 k
    const int n = p & 63, kg = p >> 6;
    const int k = k0 + kg * 8, nn = n0 + n;
    if (k >= d.kp || nn >= NB32 * 32) continue;
    __bf16 h[8][3];
#pragma unroll
    for (int j = 0; j < 8; ++j) split_bf<3>(tile[kg * 8 + j][n], h[j]);
    const size_t off = ((((size_t)(nn >> 5) * KB16 + (k >> 4)) * 64) + ((k >> 3) & 1) * 32 + (nn & 31)) * 8;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = h[j][t];
      *reinterpret_cast<bf16x8*>(dst + t * plane + off) = v;
    }
  }
}

extern "C" int tpgsr_split_bf_blocks(int K, int N) {
  const int kp = (K + 31) / 32 * 32;
  return cdiv(kp, 64) * cdiv(N, 64);
}

extern "C" int tpgsr_split_bf_program(const tpgsr_split_desc* descs_dev, int ndesc, int total_blocks, void* stream) {
  TPGSR_CHECK_ARG(descs_dev && ndesc > 0 && total_blocks > 0, "tpgsr_split_bf_program: bad arguments");
  hipLaunchKernelGGL(split_bf_program_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, descs_dev, ndesc);
  TPGSR_LAUNCH_CHECK("tpgsr_split_bf_program");
}

// ------------------------------------------------------------------------------------------------------
// diagnostic: what ds_read_b64_tr_b16 delivers.  LDS holds lds[i] = i (as bf16 bit patterns = small integers) for a
// [rows][16] image; every lane reads through frag-style addressing and the four returned halfwords are written out.
// ------------------------------------------------------------------------------------------------------
__global__ void tr_probe_kernel(int* out) {
  __shared__ __attribute__((aligned(16))) unsigned short img[64 * 16];
  for (int i = threadIdx.x; i < 64 * 16; i += 64) img[i] = (unsigned short)i;
  __syncthreads();
  const int lane = threadIdx.x, G = lane >> 4, q = lane & 15;
  const unsigned char* p = reinterpret_cast<const unsigned char*>(img) + ((G * 4 + (q >> 2)) * 16 + (q & 3) * 4) * 2;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (int)(unsigned short)v[j];
}

extern "C" int tpgsr_tr_probe(int* out, void* stream) {
  TPGSR_CHECK_ARG(out != nullptr, "tpgsr_tr_probe: null output");
  hipLaunchKernelGGL(tr_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
  TPGSR_LAUNCH_CHECK("tpgsr_tr_probe");
}


// diagnostic: ONE v_mfma_f32_32x32x16_bf16 on host-given operands: a [32][16] bf16 bits (row i, k), b [16][32] (k, col j), c / d [32][32] f32
__global__ void mfma_bf16_probe_kernel(const unsigned short* a, const unsigned short* b, const float* c, float* d, int reps) {
  const int l = threadIdx.x;
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 av, bv;
  for (int j = 0; j < 8; ++j) {
    const int k = (l >> 5) * 8 + j;
    av[j] = (short)a[(l & 31) * 16 + k];
    bv[j] = (short)b[k * 32 + (l & 31)];
  }
  floatx16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = c[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)];
  for (int it = 0; it < reps; ++it)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) d[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}

extern "C" int tpgsr_mfma_bf16_probe(const void* a, const void* b, const float* c, float* d, int reps, void* stream) {
  TPGSR_CHECK_ARG(a && b && c && d && reps > 0, "tpgsr_mfma_bf16_probe: bad arguments");
  hipLaunchKernelGGL(mfma_bf16_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const unsigned short*)a, (const unsigned short*)b, c, d, reps);
  TPGSR_LAUNCH_CHECK("tpgsr_mfma_bf16_probe");
}
