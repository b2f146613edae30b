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
#include <mutex>

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
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------------
// forward / data-gradient.  BMT pixels x 64 channels per workgroup, BMT / 16 waves each owning a 32 x 32 block (two
// accumulators: the leading term a1 b1, and the five correction terms -- independent MFMA chains, and the corrections are summed
// among themselves before they meet the large sum), K chunks of 32.
//   A (activations, split at run time): through a DOUBLE-buffered LDS image with ONE barrier per chunk -- while the matrix pipe
//     works on chunk c out of buffer c & 1, the same wave splits chunk c + 1 (in registers: its global loads were issued one
//     iteration earlier) into the other buffer and issues the loads of chunk c + 2.  Image per term: [BMT rows][32 k] bf16 =
//     64-byte rows, 16-byte slot s of row r stored at slot s ^ ((r >> 2) & 3): conflict-free ds_read_b128 fragments with no
//     padding (2 x 3 x 8 KB for BMT = 128: two or three workgroups per CU).
//   W (weights, split once per step by tpgsr_split_bf_program): NEVER touches LDS.  The planes are stored in MFMA fragment
//     order [term][n / 32][k / 16][lane][8], so a wave's B operand of one k-block is ONE fully coalesced 1 KB load straight into
//     the registers the MFMA reads.  (Measured on the previous form, which staged W through LDS as well: the LDS pipe was busy
//     46 % of the kernel, the matrix pipe 31 %, next to each other rather than on top of each other; W was half of the LDS
//     traffic and a third of the staging instructions.)
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int xa_off(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4); }

template <int LD, int T, int BMT>
__global__ __launch_bounds__(BMT * 4) void conv_fwd_xbf_kernel(tpgsr_conv_args a, int M, int K) {
  constexpr int A_PLANE = BMT * 64;               // bytes per term
  constexpr int BUF = T * A_PLANE;
  constexpr int WROWS = BMT / 32;                 // wave rows (wave columns: 2)
  __shared__ __attribute__((aligned(16))) unsigned char xsm[2 * BUF];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WROWS, wn = wave / WROWS;
  const int nbn = (a.Cout + BN - 1) / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int mblk = tile / nbn;
  const int m0 = mblk * BMT, n0 = (tile - mblk * nbn) * BN;
  const int nchunks = a.kp / KC;

  // A staging: quad (tid & 7) = 16 of a row's 64 bytes -> half of slot (tid & 7) >> 1; pixels (tid >> 3) and (tid >> 3) + BMT / 2
  const int aq = tid & 7;
  const int am0 = tid >> 3;
  const PixelPos px0 = decode_pixel(a, m0 + am0, M);
  const PixelPos px1 = decode_pixel(a, m0 + am0 + BMT / 2, M);
  const int wofs0 = xa_off(am0, aq >> 1) + (aq & 1) * 8;
  const int wofs1 = xa_off(am0 + BMT / 2, aq >> 1) + (aq & 1) * 8;

  const int Wr_ = real_w(a);
  const size_t in_floats = a.in_ps ? (size_t)a.N * a.H * a.W * a.Cin : (size_t)a.N * a.H * Wr_ * a.in_ld;
  const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(a.in, in_floats);
  const __amdgpu_buffer_rsrc_t rs_in2 = (LD & 16) ? make_rsrc(a.in_b, (size_t)a.N * Wr_ * a.in_b_ld)
                                                  : make_rsrc(a.in2 ? a.in2 : a.in, (size_t)a.N * a.H * Wr_ * a.in2_ld);
  // W fragments: plane t, 32-column block nb, k-block kb16 at ((t * NB32 + nb) * KB16 + kb16) * 1024 bytes (+ lane * 16)
  const int wrows = a.wt_ld > 0 ? a.wt_ld : a.Cout;
  const int NB32 = (wrows + 31) >> 5, KB16 = a.kp >> 4;
  const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(reinterpret_cast<const float*>(a.wt_bf), (size_t)T * NB32 * 32 * a.kp / 2);
  const int nb = (a.wt_coff + n0 + wn * 32) >> 5;
  const bool bok = n0 + wn * 32 < a.Cout;           // a column block entirely past Cout (Cout <= 32 in a 64-wide tile): zeros
  const unsigned plane_w = (unsigned)NB32 * KB16 * 1024u;
  const unsigned woff0 = ((unsigned)nb * KB16) * 1024u + lane * 16u;

  ARaw qa0, qa1;
  float4 qs = make_float4(1.f, 1.f, 1.f, 1.f), qt = make_float4(0.f, 0.f, 0.f, 0.f);
  KPos kp_ = kpos_init(a, aq);
  auto load_chunk = [&]() {
    qa0 = load_a_raw<LD>(a, rs_in, rs_in2, px0, kp_);
    qa1 = load_a_raw<LD>(a, rs_in, rs_in2, px1, kp_);
    if (LD & 1) {
      qs = *reinterpret_cast<const float4*>(a.in_scale + (kp_.kh < a.KH ? kp_.c : 0));
      qt = *reinterpret_cast<const float4*>(a.in_shift + (kp_.kh < a.KH ? kp_.c : 0));
    }
    kpos_advance(a, kp_, KC);
  };
  auto store_chunk = [&](unsigned char* buf) {
    const float4 v0 = finish_a<LD>(a, qa0, qs, qt);
    const float4 v1 = finish_a<LD>(a, qa1, qs, qt);
    uint2 h0[T], h1[T];
    split4<T>(v0, h0);
    split4<T>(v1, h1);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      *reinterpret_cast<uint2*>(buf + t * A_PLANE + wofs0) = h0[t];
      *reinterpret_cast<uint2*>(buf + t * A_PLANE + wofs1) = h1[t];
    }
  };

  floatx16 acc, accl;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = accl[i] = 0.f;

  load_chunk();
  store_chunk(xsm);
  if (nchunks > 1) load_chunk();
  __syncthreads();
  const int g = lane >> 5;
  const int acol = wm * 32 + (lane & 31);
  const int bcol = wn * 32 + (lane & 31);
  const int aoff0 = xa_off(acol, g), aoff1 = xa_off(acol, 2 + g);   // the two k-blocks of a chunk
  for (int ch = 0; ch < nchunks; ++ch) {
    const unsigned char* cur = xsm + (ch & 1) * BUF;
    unsigned char* nxt = xsm + ((ch + 1) & 1) * BUF;
    u32x4 bw[2][T];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int t = 0; t < T; ++t)
        bw[kb][t] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, bok ? (int)(woff0 + t * plane_w + (unsigned)(ch * 2 + kb) * 1024u) : (int)OOB_OFF, 0, 0);
    if (ch + 1 < nchunks) {
      store_chunk(nxt);                        // chunk ch + 1: in registers since the previous iteration
      if (ch + 2 < nchunks) load_chunk();
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      bf16x8 av[T], bv[T];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        av[t] = *reinterpret_cast<const bf16x8*>(cur + t * A_PLANE + (kb ? aoff1 : aoff0));
        bv[t] = __builtin_bit_cast(bf16x8, bw[kb][t]);
      }
      if (T == 3) {
        accl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bv[2], accl, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bv[0], acc, 0, 0, 0);
        accl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[2], bv[0], accl, 0, 0, 0);
        accl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[1], bv[1], accl, 0, 0, 0);
        accl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bv[1], accl, 0, 0, 0);
        accl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[1], bv[0], accl, 0, 0, 0);
      } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bv[0], acc, 0, 0, 0);
      }
    }
    __syncthreads();
  }
  if (T == 3) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] += accl[i];
  }

  // ---- epilogue (as conv_fwd_kernel): bias, activation, (pixel-shuffled) store, BN partial statistics ----
  const int n = n0 + bcol;
  const bool nvalid = n < a.Cout;
  const float bias = (a.bias && nvalid) ? a.bias[n] : 0.f;
  float s = 0.f, ss = 0.f;
  const int ohw = a.OH * a.OW;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    int m = m0 + wm * 32 + row;
    if (m < M && nvalid) {
      float raw = acc[r];
      s += raw;
      ss += raw * raw;
      float v = apply_act(raw + bias, a.out_act);
      if (!a.out_ps) {
        a.out[(size_t)m * a.out_ld + a.out_coff + n] = v;
      } else {
        int nn = m / ohw;
        int rem = m - nn * ohw;
        int oh = rem / a.OW, ow = rem - oh * a.OW;
        int cs = n >> 2, i = (n >> 1) & 1, j = n & 1;
        a.out[((size_t)(nn * 2 * a.OH + 2 * oh + i) * (2 * a.OW) + 2 * ow + j) * (a.Cout >> 2) + cs] = v;
      }
    }
  }
  if (a.bn_partial) {   // statistics per 64-pixel row block (the layout bn_finalize expects): wave rows 2j, 2j+1 form block j
    s += __shfl_xor(s, 32);
    ss += __shfl_xor(ss, 32);
    float* red = reinterpret_cast<float*>(xsm);     // [WROWS][2][64]; all LDS reads are behind the loop's last barrier
    if (lane < 32) {
      red[(wm * 2 + 0) * BN + bcol] = s;
      red[(wm * 2 + 1) * BN + bcol] = ss;
    }
    __syncthreads();
    constexpr int NBLK = BMT / 64;
    if (tid < BN * NBLK) {
      const int blk = tid / BN, c = tid - blk * BN;
      const long long rb64 = (long long)mblk * NBLK + blk;
      if (n0 + c < a.Cout && rb64 * 64 < M) {
        float* dst = a.bn_partial + (size_t)rb64 * 2 * a.Cout;
        dst[n0 + c] = red[((2 * blk) * 2 + 0) * BN + c] + red[((2 * blk + 1) * 2 + 0) * BN + c];
        dst[a.Cout + n0 + c] = red[((2 * blk) * 2 + 1) * BN + c] + red[((2 * blk + 1) * 2 + 1) * BN + c];
      }
    }
  }
}

// loader variants instantiated for the bf16 path (the same set as the fp32 kernel)
#define XBF_LD_CASES(X) X(0) X(1) X(2) X(3) X(4) X(5) X(7) X(8) X(17)

extern "C" int tpgsr_conv_fwd_xbf_launch(const tpgsr_conv_args* a, long long M, int K, int ld, hipStream_t st) {
  const int T = a->terms;
  // tile height: 64 pixels.  The 128-pixel variant (TPGSR_XBF_TILE=128) measured equal or slower on every layer shape of
  // the TSRN / CRNN step at batch 48 (profiles/r02_conv_prec_tiles.md): fewer, fatter workgroups lose more to the tail than
  // the halved W traffic wins; it is kept for larger problems.
  static const int force = [] { const char* e = getenv("TPGSR_XBF_TILE"); return e ? atoi(e) : 0; }();
  const bool big = force == 128;
  dim3 grid(cdiv(M, big ? 128 : 64) * cdiv(a->Cout, BN));
#define XBF_FWD_CASE(B)                                                                                          \
  case B:                                                                                                        \
    if (T == 1) {                                                                                                \
      if (big) hipLaunchKernelGGL((conv_fwd_xbf_kernel<B, 1, 128>), grid, dim3(512), 0, st, *a, (int)M, K);      \
      else hipLaunchKernelGGL((conv_fwd_xbf_kernel<B, 1, 64>), grid, dim3(256), 0, st, *a, (int)M, K);           \
    } else {                                                                                                     \
      if (big) hipLaunchKernelGGL((conv_fwd_xbf_kernel<B, 3, 128>), grid, dim3(512), 0, st, *a, (int)M, K);      \
      else hipLaunchKernelGGL((conv_fwd_xbf_kernel<B, 3, 64>), grid, dim3(256), 0, st, *a, (int)M, K);           \
    }                                                                                                            \
    break;
  switch (ld) {
    XBF_LD_CASES(XBF_FWD_CASE)
    default:
      tpgsr_set_error("tpgsr_conv_fwd: unsupported loader combination %d", ld);
      return TPGSR_ERR_ARG;
  }
#undef XBF_FWD_CASE
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

template <int LD, int T>
__global__ __launch_bounds__(256) void conv_wgrad_xbf_kernel(tpgsr_wgrad_args w, int M, int K, int MB) {
  __shared__ __attribute__((aligned(16))) unsigned char Am[T * XW_PLANE];
  __shared__ __attribute__((aligned(16))) unsigned char Ym[T * XW_PLANE];
  const tpgsr_conv_args& a = w.c;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wk = wave & 1, wn = wave >> 1;
  const int nkb = (K + WK - 1) / WK, nnb = (a.Cout + BN - 1) / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);       // k-block fastest: the k-blocks of one pixel split share an L2
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

extern "C" int tpgsr_conv_wgrad_xbf_launch(const tpgsr_wgrad_args* w, long long M, int K, int Z, int MB, int ld, hipStream_t st) {
  const tpgsr_conv_args* a = &w->c;
  dim3 grid(cdiv(K, WK) * cdiv(a->Cout, BN) * Z);
  const int T = a->terms;
#define XBF_WG_CASE(B)                                                                                       \
  case B:                                                                                                    \
    if (T == 1) hipLaunchKernelGGL((conv_wgrad_xbf_kernel<B, 1>), grid, dim3(256), 0, st, *w, (int)M, K, MB); \
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
      const int k = k0 + kr * 16 + i;
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
