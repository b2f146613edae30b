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

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

#define XA_ROW 80                 // bytes per LDS row of the fwd images (32 bf16 + 16 B pad)
#define XA_PLANE (64 * XA_ROW)
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
// forward / data-gradient
// ------------------------------------------------------------------------------------------------------
template <int LD, int T>
__global__ __launch_bounds__(256) void conv_fwd_xbf_kernel(tpgsr_conv_args a, int M, int K) {
  __shared__ __attribute__((aligned(16))) unsigned char As[T * XA_PLANE];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[T * XA_PLANE];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int nbn = (a.Cout + BN - 1) / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int mblk = tile / nbn;
  const int m0 = mblk * BM, n0 = (tile - mblk * nbn) * BN;
  const int nchunks = a.kp / KC;

  // A staging: thread -> quad (tid&7) of the chunk, pixels (tid>>3) and (tid>>3)+32  (as conv_fwd_kernel)
  const int aq = tid & 7;
  const int am0 = tid >> 3;
  const PixelPos px0 = decode_pixel(a, m0 + am0, M);
  const PixelPos px1 = decode_pixel(a, m0 + am0 + 32, M);
  // B staging: weight row (output channel) tid>>2, 16-byte part tid&3 of its 64-byte k slice
  const int bn_ = tid >> 2, bpart = tid & 3;

  const int Wr_ = real_w(a);
  const size_t in_floats = a.in_ps ? (size_t)a.N * a.H * a.W * a.Cin : (size_t)a.N * a.H * Wr_ * a.in_ld;
  const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(a.in, in_floats);
  const __amdgpu_buffer_rsrc_t rs_in2 = (LD & 16) ? make_rsrc(a.in_b, (size_t)a.N * Wr_ * a.in_b_ld)
                                                  : make_rsrc(a.in2 ? a.in2 : a.in, (size_t)a.N * a.H * Wr_ * a.in2_ld);
  const int wrows = a.wt_ld > 0 ? a.wt_ld : a.Cout;            // rows per plane of the split operand
  const size_t plane_b = (size_t)wrows * a.kp * 2;               // bytes per plane
  const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(reinterpret_cast<const float*>(a.wt_bf), (plane_b * T + 3) / 4);
  const bool brow_ok = n0 + bn_ < a.Cout;
  const unsigned boff0 = (unsigned)(((size_t)(a.wt_coff + n0 + bn_) * a.kp) * 2 + bpart * 16);

  ARaw qa0, qa1;
  float4 qs = make_float4(1.f, 1.f, 1.f, 1.f), qt = make_float4(0.f, 0.f, 0.f, 0.f);
  u32x4 rb[T];
  KPos kp_ = kpos_init(a, aq);
  auto load_chunk = [&](int ch) {
    qa0 = load_a_raw<LD>(a, rs_in, rs_in2, px0, kp_);
    qa1 = load_a_raw<LD>(a, rs_in, rs_in2, px1, kp_);
    if (LD & 1) {
      qs = *reinterpret_cast<const float4*>(a.in_scale + (kp_.kh < a.KH ? kp_.c : 0));
      qt = *reinterpret_cast<const float4*>(a.in_shift + (kp_.kh < a.KH ? kp_.c : 0));
    }
    kpos_advance(a, kp_, KC);
#pragma unroll
    for (int t = 0; t < T; ++t)
      rb[t] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, brow_ok ? (int)(boff0 + (unsigned)(t * plane_b) + (unsigned)ch * (KC * 2)) : (int)OOB_OFF, 0, 0);
  };
  auto store_chunk = [&]() {
    const float4 v0 = finish_a<LD>(a, qa0, qs, qt);
    const float4 v1 = finish_a<LD>(a, qa1, qs, qt);
    uint2 h0[T], h1[T];
    split4<T>(v0, h0);
    split4<T>(v1, h1);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      *reinterpret_cast<uint2*>(As + t * XA_PLANE + am0 * XA_ROW + aq * 8) = h0[t];
      *reinterpret_cast<uint2*>(As + t * XA_PLANE + (am0 + 32) * XA_ROW + aq * 8) = h1[t];
      *reinterpret_cast<u32x4*>(Bs + t * XA_PLANE + bn_ * XA_ROW + bpart * 16) = rb[t];
    }
  };

  floatx16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  load_chunk(0);
  store_chunk();
  __syncthreads();
  const int g = lane >> 5;
  const int acol = wm * 32 + (lane & 31);     // A row (pixel) of this lane's fragments
  const int bcol = wn * 32 + (lane & 31);     // B row (output channel)
  const unsigned char* ap = As + acol * XA_ROW + g * 16;
  const unsigned char* bp = Bs + bcol * XA_ROW + g * 16;
  for (int ch = 0; ch < nchunks; ++ch) {
    if (ch + 1 < nchunks) load_chunk(ch + 1);
#pragma unroll
    for (int kb = 0; kb < KC / 16; ++kb) {
      bf16x8 av[T], bv[T];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        av[t] = *reinterpret_cast<const bf16x8*>(ap + t * XA_PLANE + kb * 32);
        bv[t] = *reinterpret_cast<const bf16x8*>(bp + t * XA_PLANE + kb * 32);
      }
      acc = mfma_terms<T>(av, bv, acc);
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the split arithmetic / LDS stores of the next tile behind the MFMAs
    __syncthreads();
    if (ch + 1 < nchunks) {
      store_chunk();
      __syncthreads();
    }
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
  if (a.bn_partial) {
    s += __shfl_xor(s, 32);
    ss += __shfl_xor(ss, 32);
    float* red = reinterpret_cast<float*>(Bs);  // all MFMA reads finished behind the loop's final barrier
    if (lane < 32) {
      red[(wm * 2 + 0) * BN + bcol] = s;
      red[(wm * 2 + 1) * BN + bcol] = ss;
    }
    __syncthreads();
    if (tid < BN && n0 + tid < a.Cout) {
      float* dst = a.bn_partial + (size_t)mblk * 2 * a.Cout;
      dst[n0 + tid] = red[0 * BN + tid] + red[2 * BN + tid];
      dst[a.Cout + n0 + tid] = red[1 * BN + tid] + red[3 * BN + tid];
    }
  }
}

// loader variants instantiated for the bf16 path (the same set as the fp32 kernel)
#define XBF_LD_CASES(X) X(0) X(1) X(2) X(3) X(4) X(5) X(7) X(8) X(17)

extern "C" int tpgsr_conv_fwd_xbf_launch(const tpgsr_conv_args* a, long long M, int K, int ld, hipStream_t st) {
  dim3 grid(cdiv(M, BM) * cdiv(a->Cout, BN));
  const int T = a->terms;
#define XBF_FWD_CASE(B)                                                                                 \
  case B:                                                                                               \
    if (T == 1) hipLaunchKernelGGL((conv_fwd_xbf_kernel<B, 1>), grid, dim3(256), 0, st, *a, (int)M, K); \
    else hipLaunchKernelGGL((conv_fwd_xbf_kernel<B, 3>), grid, dim3(256), 0, st, *a, (int)M, K);        \
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
// operand splitting: fp32 [K][ld] (k-major, as packed for the fp32 kernels) -> bf16 planes [3][N][Kp] (k contiguous,
// Kp = K rounded up to 32, zero padded).  One launch for all operands of a network (descriptor table, 64x64 tiles
// transposed through LDS so both sides stay coalesced).
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
  const int kb = b % nkb, nb = b / nkb;
  const int k0 = kb * 64, n0 = nb * 64;
  {  // load: coalesced along n
    const int n = threadIdx.x & 63, kr = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = k0 + kr * 16 + i;
      tile[kr * 16 + i][n] = (k < d.K && n0 + n < d.N) ? d.src[(size_t)k * d.ld + n0 + n] : 0.f;
    }
  }
  __syncthreads();
  const int n = threadIdx.x >> 2, kq = (threadIdx.x & 3) * 16;
  if (n0 + n >= d.N) return;
  unsigned short* dst = reinterpret_cast<unsigned short*>(d.dst);
  const size_t plane = (size_t)d.N * d.kp;
  for (int half = 0; half < 2; ++half) {
    const int kk = k0 + kq + half * 8;
    if (kk >= d.kp) break;                 // kp is a multiple of 32: 8-element groups are all-in or all-out
    __bf16 h[8][3];
#pragma unroll
    for (int j = 0; j < 8; ++j) split_bf<3>(tile[kq + half * 8 + j][n], h[j]);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = h[j][t];
      *reinterpret_cast<bf16x8*>(dst + t * plane + (size_t)(n0 + n) * d.kp + kk) = v;
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
