// Text-prior generator pieces (CRNN, model/crnn/crnn.py:29-90, and its callers in interfaces/):
//   * parse_crnn_data (interfaces/base.py:806-829): bicubic resize of RGB to 32x100 + luminance, forward / backward
//   * general max-pool with the producer's BN affine + ReLU folded in (crnn.py:56-66 pooling0..3)
//   * BiLSTM time-step gate kernels (nn.LSTM, crnn.py:10); the recurrent GEMMs run on the MFMA conv kernel
//   * softmax over the 37 classes + SemanticLoss (loss/semantic_loss.py:21-39) + the (N,37,1,26) prior with the
//     deterministic prior dropout of interfaces/super_resolution.py:376-382
#include "common.h"
#include "gru_common.h"   // the recurrences' gate functions: compensated v_exp_f32 + v_rcp_f32 with a Newton step (<= 4.5 ulp, a third of libm's instructions)
#include <stdlib.h>
#include <type_traits>

// ------------------------------------------------------------------------------------------------------
// bicubic (A = -0.75, align_corners = False, no antialias) + luminance
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cubic_coeffs(float t, float (&w)[4]) {
  const float A = -0.75f;
  float x = t + 1.f;
  w[0] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
  x = t;
  w[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  x = 1.f - t;
  w[2] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  x = 2.f - t;
  w[3] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
}

__device__ __forceinline__ void bicubic_src(int o, int in_size, int out_size, int& i0, float (&w)[4]) {
  float scale = (float)in_size / (float)out_size;
  float x = ((float)o + 0.5f) * scale - 0.5f;
  float fx = floorf(x);
  i0 = (int)fx;
  cubic_coeffs(x - fx, w);
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__global__ __launch_bounds__(256) void bicubic_gray_fwd_kernel(const float* __restrict__ in, int N, int Ctot, int H, int W, int OH,
                                                               int OW, float* __restrict__ out) {
  long long total = (long long)N * OH * OW;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int ow = (int)(i % OW);
  long long r = i / OW;
  int oh = (int)(r % OH);
  int n = (int)(r / OH);
  int y0, x0;
  float wy[4], wx[4];
  bicubic_src(oh, H, OH, y0, wy);
  bicubic_src(ow, W, OW, x0, wx);
  const float lum[3] = {0.299f, 0.587f, 0.114f};
  float g = 0.f;
  for (int c = 0; c < 3; ++c) {
    const float* p = in + ((size_t)n * Ctot + c) * H * W;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      int yy = clampi(y0 - 1 + a, 0, H - 1);
      float rowv = 0.f;
#pragma unroll
      for (int b = 0; b < 4; ++b) rowv += p[(size_t)yy * W + clampi(x0 - 1 + b, 0, W - 1)] * wx[b];
      acc += rowv * wy[a];
    }
    g += lum[c] * acc;
  }
  out[i] = g;
}

extern "C" int tpgsr_bicubic_gray_fwd(const float* in_nchw, int N, int Ctot, int H, int W, int OH, int OW, float* out, void* stream) {
  TPGSR_CHECK_ARG(in_nchw && out && N > 0 && Ctot >= 3 && H > 0 && W > 0 && OH > 0 && OW > 0, "tpgsr_bicubic_gray_fwd: bad arguments");
  long long total = (long long)N * OH * OW;
  hipLaunchKernelGGL(bicubic_gray_fwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, in_nchw, N, Ctot, H, W, OH, OW, out);
  TPGSR_LAUNCH_CHECK("tpgsr_bicubic_gray_fwd");
}

// Gather form of the adjoint (deterministic: every input pixel sums, in a fixed order, the output pixels whose 4x4
// footprint -- border-clamped like the forward -- touches it).  One thread per (n, y, x); the three colour channels
// differ only by the luminance weight, further channels (the mask) receive zero.
__device__ __forceinline__ float bicubic_tap_weight(int o, int in_size, int out_size, int target) {
  int i0;
  float w[4];
  bicubic_src(o, in_size, out_size, i0, w);
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a) s += clampi(i0 - 1 + a, 0, in_size - 1) == target ? w[a] : 0.f;
  return s;
}
// conservative range of output indices whose (unclamped) taps [src-2, src+3) can reach input index i (border indices also
// collect the clamped taps, which lie inside the same range)
__device__ __forceinline__ void bicubic_out_range(int i, int in_size, int out_size, int& lo, int& hi) {
  float inv = (float)out_size / (float)in_size;
  lo = (int)floorf(((float)i - 3.f + 0.5f) * inv - 0.5f) - 1;
  hi = (int)ceilf(((float)i + 3.f + 0.5f) * inv - 0.5f) + 1;
  lo = lo < 0 ? 0 : lo;
  hi = hi > out_size - 1 ? out_size - 1 : hi;
}

__global__ __launch_bounds__(256) void bicubic_gray_bwd_kernel(const float* __restrict__ dout, int N, int Ctot, int H, int W, int OH,
                                                               int OW, float* __restrict__ din) {
  long long total = (long long)N * H * W;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int x = (int)(i % W);
  long long r = i / W;
  int y = (int)(r % H);
  int n = (int)(r / H);
  int oh_lo, oh_hi, ow_lo, ow_hi;
  bicubic_out_range(y, H, OH, oh_lo, oh_hi);
  bicubic_out_range(x, W, OW, ow_lo, ow_hi);
  float s = 0.f;
  for (int oh = oh_lo; oh <= oh_hi; ++oh) {
    const float wy = bicubic_tap_weight(oh, H, OH, y);
    if (wy == 0.f) continue;
    const float* row = dout + ((size_t)n * OH + oh) * OW;
    float rs = 0.f;
    for (int ow = ow_lo; ow <= ow_hi; ++ow) {
      const float wx = bicubic_tap_weight(ow, W, OW, x);
      rs += wx * row[ow];
    }
    s += wy * rs;
  }
  const float lum[3] = {0.299f, 0.587f, 0.114f};
  for (int c = 0; c < Ctot; ++c) din[(((size_t)n * Ctot + c) * H + y) * W + x] = c < 3 ? lum[c] * s : 0.f;
}

extern "C" int tpgsr_bicubic_gray_bwd(const float* dout, int N, int Ctot, int H, int W, int OH, int OW, float* din_nchw, void* stream) {
  TPGSR_CHECK_ARG(dout && din_nchw && N > 0 && Ctot >= 3 && H > 0 && W > 0 && OH > 0 && OW > 0, "tpgsr_bicubic_gray_bwd: bad arguments");
  long long total = (long long)N * H * W;
  hipLaunchKernelGGL(bicubic_gray_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, dout, N, Ctot, H, W, OH, OW, din_nchw);
  TPGSR_LAUNCH_CHECK("tpgsr_bicubic_gray_bwd");
}

// ------------------------------------------------------------------------------------------------------
// CRNN conv0 (1 -> 64 channels, 3x3, pad 1; crnn.py:45): with Cin = 1 the implicit-GEMM loader would gather single
// floats, so the 3x3 neighbourhood is written out once as a 12-channel map (9 taps + 3 zero channels) and the conv runs
// as a 1x1 conv with Cin = 12 on the vector loader.  col2im is the gather-form adjoint (for d gray of later cascade stages).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void im2col3x3_c1_kernel(const float* __restrict__ in, int N, int H, int W, float* __restrict__ col) {
  long long total = (long long)N * H * W;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int x = (int)(i % W);
  long long r = i / W;
  int y = (int)(r % H);
  int n = (int)(r / H);
  const float* p = in + (size_t)n * H * W;
  float v[12];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      int yy = y + kh - 1, xx = x + kw - 1;
      v[kh * 3 + kw] = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? p[(size_t)yy * W + xx] : 0.f;
    }
  v[9] = v[10] = v[11] = 0.f;
  float4* d = reinterpret_cast<float4*>(col + i * 12);
  d[0] = make_float4(v[0], v[1], v[2], v[3]);
  d[1] = make_float4(v[4], v[5], v[6], v[7]);
  d[2] = make_float4(v[8], v[9], v[10], v[11]);
}

extern "C" int tpgsr_im2col3x3_c1(const float* in, int N, int H, int W, float* col, void* stream) {
  TPGSR_CHECK_ARG(in && col && N > 0 && H > 0 && W > 0 && ((uintptr_t)col & 15) == 0, "tpgsr_im2col3x3_c1: bad arguments");
  long long total = (long long)N * H * W;
  hipLaunchKernelGGL(im2col3x3_c1_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, in, N, H, W, col);
  TPGSR_LAUNCH_CHECK("tpgsr_im2col3x3_c1");
}

__global__ __launch_bounds__(256) void col2im3x3_c1_kernel(const float* __restrict__ dcol, int N, int H, int W, float* __restrict__ din) {
  long long total = (long long)N * H * W;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int x = (int)(i % W);
  long long r = i / W;
  int y = (int)(r % H);
  int n = (int)(r / H);
  float s = 0.f;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      int oy = y - kh + 1, ox = x - kw + 1;   // output pixel whose tap (kh, kw) read (y, x)
      if ((unsigned)oy < (unsigned)H && (unsigned)ox < (unsigned)W) s += dcol[(((size_t)n * H + oy) * W + ox) * 12 + kh * 3 + kw];
    }
  din[i] = s;
}

extern "C" int tpgsr_col2im3x3_c1(const float* dcol, int N, int H, int W, float* din, void* stream) {
  TPGSR_CHECK_ARG(dcol && din && N > 0 && H > 0 && W > 0, "tpgsr_col2im3x3_c1: bad arguments");
  long long total = (long long)N * H * W;
  hipLaunchKernelGGL(col2im3x3_c1_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, dcol, N, H, W, din);
  TPGSR_LAUNCH_CHECK("tpgsr_col2im3x3_c1");
}

// dst[m][c] = c < Cs ? src[m][c] : 0  (Cd >= Cs): pads a channel count that is not a multiple of 4 (the 37 classes of the
// recogniser / text prior) so the consumers stay on the vector loaders
__global__ __launch_bounds__(256) void pad_channels_kernel(const float* __restrict__ src, long long M, int Cs, int Cd, float* __restrict__ dst) {
  long long total = M * Cd;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int c = (int)(i % Cd);
    long long m = i / Cd;
    dst[i] = c < Cs ? src[m * Cs + c] : 0.f;
  }
}

extern "C" int tpgsr_pad_channels(const float* src, long long M, int Cs, int Cd, float* dst, void* stream) {
  TPGSR_CHECK_ARG(src && dst && M > 0 && Cs > 0 && Cd >= Cs, "tpgsr_pad_channels: bad arguments");
  int grid = (int)min((long long)4096, (M * Cd + 255) / 256);
  hipLaunchKernelGGL(pad_channels_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, M, Cs, Cd, dst);
  TPGSR_LAUNCH_CHECK("tpgsr_pad_channels");
}

// ------------------------------------------------------------------------------------------------------
// general max-pool (kernel KHxKW, stride SHxSW, zero... -inf padding PHxPW) of act(scale*x+shift), NHWC
// ------------------------------------------------------------------------------------------------------
// one thread = VEC consecutive channels of one output position (VEC = 4 when C % 4 == 0: 16-byte accesses, the window arithmetic shared
// by the four channels, 32-bit index decoding -- the scalar form with its 64-bit divisions per element was 13-20 us per launch, four
// launches in the text-prior generator's forward pass and four in the teacher's)
template <int VEC>
__global__ __launch_bounds__(256) void pool2d_fwd_kernel(const float* __restrict__ x, int N, int H, int W, int C,
                                                         const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                                         int KH, int KW, int SH, int SW, int PH, int PW, int OH, int OW,
                                                         float* __restrict__ out) {
  const int CV = C / VEC;
  const unsigned total = (unsigned)N * OH * OW * CV;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    unsigned p = i / (unsigned)CV;
    const int c = (int)(i - p * CV) * VEC;
    const unsigned p2 = p / (unsigned)OW;
    const int ow = (int)(p - p2 * OW);
    const int n = (int)(p2 / (unsigned)OH);
    const int oh = (int)(p2 - (unsigned)n * OH);
    float sc[VEC], sh[VEC], best[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      sc[v] = scale ? scale[c + v] : 1.f;
      sh[v] = shift ? shift[c + v] : 0.f;
      best[v] = -INFINITY;
    }
    for (int a = 0; a < KH; ++a) {
      const int h = oh * SH - PH + a;
      if ((unsigned)h >= (unsigned)H) continue;
      for (int b = 0; b < KW; ++b) {
        const int w = ow * SW - PW + b;
        if ((unsigned)w >= (unsigned)W) continue;
        const size_t o = ((size_t)(n * H + h) * W + w) * C + c;
        float xv[VEC];
        if (VEC == 4) {
          const float4 t = *reinterpret_cast<const float4*>(x + o);
          xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
        } else {
          xv[0] = x[o];
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const float val = apply_act(xv[v] * sc[v] + sh[v], act);
          if (val > best[v] || val != val) best[v] = val;
        }
      }
    }
    if (VEC == 4) *reinterpret_cast<float4*>(out + (size_t)i * 4) = make_float4(best[0], best[1], best[2], best[3]);
    else out[i] = best[0];
  }
}

extern "C" int tpgsr_pool2d_fwd(const float* x, int N, int H, int W, int C, const float* scale, const float* shift, int act, int KH,
                                int KW, int SH, int SW, int PH, int PW, float* out, void* stream) {
  TPGSR_CHECK_ARG(x && out && KH > 0 && KW > 0 && SH > 0 && SW > 0 && PH >= 0 && PW >= 0, "tpgsr_pool2d_fwd: bad arguments");
  int OH = (H + 2 * PH - KH) / SH + 1, OW = (W + 2 * PW - KW) / SW + 1;
  TPGSR_CHECK_ARG(OH > 0 && OW > 0, "tpgsr_pool2d_fwd: empty output");
  TPGSR_CHECK_ARG((long long)N * H * W * C < (1ll << 31) && (long long)N * OH * OW * C < (1ll << 31), "tpgsr_pool2d_fwd: map too large for the kernel's 32-bit indices");
  const bool vec = (C & 3) == 0 && ((((uintptr_t)x | (uintptr_t)out) & 15) == 0);
  long long total = (long long)N * OH * OW * (vec ? C / 4 : C);
  int grid = (int)min((long long)8192, (total + 255) / 256);
  if (vec)
    hipLaunchKernelGGL(pool2d_fwd_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, N, H, W, C, scale, shift, act, KH, KW, SH, SW,
                       PH, PW, OH, OW, out);
  else
    hipLaunchKernelGGL(pool2d_fwd_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, N, H, W, C, scale, shift, act, KH, KW, SH, SW,
                       PH, PW, OH, OW, out);
  TPGSR_LAUNCH_CHECK("tpgsr_pool2d_fwd");
}

// gather form: dz[n][h][w][c] = act'(pre) * sum over windows that contain (h,w) and whose FIRST arg-max is (h,w) of dout
// one thread = VEC consecutive channels of one input position (VEC = 4 when C % 4 == 0: 16-byte accesses, the window
// arithmetic shared by the four channels; the scalar form was 91 us per launch on the recognizer's 10 M-element maps)
template <int VEC>
__global__ __launch_bounds__(256) void pool2d_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dout, int N, int H,
                                                         int W, int C, const float* __restrict__ scale, const float* __restrict__ shift,
                                                         int act, int KH, int KW, int SH, int SW, int PH, int PW, int OH, int OW,
                                                         float* __restrict__ dz) {
  const int CV = C / VEC;
  long long total = (long long)N * H * W * CV;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int c = (int)(i % CV) * VEC;
    long long p = i / CV;
    int w = (int)(p % W);
    p /= W;
    int h = (int)(p % H);
    int n = (int)(p / H);
    float sc[VEC], sh[VEC], pre[VEC], mine[VEC], g[VEC], xv[VEC];
    const size_t self = ((size_t)(n * H + h) * W + w) * C + c;
    if (VEC == 4) {
      const float4 t = *reinterpret_cast<const float4*>(x + self);
      xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
    } else {
      xv[0] = x[self];
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      sc[v] = scale ? scale[c + v] : 1.f;
      sh[v] = shift ? shift[c + v] : 0.f;
      pre[v] = xv[v] * sc[v] + sh[v];
      mine[v] = apply_act(pre[v], act);
      g[v] = 0.f;
    }
    // windows (oh, ow) with oh*SH - PH <= h < oh*SH - PH + KH
    int oh_lo = (h + PH - KH + SH) / SH;
    if (h + PH - KH + 1 <= 0) oh_lo = 0;
    int oh_hi = min(OH - 1, (h + PH) / SH);
    int ow_lo = (w + PW - KW + SW) / SW;
    if (w + PW - KW + 1 <= 0) ow_lo = 0;
    int ow_hi = min(OW - 1, (w + PW) / SW);
    for (int oh = max(oh_lo, 0); oh <= oh_hi; ++oh)
      for (int ow = max(ow_lo, 0); ow <= ow_hi; ++ow) {
        // is (h, w) the first arg-max of window (oh, ow)?  (per channel)
        bool first[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) first[v] = true;
        for (int a = 0; a < KH; ++a) {
          int hh = oh * SH - PH + a;
          if ((unsigned)hh >= (unsigned)H) continue;
          for (int b = 0; b < KW; ++b) {
            int ww = ow * SW - PW + b;
            if ((unsigned)ww >= (unsigned)W) continue;
            if (hh == h && ww == w) continue;
            const size_t o = ((size_t)(n * H + hh) * W + ww) * C + c;
            float nv[VEC];
            if (VEC == 4) {
              const float4 t = *reinterpret_cast<const float4*>(x + o);
              nv[0] = t.x; nv[1] = t.y; nv[2] = t.z; nv[3] = t.w;
            } else {
              nv[0] = x[o];
            }
            const bool before = (hh < h) || (hh == h && ww < w);
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
              const float val = apply_act(nv[v] * sc[v] + sh[v], act);
              if (val > mine[v] || (before && val == mine[v])) first[v] = false;
            }
          }
        }
        const size_t od = ((size_t)(n * OH + oh) * OW + ow) * C + c;
        if (VEC == 4) {
          const float4 t = *reinterpret_cast<const float4*>(dout + od);
          g[0] += first[0] ? t.x : 0.f; g[1] += first[1] ? t.y : 0.f; g[2] += first[2] ? t.z : 0.f; g[3] += first[3] ? t.w : 0.f;
        } else {
          g[0] += first[0] ? dout[od] : 0.f;
        }
      }
    if (VEC == 4) {
      float4 o4;
      o4.x = g[0] * act_grad(pre[0], act); o4.y = g[1] * act_grad(pre[1], act);
      o4.z = g[2] * act_grad(pre[2], act); o4.w = g[3] * act_grad(pre[3], act);
      *reinterpret_cast<float4*>(dz + self) = o4;
    } else {
      dz[self] = g[0] * act_grad(pre[0], act);
    }
  }
}

// 2 x 2 windows, stride 2, no padding, even H and W (pooling0 / pooling1 of the recogniser, crnn.py:56-59: its two largest maps): the
// windows do not overlap, so one thread owns one WINDOW x 4 channels -- every input is read once and every gradient written once
// (the gather form above re-reads each window from all four of its positions: 73 / 55 us per launch at batch 48).  First arg-max in
// row-major scan order, as the gather form and ATen.
__global__ __launch_bounds__(256) void pool2x2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dout, int N, int H,
                                                          int W, int C, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int act, float* __restrict__ dz) {
  const int C4 = C >> 2, OH = H >> 1, OW = W >> 1;
  const unsigned total = (unsigned)N * OH * OW * C4;      // (< 2^31: the launcher checks; 64-bit divisions were a third of this kernel)
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    unsigned p = i / (unsigned)C4;
    const int c = (int)(i - p * C4) * 4;
    const unsigned p2 = p / (unsigned)OW;
    const int ow = (int)(p - p2 * OW);
    const int n = (int)(p2 / (unsigned)OH);
    const int oh = (int)(p2 - (unsigned)n * OH);
    const size_t base = ((size_t)(n * H + 2 * oh) * W + 2 * ow) * C + c;
    const size_t offs[4] = {0, (size_t)C, (size_t)W * C, (size_t)W * C + C};
    float4 xv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) xv[k] = *reinterpret_cast<const float4*>(x + base + offs[k]);
    const float4 g = *reinterpret_cast<const float4*>(dout + ((size_t)(n * OH + oh) * OW + ow) * C + c);
    const float4 sc = scale ? *reinterpret_cast<const float4*>(scale + c) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 sh = shift ? *reinterpret_cast<const float4*>(shift + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 out[4];
    auto one = [&](float x0, float x1, float x2, float x3, float s, float t, float gg, float& o0, float& o1, float& o2, float& o3)
                   __attribute__((always_inline)) {
      const float p0 = x0 * s + t, p1 = x1 * s + t, p2 = x2 * s + t, p3 = x3 * s + t;
      const float v0 = apply_act(p0, act), v1 = apply_act(p1, act), v2 = apply_act(p2, act), v3 = apply_act(p3, act);
      int k = 0;
      float best = v0;
      if (v1 > best) { best = v1; k = 1; }
      if (v2 > best) { best = v2; k = 2; }
      if (v3 > best) { best = v3; k = 3; }
      o0 = k == 0 ? gg * act_grad(p0, act) : 0.f;
      o1 = k == 1 ? gg * act_grad(p1, act) : 0.f;
      o2 = k == 2 ? gg * act_grad(p2, act) : 0.f;
      o3 = k == 3 ? gg * act_grad(p3, act) : 0.f;
    };
    one(xv[0].x, xv[1].x, xv[2].x, xv[3].x, sc.x, sh.x, g.x, out[0].x, out[1].x, out[2].x, out[3].x);
    one(xv[0].y, xv[1].y, xv[2].y, xv[3].y, sc.y, sh.y, g.y, out[0].y, out[1].y, out[2].y, out[3].y);
    one(xv[0].z, xv[1].z, xv[2].z, xv[3].z, sc.z, sh.z, g.z, out[0].z, out[1].z, out[2].z, out[3].z);
    one(xv[0].w, xv[1].w, xv[2].w, xv[3].w, sc.w, sh.w, g.w, out[0].w, out[1].w, out[2].w, out[3].w);
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<float4*>(dz + base + offs[k]) = out[k];
  }
}

// 2 x 2 windows, stride (2, 1), padding (0, 1), even H (pooling2 / pooling3 of the recogniser, crnn.py:60-66): the windows overlap along
// W only -- input column w sits in window ow = w (columns w-1, w) and in window ow = w+1 (columns w, w+1) of its row pair.  One thread
// owns the two inputs (2 oh, w), (2 oh + 1, w) x 4 channels: six independent input loads + two gradient loads, no loop (the gather
// form above walks windows and positions with dependent loads: 40 / 60 us per launch at batch 48 for 10 / 20 MB of traffic).
__global__ __launch_bounds__(256) void pool2x2s21_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dout, int N, int H,
                                                             int W, int C, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, int act, float* __restrict__ dz) {
  const int C4 = C >> 2, OH = H >> 1, OW = W + 1;
  const unsigned total = (unsigned)N * OH * W * C4;       // (< 2^31: the launcher checks)
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    unsigned p = i / (unsigned)C4;
    const int c = (int)(i - p * C4) * 4;
    const unsigned p2 = p / (unsigned)W;
    const int w = (int)(p - p2 * W);
    const int n = (int)(p2 / (unsigned)OH);
    const int oh = (int)(p2 - (unsigned)n * OH);
    const size_t r0 = ((size_t)(n * H + 2 * oh) * W + w) * C + c, r1 = r0 + (size_t)W * C;
    const bool hasl = w > 0, hasr = w + 1 < W;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 xv[2][3];     // [row][left, mine, right]
    xv[0][1] = *reinterpret_cast<const float4*>(x + r0);
    xv[1][1] = *reinterpret_cast<const float4*>(x + r1);
    xv[0][0] = hasl ? *reinterpret_cast<const float4*>(x + r0 - C) : zero;
    xv[1][0] = hasl ? *reinterpret_cast<const float4*>(x + r1 - C) : zero;
    xv[0][2] = hasr ? *reinterpret_cast<const float4*>(x + r0 + C) : zero;
    xv[1][2] = hasr ? *reinterpret_cast<const float4*>(x + r1 + C) : zero;
    const size_t od = ((size_t)(n * OH + oh) * OW + w) * C + c;
    const float4 ga = *reinterpret_cast<const float4*>(dout + od);         // window ow = w
    const float4 gb = *reinterpret_cast<const float4*>(dout + od + C);     // window ow = w + 1
    const float4 sc = scale ? *reinterpret_cast<const float4*>(scale + c) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 sh = shift ? *reinterpret_cast<const float4*>(shift + c) : zero;
    auto one = [&](float l0, float m0, float q0, float l1, float m1, float q1, float s, float t, float gA, float gB, float& o0, float& o1)
                   __attribute__((always_inline)) {
      const float pm0 = m0 * s + t, pm1 = m1 * s + t;
      const float vm0 = apply_act(pm0, act), vm1 = apply_act(pm1, act);
      const float vl0 = hasl ? apply_act(l0 * s + t, act) : -INFINITY, vl1 = hasl ? apply_act(l1 * s + t, act) : -INFINITY;
      const float vr0 = hasr ? apply_act(q0 * s + t, act) : -INFINITY, vr1 = hasr ? apply_act(q1 * s + t, act) : -INFINITY;
      // window A (columns w-1, w), scan order (r0, w-1), (r0, w), (r1, w-1), (r1, w): first arg-max
      int ka = 0;
      float best = vl0;
      if (vm0 > best) { best = vm0; ka = 1; }
      if (vl1 > best) { best = vl1; ka = 2; }
      if (vm1 > best) { best = vm1; ka = 3; }
      // window B (columns w, w+1): (r0, w), (r0, w+1), (r1, w), (r1, w+1)
      int kb = 0;
      best = vm0;
      if (vr0 > best) { best = vr0; kb = 1; }
      if (vm1 > best) { best = vm1; kb = 2; }
      if (vr1 > best) { best = vr1; kb = 3; }
      const float g0 = (ka == 1 ? gA : 0.f) + (kb == 0 ? gB : 0.f);
      const float g1 = (ka == 3 ? gA : 0.f) + (kb == 2 ? gB : 0.f);
      o0 = g0 * act_grad(pm0, act);
      o1 = g1 * act_grad(pm1, act);
    };
    float4 o0, o1;
    one(xv[0][0].x, xv[0][1].x, xv[0][2].x, xv[1][0].x, xv[1][1].x, xv[1][2].x, sc.x, sh.x, ga.x, gb.x, o0.x, o1.x);
    one(xv[0][0].y, xv[0][1].y, xv[0][2].y, xv[1][0].y, xv[1][1].y, xv[1][2].y, sc.y, sh.y, ga.y, gb.y, o0.y, o1.y);
    one(xv[0][0].z, xv[0][1].z, xv[0][2].z, xv[1][0].z, xv[1][1].z, xv[1][2].z, sc.z, sh.z, ga.z, gb.z, o0.z, o1.z);
    one(xv[0][0].w, xv[0][1].w, xv[0][2].w, xv[1][0].w, xv[1][1].w, xv[1][2].w, sc.w, sh.w, ga.w, gb.w, o0.w, o1.w);
    *reinterpret_cast<float4*>(dz + r0) = o0;
    *reinterpret_cast<float4*>(dz + r1) = o1;
  }
}

extern "C" int tpgsr_pool2d_bwd(const float* x, const float* dout, int N, int H, int W, int C, const float* scale, const float* shift,
                                int act, int KH, int KW, int SH, int SW, int PH, int PW, float* dz, void* stream) {
  TPGSR_CHECK_ARG(x && dout && dz && KH > 0 && KW > 0 && SH > 0 && SW > 0, "tpgsr_pool2d_bwd: bad arguments");
  int OH = (H + 2 * PH - KH) / SH + 1, OW = (W + 2 * PW - KW) / SW + 1;
  const bool vec = (C & 3) == 0 && ((((uintptr_t)x | (uintptr_t)dout | (uintptr_t)dz) & 15) == 0);
  long long total = (long long)N * H * W * (vec ? C / 4 : C);
  TPGSR_CHECK_ARG((long long)N * H * W * C < (1ll << 31), "tpgsr_pool2d_bwd: map too large for the kernels' 32-bit indices");
  int grid = (int)min((long long)16384, (total + 255) / 256);
  static const bool fast2x2 = [] { const char* e = getenv("TPGSR_POOL2X2_FAST"); return !(e && e[0] == '0'); }();
  if (fast2x2 && vec && KH == 2 && KW == 2 && SH == 2 && SW == 2 && PH == 0 && PW == 0 && !(H & 1) && !(W & 1) &&
      (!scale || !(((uintptr_t)scale | (uintptr_t)shift) & 15))) {
    const long long tw = (long long)N * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(pool2x2_bwd_kernel, dim3((int)min((long long)16384, (tw + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, dout, N, H, W,
                       C, scale, shift, act, dz);
    TPGSR_LAUNCH_CHECK("tpgsr_pool2d_bwd");
  }
  if (fast2x2 && vec && KH == 2 && KW == 2 && SH == 2 && SW == 1 && PH == 0 && PW == 1 && !(H & 1) &&
      (!scale || !(((uintptr_t)scale | (uintptr_t)shift) & 15))) {
    const long long tw = (long long)N * (H / 2) * W * (C / 4);
    hipLaunchKernelGGL(pool2x2s21_bwd_kernel, dim3((int)min((long long)16384, (tw + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, dout, N, H,
                       W, C, scale, shift, act, dz);
    TPGSR_LAUNCH_CHECK("tpgsr_pool2d_bwd");
  }
  if (vec)
    hipLaunchKernelGGL(pool2d_bwd_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, dout, N, H, W, C, scale, shift, act, KH, KW,
                       SH, SW, PH, PW, OH, OW, dz);
  else
    hipLaunchKernelGGL(pool2d_bwd_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, dout, N, H, W, C, scale, shift, act, KH, KW,
                       SH, SW, PH, PW, OH, OW, dz);
  TPGSR_LAUNCH_CHECK("tpgsr_pool2d_bwd");
}

// ------------------------------------------------------------------------------------------------------
// Recurrent projection of ONE BiLSTM time step, both directions in one launch, split over K:
//   slab[sp][d][n][c] = sum_{k in slice sp} A_d[n][k] * B_d[k][c]      (n < Nrows <= 64, c < Nc, Nc % 64 == 0)
// A_d row n starts at a_d + n * a_stride (K contiguous floats), B_d is row-major [K][Nc].  The step's GEMM is tiny
// (48 x 256 x 1024) and latency-bound: run as one 64x64 MFMA tile per column block it is 16 workgroups walking 8 K
// chunks in sequence (27 us); split over K it is 128 workgroups of one chunk each, and the gate kernel adds the slabs in
// a fixed order.  Grid: (Nc / 64, S, 2).
// ------------------------------------------------------------------------------------------------------
#define RG_KC 32
__global__ __launch_bounds__(256) void lstm_rec_gemm_kernel(const float* __restrict__ a0, const float* __restrict__ a1,
                                                            long long a_stride, const float* __restrict__ b0,
                                                            const float* __restrict__ b1, int Nrows, int K, int Nc, int S,
                                                            float* __restrict__ out) {
  __shared__ float As[RG_KC][65];
  __shared__ float Bs[RG_KC][64];
  const int d = blockIdx.z, sp = blockIdx.y, n0 = blockIdx.x * 64;
  const float* A = d == 0 ? a0 : a1;
  const float* B = d == 0 ? b0 : b1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int kper = (K / RG_KC + S - 1) / S * RG_KC;          // K slice of this split, whole chunks
  const int kbeg = sp * kper, kend = min(K, kbeg + kper);
  const int aq = tid & 7, am0 = tid >> 3, bk0 = tid >> 4, bc = (tid & 15) * 4;
  floatx16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int arow = lane >> 5, acol = wm * 32 + (lane & 31), bcol = wn * 32 + (lane & 31);
  for (int k0 = kbeg; k0 < kend; k0 += RG_KC) {
    float4 ra0 = make_float4(0.f, 0.f, 0.f, 0.f), ra1 = ra0;
    const int k = k0 + aq * 4;
    if (am0 < Nrows && k < kend) ra0 = *reinterpret_cast<const float4*>(A + (size_t)am0 * a_stride + k);
    if (am0 + 32 < Nrows && k < kend) ra1 = *reinterpret_cast<const float4*>(A + (size_t)(am0 + 32) * a_stride + k);
    float4 rb0 = make_float4(0.f, 0.f, 0.f, 0.f), rb1 = rb0;
    if (k0 + bk0 < kend) rb0 = *reinterpret_cast<const float4*>(B + (size_t)(k0 + bk0) * Nc + n0 + bc);
    if (k0 + bk0 + 16 < kend) rb1 = *reinterpret_cast<const float4*>(B + (size_t)(k0 + bk0 + 16) * Nc + n0 + bc);
    __syncthreads();   // previous chunk's fragment reads are done
    As[aq * 4 + 0][am0] = ra0.x; As[aq * 4 + 1][am0] = ra0.y; As[aq * 4 + 2][am0] = ra0.z; As[aq * 4 + 3][am0] = ra0.w;
    As[aq * 4 + 0][am0 + 32] = ra1.x; As[aq * 4 + 1][am0 + 32] = ra1.y; As[aq * 4 + 2][am0 + 32] = ra1.z; As[aq * 4 + 3][am0 + 32] = ra1.w;
    *reinterpret_cast<float4*>(&Bs[bk0][bc]) = rb0;
    *reinterpret_cast<float4*>(&Bs[bk0 + 16][bc]) = rb1;
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < RG_KC / 2; ++kk)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[2 * kk + arow][acol], Bs[2 * kk + arow][bcol], acc, 0, 0, 0);
  }
  float* dst = out + ((size_t)(sp * 2 + d) * Nrows) * Nc;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (row < Nrows) dst[(size_t)row * Nc + n0 + bcol] = acc[r];
  }
}

extern "C" int tpgsr_lstm_rec_gemm(const float* a0, const float* a1, long long a_stride, const float* b0, const float* b1,
                                   int Nrows, int K, int Nc, int S, float* out, void* stream) {
  TPGSR_CHECK_ARG(a0 && a1 && b0 && b1 && out, "tpgsr_lstm_rec_gemm: null pointer");
  TPGSR_CHECK_ARG(Nrows > 0 && Nrows <= 64 && K > 0 && (K % RG_KC) == 0 && Nc > 0 && (Nc % 64) == 0 && S > 0 && S <= K / RG_KC &&
                      (a_stride & 3) == 0 && (((uintptr_t)a0 | (uintptr_t)a1 | (uintptr_t)b0 | (uintptr_t)b1) & 15) == 0,
                  "tpgsr_lstm_rec_gemm: needs Nrows <= 64, K %% 32 == 0, Nc %% 64 == 0, 1 <= S <= K/32, 16-byte aligned operands");
  hipLaunchKernelGGL(lstm_rec_gemm_kernel, dim3(Nc / 64, S, 2), dim3(256), 0, (hipStream_t)stream, a0, a1, a_stride, b0, b1, Nrows, K,
                     Nc, S, out);
  TPGSR_LAUNCH_CHECK("tpgsr_lstm_rec_gemm");
}

// ------------------------------------------------------------------------------------------------------
// BiLSTM time step (gate order i, f, g, o).  Layouts (batch-major, T = sequence length, Hh = hidden):
//   G   [N][T][2][4*Hh]  input projections (+b_ih+b_hh) on entry, ACTIVATED gates on exit (saved for backward)
//   gh  [S][2][N][4*Hh]  this step's recurrent projections W_hh h_{prev} as S K-split slabs (tpgsr_lstm_rec_gemm), summed
//                        here in slab order (ignored at step 0)
//   Cst [N][T][2][Hh]    cell states,   out [N][T][2*Hh]  hidden states (direction d in columns d*Hh..)
// step s processes t = s for the forward direction and t = T-1-s for the reverse direction.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lstm_step_fwd_kernel(float* __restrict__ G, const float* __restrict__ gh,
                                                            const float* __restrict__ bhh, float* __restrict__ Cst,
                                                            float* __restrict__ out, int N, int T, int Hh, int s, int nsplit) {
  int total = N * 2 * Hh;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int j = i % Hh;
  int d = (i / Hh) & 1;
  int n = i / (2 * Hh);
  int t = d == 0 ? s : T - 1 - s;
  int tp = d == 0 ? t - 1 : t + 1;
  float* g = G + (((size_t)n * T + t) * 2 + d) * 4 * Hh;
  float pi = g[j], pf = g[Hh + j], pg = g[2 * Hh + j], po = g[3 * Hh + j];
  if (bhh) {   // b_hh [2][4Hh] applies at every step (also at step 0 where h_prev = 0)
    const float* b = bhh + (size_t)d * 4 * Hh;
    pi += b[j];
    pf += b[Hh + j];
    pg += b[2 * Hh + j];
    po += b[3 * Hh + j];
  }
  float cprev = 0.f;
  if (s > 0) {
    float ri = 0.f, rf = 0.f, rg = 0.f, ro = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) {
      const float* r = gh + ((size_t)(sp * 2 + d) * N + n) * 4 * Hh;
      ri += r[j];
      rf += r[Hh + j];
      rg += r[2 * Hh + j];
      ro += r[3 * Hh + j];
    }
    pi += ri;
    pf += rf;
    pg += rg;
    po += ro;
    cprev = Cst[(((size_t)n * T + tp) * 2 + d) * Hh + j];
  }
  float ig = gru_sigmoid1(pi), fg = gru_sigmoid1(pf), gg = gru_tanh(pg), og = gru_sigmoid1(po);
  float c = fg * cprev + ig * gg;
  float h = og * gru_tanh(c);
  g[j] = ig;
  g[Hh + j] = fg;
  g[2 * Hh + j] = gg;
  g[3 * Hh + j] = og;
  Cst[(((size_t)n * T + t) * 2 + d) * Hh + j] = c;
  out[((size_t)n * T + t) * 2 * Hh + d * Hh + j] = h;
}

extern "C" int tpgsr_lstm_step_fwd(float* G, const float* gh, int nsplit, const float* bhh, float* Cst, float* out, int N, int T, int Hh,
                                   int step, void* stream) {
  TPGSR_CHECK_ARG(G && Cst && out && ((gh && nsplit > 0) || step == 0) && N > 0 && T > 0 && Hh > 0 && step >= 0 && step < T,
                  "tpgsr_lstm_step_fwd: bad arguments");
  hipLaunchKernelGGL(lstm_step_fwd_kernel, dim3(cdiv((long long)N * 2 * Hh, 256)), dim3(256), 0, (hipStream_t)stream, G, gh, bhh, Cst, out, N, T, Hh,
                     step, nsplit);
  TPGSR_LAUNCH_CHECK("tpgsr_lstm_step_fwd");
}

// ------------------------------------------------------------------------------------------------------
// Helpers shared with lstm_seq.hip (BiLSTM forward / backward as ONE persistent launch each).  Exchange layout of the forward pass:
// 2 x 32 workgroups, workgroup (d, u) owns hidden units
// 8u .. 8u+7 of direction d -- its 32 gate columns of W_hh^T stay in LDS, pre-split into three bf16 terms in MFMA fragment order,
// for all T steps.  Per step every workgroup computes  gh[n][32 cols] = h_prev[n][256] x W[256][32]  on the bf16 matrix cores with
// split operands (fp32-equivalent, as the conv kernels), adds the input projections, runs the gate math for its 8 units of
// all sequences and publishes h as bf16 terms in A-FRAGMENT order (8 consecutive k of a fragment row = exactly one workgroup's
// units, so a workgroup's contribution is 512 contiguous bytes per row block and term) to a parity-double-buffered exchange
// buffer; a direction-local grid barrier (agent-scope release -> relaxed counter -> one acquire, cdna_hip_programming.md G16)
// ends the step.  Replaces 2 launches per time step (tpgsr_lstm_rec_gemm + tpgsr_lstm_step_fwd: ~10.5 us) by ~4 us.
//   hx    [2 parity][2 dir][3 terms][2 row blocks][16 k-blocks][64 lanes][8] bf16, zeroed once by the caller (rows >= N stay 0)
//   sync  [4] u32: arrival counters of the two directions, a timeout flag, spare; zeroed by the launcher on the stream
// ------------------------------------------------------------------------------------------------------
#define LS_NW 32
typedef __bf16 ls_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int ls_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void ls_split3(float x, unsigned short (&h)[3]) {
  __bf16 a = (__bf16)x;
  float r = x - (float)a;
  __bf16 b = (__bf16)r;
  __bf16 c = (__bf16)(r - (float)b);
  h[0] = __builtin_bit_cast(unsigned short, a);
  h[1] = __builtin_bit_cast(unsigned short, b);
  h[2] = __builtin_bit_cast(unsigned short, c);
}

// (the persistent kernels themselves live in lstm_seq.hip)
/* bytes of the exchange buffer `hx` of tpgsr_lstm_seq_fwd */
extern "C" long long tpgsr_lstm_seq_hx_bytes(void) { return 2ll * 2 * 3 * 2 * 16 * 512 * 2; }

// ------------------------------------------------------------------------------------------------------
// BiLSTM time step as ONE launch (Hh = 256, N <= 64): recurrent projection + gate math fused, no split-K slabs, no
// inter-workgroup exchange inside the launch.  2 x 16 workgroups; workgroup (d, ub) owns hidden units 16 ub .. 16 ub + 15 of
// direction d = 64 gate columns (column c: gate c >> 4, unit 16 ub + (c & 15)); its four waves compute the four 32 x 32 blocks of
//   gh[n][64] = h_prev[n][256] x W[256][64]
// over the FULL K on the bf16 matrix cores with split operands (16 k-blocks x 6 MFMAs per wave = 1.3 us), both operands straight
// from global memory in fragment order: W pre-split once per pass by lstm_wfrag_kernel, h_prev published by the previous step's
// launch as bf16 terms in A-fragment order (a k-block of 16 = exactly one workgroup's units: 1 KB per row block and term).
// Then the gate math for the workgroup's 16 units of all sequences.  Replaces tpgsr_lstm_rec_gemm + tpgsr_lstm_step_fwd
// (128 + 96 workgroups, two launches, ~10.5 us + a launch gap) per step.
//   wfr  [2 dir][16 ub][2 col blocks][3 terms][16 k-blocks][64 lanes][8] bf16
//   hx   [2 parity][2 dir][3 terms][2 row blocks][16 k-blocks][64 lanes][8] bf16, zeroed once by the caller
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lstm_wfrag_kernel(const float* __restrict__ whhT, unsigned short* __restrict__ wfr) {
  constexpr int Hh = 256, G4 = 1024;
  const int d = blockIdx.x >> 4, ub = blockIdx.x & 15;
  const float* W = whhT + (size_t)d * Hh * G4;
  unsigned short* dst0 = wfr + (size_t)blockIdx.x * 2 * 3 * 16 * 512;
  for (int idx = threadIdx.x; idx < 2 * 16 * 64; idx += 256) {
    const int cb = idx >> 10, kb = (idx >> 6) & 15, l = idx & 63;
    const int c = cb * 32 + (l & 31);
    const int col = (c >> 4) * Hh + ub * 16 + (c & 15);
    unsigned short hv[8][3];
#pragma unroll
    for (int j = 0; j < 8; ++j) ls_split3(W[(size_t)(kb * 16 + (l >> 5) * 8 + j) * G4 + col], hv[j]);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      unsigned short v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = hv[j][t];
      *reinterpret_cast<ls_u32x4*>(dst0 + ((size_t)((cb * 3 + t) * 16 + kb) * 64 + l) * 8) = *reinterpret_cast<ls_u32x4*>(v);
    }
  }
}

__global__ __launch_bounds__(256) void lstm_stepx_fwd_kernel(float* __restrict__ G, const unsigned short* __restrict__ wfr,
                                                             const float* __restrict__ bhh, float* __restrict__ Cst,
                                                             float* __restrict__ out, unsigned short* __restrict__ hx, int N, int T,
                                                             int s) {
  constexpr int Hh = 256, G4 = 1024;
  __shared__ __attribute__((aligned(16))) float red[64 * 64];      // gh[row][col]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = blockIdx.x >> 4, ub = blockIdx.x & 15;
  const int t = d == 0 ? s : T - 1 - s;
  const int tp = d == 0 ? t - 1 : t + 1;
  const size_t par_elems = (size_t)2 * 3 * 2 * 16 * 512;
  if (s > 0) {
    const int rb = wave & 1, cb = wave >> 1;
    const unsigned short* hp = hx + (size_t)((s - 1) & 1) * par_elems + (size_t)d * 3 * 2 * 16 * 512;
    const unsigned short* wp = wfr + ((size_t)blockIdx.x * 2 + cb) * 3 * 16 * 512;
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    ls_u32x4 av[2][4][3], bv[2][4][3];      // two register sets of four k-blocks each: loads of the next four under these MFMAs
    auto fetch = [&](auto set_tag, const int kb0) __attribute__((always_inline)) {
      constexpr int S = decltype(set_tag)::value;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
          av[S][i][tt] = *reinterpret_cast<const ls_u32x4*>(hp + (((size_t)(tt * 2 + rb) * 16 + kb0 + i) * 64 + lane) * 8);
          bv[S][i][tt] = *reinterpret_cast<const ls_u32x4*>(wp + (((size_t)tt * 16 + kb0 + i) * 64 + lane) * 8);
        }
    };
    auto multiply = [&](auto set_tag) __attribute__((always_inline)) {
      constexpr int S = decltype(set_tag)::value;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ls_bf16x8 a[3], b[3];
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
          a[tt] = __builtin_bit_cast(ls_bf16x8, av[S][i][tt]);
          b[tt] = __builtin_bit_cast(ls_bf16x8, bv[S][i][tt]);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
      }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    fetch(S0{}, 0);
    fetch(S1{}, 4);
    __builtin_amdgcn_sched_barrier(0);
    multiply(S0{});
    __builtin_amdgcn_sched_barrier(0);
    fetch(S0{}, 8);
    __builtin_amdgcn_sched_barrier(0);
    multiply(S1{});
    __builtin_amdgcn_sched_barrier(0);
    fetch(S1{}, 12);
    __builtin_amdgcn_sched_barrier(0);
    multiply(S0{});
    __builtin_amdgcn_sched_barrier(0);
    multiply(S1{});
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      red[row * 64 + cb * 32 + (lane & 31)] = acc[r];
    }
    __syncthreads();
  }
  unsigned short* hw = hx + (size_t)(s & 1) * par_elems + (size_t)d * 3 * 2 * 16 * 512;
  for (int item = tid; item < N * 16; item += 256) {
    const int n = item >> 4, ul = item & 15, unit = ub * 16 + ul;
    float* g = G + (((size_t)n * T + t) * 2 + d) * G4;
    const float* b = bhh ? bhh + (size_t)d * G4 : nullptr;
    float pre[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      pre[q] = g[q * Hh + unit] + (b ? b[q * Hh + unit] : 0.f);
      if (s > 0) pre[q] += red[n * 64 + q * 16 + ul];
    }
    const float cprev = s > 0 ? Cst[(((size_t)n * T + tp) * 2 + d) * Hh + unit] : 0.f;
    const float ig = gru_sigmoid1(pre[0]), fg = gru_sigmoid1(pre[1]), gg = gru_tanh(pre[2]), og = gru_sigmoid1(pre[3]);
    const float c = fg * cprev + ig * gg;
    const float h = og * gru_tanh(c);
    g[unit] = ig;
    g[Hh + unit] = fg;
    g[2 * Hh + unit] = gg;
    g[3 * Hh + unit] = og;
    Cst[(((size_t)n * T + t) * 2 + d) * Hh + unit] = c;
    out[((size_t)n * T + t) * 2 * Hh + d * Hh + unit] = h;
    unsigned short hv[3];
    ls_split3(h, hv);
    const int ln = ((ul >> 3) & 1) * 32 + (n & 31);
#pragma unroll
    for (int tt = 0; tt < 3; ++tt) hw[(((size_t)(tt * 2 + (n >> 5)) * 16 + ub) * 64 + ln) * 8 + (ul & 7)] = hv[tt];
  }
}

extern "C" long long tpgsr_lstm_wfrag_bytes(void) { return 2ll * 16 * 2 * 3 * 16 * 512 * 2; }
extern "C" int tpgsr_lstm_wfrag(const float* whhT, void* wfr, int Hh, void* stream) {
  TPGSR_CHECK_ARG(whhT && wfr && Hh == 256, "tpgsr_lstm_wfrag: needs Hh == 256 and non-null buffers");
  hipLaunchKernelGGL(lstm_wfrag_kernel, dim3(32), dim3(256), 0, (hipStream_t)stream, whhT, (unsigned short*)wfr);
  TPGSR_LAUNCH_CHECK("tpgsr_lstm_wfrag");
}
extern "C" int tpgsr_lstm_stepx_fwd(float* G, const void* wfr, const float* bhh, float* Cst, float* out, void* hx, int N, int T, int Hh,
                                    int step, void* stream) {
  TPGSR_CHECK_ARG(G && wfr && Cst && out && hx && N > 0 && N <= 64 && T > 0 && Hh == 256 && step >= 0 && step < T,
                  "tpgsr_lstm_stepx_fwd: needs Hh == 256, 1 <= N <= 64, 0 <= step < T and non-null buffers");
  hipLaunchKernelGGL(lstm_stepx_fwd_kernel, dim3(32), dim3(256), 0, (hipStream_t)stream, G, (const unsigned short*)wfr, bhh, Cst, out,
                     (unsigned short*)hx, N, T, step);
  TPGSR_LAUNCH_CHECK("tpgsr_lstm_stepx_fwd");
}

// backward step s' (reverse of the forward order): t = T-1-s' (forward dir), t = s' (reverse dir).
//   dout [N][T][2*Hh] gradient w.r.t. the hidden states,  dhc [S][2][N][Hh] recurrent gradient W_hh^T dG[t_next] as K-split slabs
//   (ignored at s' = 0)
//   dcc [N][2][Hh] running cell-state gradient (in/out),  G: activated gates in, dG (pre-activation gate gradients) out
__global__ __launch_bounds__(256) void lstm_step_bwd_kernel(float* __restrict__ G, const float* __restrict__ Cst,
                                                            const float* __restrict__ dout, const float* __restrict__ dhc,
                                                            float* __restrict__ dcc, int N, int T, int Hh, int s, int nsplit) {
  int total = N * 2 * Hh;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int j = i % Hh;
  int d = (i / Hh) & 1;
  int n = i / (2 * Hh);
  int t = d == 0 ? T - 1 - s : s;
  int tp = d == 0 ? t - 1 : t + 1;   // time index of the previous state in this direction's forward order
  bool has_prev = d == 0 ? t > 0 : t < T - 1;
  float* g = G + (((size_t)n * T + t) * 2 + d) * 4 * Hh;
  float ig = g[j], fg = g[Hh + j], gg = g[2 * Hh + j], og = g[3 * Hh + j];
  float c = Cst[(((size_t)n * T + t) * 2 + d) * Hh + j];
  float cprev = has_prev ? Cst[(((size_t)n * T + tp) * 2 + d) * Hh + j] : 0.f;
  float dh = dout[((size_t)n * T + t) * 2 * Hh + d * Hh + j];
  float dc = 0.f;
  if (s > 0) {
    float rh = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) rh += dhc[((size_t)(sp * 2 + d) * N + n) * Hh + j];
    dh += rh;
    dc = dcc[((size_t)n * 2 + d) * Hh + j];
  }
  float tc = gru_tanh(c);
  float dog = dh * tc * og * (1.f - og);
  dc += dh * og * (1.f - tc * tc);
  float dig = dc * gg * ig * (1.f - ig);
  float dfg = dc * cprev * fg * (1.f - fg);
  float dgg = dc * ig * (1.f - gg * gg);
  dcc[((size_t)n * 2 + d) * Hh + j] = dc * fg;
  g[j] = dig;
  g[Hh + j] = dfg;
  g[2 * Hh + j] = dgg;
  g[3 * Hh + j] = dog;
}

extern "C" int tpgsr_lstm_step_bwd(float* G, const float* Cst, const float* dout, const float* dhc, int nsplit, float* dcc, int N, int T,
                                   int Hh, int step, void* stream) {
  TPGSR_CHECK_ARG(G && Cst && dout && dcc && ((dhc && nsplit > 0) || step == 0) && N > 0 && T > 0 && Hh > 0 && step >= 0 && step < T,
                  "tpgsr_lstm_step_bwd: bad arguments");
  hipLaunchKernelGGL(lstm_step_bwd_kernel, dim3(cdiv((long long)N * 2 * Hh, 256)), dim3(256), 0, (hipStream_t)stream, G, Cst, dout, dhc, dcc, N,
                     T, Hh, step, nsplit);
  TPGSR_LAUNCH_CHECK("tpgsr_lstm_step_bwd");
}

// ------------------------------------------------------------------------------------------------------
// softmax over classes + SemanticLoss partials + (N, C, 1, T) prior with prior dropout
//   logits [N][T][C] -> p [N][T][C];  prior[n][c][0][t] = (n < drop_n ? 0 : p[n][t][c])
//   if q: partial[blk][0] = sum |q - p|, partial[blk][1] = sum q' (log q' - log p'),  p' = p + 1e-20, q' = q + 1e-20
// one wavefront per (n, t) row
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_prior_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ q, int N, int T,
                                                                int C, int drop_n, float* __restrict__ p, float* __restrict__ prior,
                                                                float* __restrict__ partial) {
  __shared__ float red[2][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rows = N * T;
  float l1 = 0.f, kl = 0.f;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const float* x = logits + (size_t)row * C;
    float v = lane < C ? x[lane] : -INFINITY;
    float m = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float e = lane < C ? expf(v - m) : 0.f;
    float ssum = wave_sum(e);
    float pv = e / ssum;
    if (lane < C) {
      p[(size_t)row * C + lane] = pv;
      int n = row / T, t = row - n * T;
      if (prior) prior[((size_t)n * C + lane) * T + t] = n < drop_n ? 0.f : pv;
      if (q) {
        float qv = q[(size_t)row * C + lane];
        l1 += fabsf(qv - pv);
        float qp = qv + 1e-20f;
        kl += qp * (logf(qp) - logf(pv + 1e-20f));
      }
    }
  }
  if (partial) {
    l1 = wave_sum(l1);
    kl = wave_sum(kl);
    if (lane == 0) {
      red[0][wave] = l1;
      red[1][wave] = kl;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      partial[blockIdx.x * 2] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
      partial[blockIdx.x * 2 + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
  }
}

extern "C" int tpgsr_softmax_prior_fwd(const float* logits, const float* q, int N, int T, int C, int drop_n, float* p, float* prior_nchw,
                                       float* partial, int nblk, void* stream) {
  TPGSR_CHECK_ARG(logits && p && N > 0 && T > 0 && C > 0 && C <= 64 && nblk > 0, "tpgsr_softmax_prior_fwd: bad arguments (C <= 64)");
  TPGSR_CHECK_ARG(!q || partial, "tpgsr_softmax_prior_fwd: semantic-loss partials need a partial buffer");
  hipLaunchKernelGGL(softmax_prior_fwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, logits, q, N, T, C, drop_n, p, prior_nchw,
                     q ? partial : nullptr);
  TPGSR_LAUNCH_CHECK("tpgsr_softmax_prior_fwd");
}

// loss = w * (sum|q-p| + sum kl) / (N*T*C)
__global__ void semantic_loss_finalize_kernel(const float* __restrict__ partial, int nblk, long long count, float w, float* loss) {
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 64) {
    a += (double)partial[i * 2];
    b += (double)partial[i * 2 + 1];
  }
  a = wave_sum_d(a);
  b = wave_sum_d(b);
  if (threadIdx.x == 0) loss[0] = (float)((double)w * (a + b) / (double)count);
}

extern "C" int tpgsr_semantic_loss_finalize(const float* partial, int nblk, long long count, float w, float* loss, void* stream) {
  TPGSR_CHECK_ARG(partial && loss && nblk > 0 && count > 0, "tpgsr_semantic_loss_finalize: bad arguments");
  hipLaunchKernelGGL(semantic_loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, nblk, count, w, loss);
  TPGSR_LAUNCH_CHECK("tpgsr_semantic_loss_finalize");
}

// dlogits = softmax backward of dp, dp = [n >= drop_n] * dprior[n][c][0][t] + wsem/(count) * ( -sign(q-p) - q'/(p') )
__global__ __launch_bounds__(256) void softmax_prior_bwd_kernel(const float* __restrict__ p, const float* __restrict__ q,
                                                                const float* __restrict__ dprior, const float* __restrict__ dp_in,
                                                                int N, int T, int C, int drop_n, float wsem_over_count,
                                                                float* __restrict__ dlogits) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rows = N * T;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    int n = row / T, t = row - n * T;
    float pv = 0.f, dp = 0.f;
    if (lane < C) {
      pv = p[(size_t)row * C + lane];
      if (dprior && n >= drop_n) dp += dprior[((size_t)n * C + lane) * T + t];
      if (dp_in) dp += dp_in[(size_t)row * C + lane];
      if (q) {
        float qv = q[(size_t)row * C + lane];
        float diff = qv - pv;
        float sg = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
        dp += wsem_over_count * (-sg - (qv + 1e-20f) / (pv + 1e-20f));
      }
    }
    float dot = wave_sum(dp * pv);
    if (lane < C) dlogits[(size_t)row * C + lane] = pv * (dp - dot);
  }
}

extern "C" int tpgsr_softmax_prior_bwd(const float* p, const float* q, const float* dprior_nchw, const float* dp_in, int N, int T, int C,
                                       int drop_n, float wsem, float* dlogits, int nblk, void* stream) {
  TPGSR_CHECK_ARG(p && dlogits && N > 0 && T > 0 && C > 0 && C <= 64 && nblk > 0, "tpgsr_softmax_prior_bwd: bad arguments");
  float wc = q ? wsem / (float)((long long)N * T * C) : 0.f;
  hipLaunchKernelGGL(softmax_prior_bwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, p, q, dprior_nchw, dp_in, N, T, C, drop_n, wc,
                     dlogits);
  TPGSR_LAUNCH_CHECK("tpgsr_softmax_prior_bwd");
}


// ------------------------------------------------------------------------------------------------------
// SemanticLoss on probability tensors (loss/semantic_loss.py:21-39, the nn.Module API): no softmax involved, rows need
// not sum to 1.  partial[blk] = (sum |q - p|, sum q' (log q' - log p')), p' = p + 1e-20, q' = q + 1e-20;
// backward: dp = dloss * (-sign(q - p) - q'/p') / count
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void semantic_loss_fwd_kernel(const float* __restrict__ p, const float* __restrict__ q, long long n,
                                                                float* __restrict__ partial) {
  __shared__ float red[2][4];
  float l1 = 0.f, kl = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float pv = p[i], qv = q[i];
    l1 += fabsf(qv - pv);
    float qp = qv + 1e-20f;
    kl += qp * (logf(qp) - logf(pv + 1e-20f));
  }
  l1 = wave_sum(l1);
  kl = wave_sum(kl);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = l1;
    red[1][threadIdx.x >> 6] = kl;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[blockIdx.x * 2] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    partial[blockIdx.x * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

extern "C" int tpgsr_semantic_loss_fwd(const float* p, const float* q, long long n, float* partial, int nblk, void* stream) {
  TPGSR_CHECK_ARG(p && q && partial && n > 0 && nblk > 0, "tpgsr_semantic_loss_fwd: bad arguments");
  hipLaunchKernelGGL(semantic_loss_fwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, p, q, n, partial);
  TPGSR_LAUNCH_CHECK("tpgsr_semantic_loss_fwd");
}

__global__ __launch_bounds__(256) void semantic_loss_bwd_kernel(const float* __restrict__ p, const float* __restrict__ q,
                                                                const float* __restrict__ dloss, long long n, float* __restrict__ dp) {
  const float w = dloss[0] / (float)n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float pv = p[i], qv = q[i];
    float diff = qv - pv;
    float sg = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
    dp[i] = w * (-sg - (qv + 1e-20f) / (pv + 1e-20f));
  }
}

extern "C" int tpgsr_semantic_loss_bwd(const float* p, const float* q, const float* dloss, long long n, float* dp, void* stream) {
  TPGSR_CHECK_ARG(p && q && dloss && dp && n > 0, "tpgsr_semantic_loss_bwd: bad arguments");
  int grid = (int)min((long long)1024, (n + 255) / 256);
  hipLaunchKernelGGL(semantic_loss_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, q, dloss, n, dp);
  TPGSR_LAUNCH_CHECK("tpgsr_semantic_loss_bwd");
}

// ------------------------------------------------------------------------------------------------------
// CTC loss of the text-prior generator's logits (`--use_label`, interfaces/super_resolution.py:40, :347-366:
// ctc_loss = torch.nn.CTCLoss(blank=0, reduction='none') on label_vecs_logits.log_softmax(2); the per-sample values are weighted by
// weighted_tics and averaged).  One wavefront per sample: lane s owns state s of the extended label l' (blank, l_1, blank, ... l_L, blank:
// S = 2 L + 1 <= 63), the forward variables alpha_t(s) are kept in LDS for all T <= 32 steps, the backward variables beta_t(s) in a register;
// everything in log space as ATen's LossCTC.cpp does (alpha and beta both include log p_t(l'_s)):
//   nll      = -logsumexp(alpha_{T-1}(S-1), alpha_{T-1}(S-2))
//   d nll / d logit[t][k] = p_t(k) - exp( logsumexp_{s: l'_s = k}(alpha_t(s) + beta_t(s)) + nll - log p_t(k) )
// logits: element (n, t, c) at n * sn + t * st + c (the fused step keeps [N][T][C], the module API's tensor is [T][N][C]).
// dlogits (same addressing) (+)= scale * weight[n] * d nll_n / d logits.  Infeasible targets (T too short) give nll = +inf as in ATen.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float ctc_lse2(float a, float b) {
  const float m = fmaxf(a, b);
  return m == -INFINITY ? -INFINITY : m + logf(expf(a - m) + expf(b - m));
}
__device__ __forceinline__ float ctc_lse3(float a, float b, float c) {
  const float m = fmaxf(fmaxf(a, b), c);
  return m == -INFINITY ? -INFINITY : m + logf(expf(a - m) + expf(b - m) + expf(c - m));
}

__global__ __launch_bounds__(64) void ctc_loss_kernel(const float* __restrict__ logits, int sn, int st, const int* __restrict__ targets,
                                                      const int* __restrict__ tgt_off, const int* __restrict__ tgt_len,
                                                      const float* __restrict__ weight, int T, int C, int blank, float scale,
                                                      float* __restrict__ nll_out, float* __restrict__ dlogits, int accumulate) {
  __shared__ float lp[32][64];      // log-softmax [t][c]
  __shared__ float al[32][64];      // alpha [t][s]
  __shared__ float ab[64];          // alpha_t(s) + beta_t(s) of the current step
  __shared__ int lab[64];           // l'_s
  const int n = blockIdx.x, lane = threadIdx.x;
  const int L = tgt_len[n], S = 2 * L + 1, off = tgt_off[n];
  const float* x = logits + (size_t)n * sn;
  for (int t = 0; t < T; ++t) {
    const float v = lane < C ? x[(size_t)t * st + lane] : -INFINITY;
    float m = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    const float e = lane < C ? expf(v - m) : 0.f;
    const float lse = m + logf(wave_sum(e));
    lp[t][lane] = v - lse;
  }
  const int ls = lane < S ? ((lane & 1) ? targets[off + (lane >> 1)] : blank) : blank;
  lab[lane] = lane < S ? ls : -1;
  __builtin_amdgcn_wave_barrier();
  const int l_m2 = __shfl_up(ls, 2), l_p2 = __shfl_down(ls, 2);
  // (ATen's rule, LossCTC.cpp: the skip s - 2 -> s exists iff l'_{s-2} != l'_s -- by VALUE, with no test for the blank: between two
  //  blank positions the labels are equal anyway, and a LABEL that happens to carry the blank's index -- the collate's "-" -> 0 -- is
  //  skipped over like any other label)
  const bool skip_in = lane >= 2 && lane < S && ls != l_m2;             // alpha: from s - 2
  const bool skip_out = lane + 2 < S && l_p2 != ls;                      // beta: to s + 2
  // ---- forward variables ----
  float a = -INFINITY;
  if (lane == 0) a = lp[0][blank];
  if (lane == 1 && S > 1) a = lp[0][ls];
  al[0][lane] = a;
  for (int t = 1; t < T; ++t) {
    float a1 = __shfl_up(a, 1), a2 = __shfl_up(a, 2);
    if (lane < 1) a1 = -INFINITY;
    if (!skip_in) a2 = -INFINITY;
    a = lane < S ? lp[t][ls] + ctc_lse3(a, a1, a2) : -INFINITY;
    al[t][lane] = a;
  }
  const float aS1 = __shfl(a, S - 1), aS2 = S > 1 ? __shfl(a, S - 2) : -INFINITY;
  const float nll = -ctc_lse2(aS1, aS2);
  if (lane == 0) nll_out[n] = nll;
  if (!dlogits) return;
  // ---- backward variables + gradient ----
  const float g = scale * (weight ? weight[n] : 1.f);
  float b = -INFINITY;
  if (lane == S - 1 || (S > 1 && lane == S - 2)) b = lp[T - 1][ls];
  for (int t = T - 1; t >= 0; --t) {
    ab[lane] = lane < S ? al[t][lane] + b : -INFINITY;
    __builtin_amdgcn_wave_barrier();
    if (lane < C) {
      float m = -INFINITY;
      for (int s2 = 0; s2 < S; ++s2)
        if (lab[s2] == lane) m = fmaxf(m, ab[s2]);
      float occ = 0.f;
      if (m != -INFINITY) {
        float sum = 0.f;
        for (int s2 = 0; s2 < S; ++s2)
          if (lab[s2] == lane) sum += expf(ab[s2] - m);
        occ = expf(m + logf(sum) + nll - lp[t][lane]);
      }
      const float gr = g * (expf(lp[t][lane]) - occ);
      float* d = dlogits + (size_t)n * sn + (size_t)t * st + lane;
      *d = accumulate ? *d + gr : gr;
    }
    __builtin_amdgcn_wave_barrier();
    if (t > 0) {
      float b1 = __shfl_down(b, 1), b2 = __shfl_down(b, 2);
      if (lane + 1 >= S) b1 = -INFINITY;
      if (!skip_out) b2 = -INFINITY;
      b = lane < S ? lp[t - 1][ls] + ctc_lse3(b, b1, b2) : -INFINITY;
    }
  }
}

/* nll[n] = CTC negative log-likelihood of sample n (blank index `blank`, targets concatenated: sample n's labels are
 * targets[tgt_off[n] .. + tgt_len[n])), T <= 32 time steps, C <= 64 classes, at most 31 labels per sample; dlogits (optional, same
 * addressing as logits: n * sn + t * st + c) (+)= scale * weight[n] * d nll[n] / d logits.  torch.nn.CTCLoss(blank, reduction='none') on
 * log_softmax(logits), interfaces/super_resolution.py:40, :355-366. */
extern "C" int tpgsr_ctc_loss(const float* logits, int sn, int st, const int* targets, const int* tgt_off, const int* tgt_len, const float* weight,
                              int N, int T, int C, int blank, float scale, float* nll, float* dlogits, int accumulate, int max_len, void* stream) {
  TPGSR_CHECK_ARG(logits && targets && tgt_off && tgt_len && nll && N > 0 && T > 0 && T <= 32 && C > 0 && C <= 64 && blank >= 0 && blank < C &&
                  max_len >= 0 && max_len <= 31, "tpgsr_ctc_loss: needs T <= 32, C <= 64, at most 31 labels per sample (got T %d, C %d, max_len %d)", T, C, max_len);
  hipLaunchKernelGGL(ctc_loss_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, logits, sn, st, targets, tgt_off, tgt_len, weight, T, C, blank,
                     scale, nll, dlogits, accumulate);
  TPGSR_LAUNCH_CHECK("tpgsr_ctc_loss");
}
