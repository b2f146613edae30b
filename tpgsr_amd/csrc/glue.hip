// Layout / resampling "glue" of the standalone operator API and the _TL baseline backbones (HBM-bound, one pass each):
//   channel-slice copies (torch.cat / slicing in NHWC), nearest and bilinear resize (F.interpolate), zero-dilation and its
//   adjoint sub-sampling (ConvTranspose2d / strided convs expressed through the stride-1 MFMA conv), mean over the height
//   axis (nn.AdaptiveAvgPool2d((None, 1)) of the OPT text-prior generator).
// Reference call sites: model/srresnet.py:88-235, model/srcnn.py:50-106, model/vdsr.py:21-233, model/rdn.py:126-214,
// model/crnn/model.py:25-95, model/crnn/modules/feature_extraction.py:196-246.  Every backward is a gather (deterministic).
#include "common.h"

// dst[m][dst_coff + c] (+)= src[m][src_coff + c], c < C
__global__ __launch_bounds__(256) void copy_strided_kernel(const float* __restrict__ src, int src_ld, int src_coff, float* dst,
                                                           int dst_ld, int dst_coff, long long M, int C, int accumulate) {
  long long total = M * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    long long m = i / C;
    float v = src[m * src_ld + src_coff + c];
    float* d = dst + m * dst_ld + dst_coff + c;
    *d = accumulate ? *d + v : v;
  }
}

extern "C" int tpgsr_copy_strided(const float* src, int src_ld, int src_coff, float* dst, int dst_ld, int dst_coff, long long M, int C,
                                  int accumulate, void* stream) {
  TPGSR_CHECK_ARG(src && dst && M > 0 && C > 0 && src_ld >= src_coff + C && dst_ld >= dst_coff + C, "tpgsr_copy_strided: bad arguments");
  int grid = (int)min((long long)8192, (M * C + 255) / 256);
  hipLaunchKernelGGL(copy_strided_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, src_ld, src_coff, dst, dst_ld, dst_coff, M, C,
                     accumulate);
  TPGSR_LAUNCH_CHECK("tpgsr_copy_strided");
}

// F.interpolate(x, scale_factor=s) (mode 'nearest'), NHWC: out[n][oh][ow][c] = in[n][oh/s][ow/s][c]
__global__ __launch_bounds__(256) void resize_nearest_fwd_kernel(const float* __restrict__ in, int N, int H, int W, int C, int s,
                                                                 float* __restrict__ out) {
  long long total = (long long)N * H * s * W * s * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    long long p = i / C;
    int ow = (int)(p % (W * s));
    p /= W * s;
    int oh = (int)(p % (H * s));
    int n = (int)(p / (H * s));
    out[i] = in[(((size_t)n * H + oh / s) * W + ow / s) * C + c];
  }
}
__global__ __launch_bounds__(256) void resize_nearest_bwd_kernel(const float* __restrict__ dout, int N, int H, int W, int C, int s,
                                                                 float* __restrict__ din) {
  long long total = (long long)N * H * W * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    long long p = i / C;
    int w = (int)(p % W);
    p /= W;
    int h = (int)(p % H);
    int n = (int)(p / H);
    float g = 0.f;
    for (int a = 0; a < s; ++a)
      for (int b = 0; b < s; ++b) g += dout[(((size_t)n * H * s + h * s + a) * (W * s) + w * s + b) * C + c];
    din[i] = g;
  }
}

extern "C" int tpgsr_resize_nearest_fwd(const float* in, int N, int H, int W, int C, int s, float* out, void* stream) {
  TPGSR_CHECK_ARG(in && out && N > 0 && H > 0 && W > 0 && C > 0 && s >= 1, "tpgsr_resize_nearest_fwd: bad arguments");
  long long total = (long long)N * H * s * W * s * C;
  hipLaunchKernelGGL(resize_nearest_fwd_kernel, dim3((int)min((long long)8192, (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, N,
                     H, W, C, s, out);
  TPGSR_LAUNCH_CHECK("tpgsr_resize_nearest_fwd");
}
extern "C" int tpgsr_resize_nearest_bwd(const float* dout, int N, int H, int W, int C, int s, float* din, void* stream) {
  TPGSR_CHECK_ARG(dout && din && N > 0 && H > 0 && W > 0 && C > 0 && s >= 1, "tpgsr_resize_nearest_bwd: bad arguments");
  long long total = (long long)N * H * W * C;
  hipLaunchKernelGGL(resize_nearest_bwd_kernel, dim3((int)min((long long)8192, (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dout,
                     N, H, W, C, s, din);
  TPGSR_LAUNCH_CHECK("tpgsr_resize_nearest_bwd");
}

// F.interpolate(x, (OH, OW), mode='bilinear', align_corners=True), NHWC
__device__ __forceinline__ void lin_src(int o, int in_size, int out_size, int& i0, int& i1, float& l) {
  float s = out_size > 1 ? (float)o * ((float)(in_size - 1) / (float)(out_size - 1)) : 0.f;
  i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + 1 < in_size ? i0 + 1 : i0;
  l = s - (float)i0;
}
__global__ __launch_bounds__(256) void resize_bilinear_fwd_kernel(const float* __restrict__ in, int N, int H, int W, int C, int OH, int OW,
                                                                  float* __restrict__ out) {
  long long total = (long long)N * OH * OW * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    long long p = i / C;
    int ow = (int)(p % OW);
    p /= OW;
    int oh = (int)(p % OH);
    int n = (int)(p / OH);
    int y0, y1, x0, x1;
    float ly, lx;
    lin_src(oh, H, OH, y0, y1, ly);
    lin_src(ow, W, OW, x0, x1, lx);
    const float* b = in + (size_t)n * H * W * C + c;
    float v00 = b[((size_t)y0 * W + x0) * C], v01 = b[((size_t)y0 * W + x1) * C];
    float v10 = b[((size_t)y1 * W + x0) * C], v11 = b[((size_t)y1 * W + x1) * C];
    out[i] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
  }
}
// gather form: every input element sums the outputs that sampled it (outputs sampling row y lie in a contiguous range)
__device__ __forceinline__ void lin_out_range(int i, int in_size, int out_size, int& lo, int& hi) {
  if (in_size <= 1 || out_size <= 1) {
    lo = 0;
    hi = out_size - 1;
    return;
  }
  float inv = (float)(out_size - 1) / (float)(in_size - 1);
  lo = (int)floorf(((float)i - 1.f) * inv) - 1;
  hi = (int)ceilf(((float)i + 1.f) * inv) + 1;
  lo = lo < 0 ? 0 : lo;
  hi = hi > out_size - 1 ? out_size - 1 : hi;
}
__global__ __launch_bounds__(256) void resize_bilinear_bwd_kernel(const float* __restrict__ dout, int N, int H, int W, int C, int OH, int OW,
                                                                  float* __restrict__ din) {
  long long total = (long long)N * H * W * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    long long p = i / C;
    int x = (int)(p % W);
    p /= W;
    int y = (int)(p % H);
    int n = (int)(p / H);
    int oh_lo, oh_hi, ow_lo, ow_hi;
    lin_out_range(y, H, OH, oh_lo, oh_hi);
    lin_out_range(x, W, OW, ow_lo, ow_hi);
    float g = 0.f;
    for (int oh = oh_lo; oh <= oh_hi; ++oh) {
      int y0, y1;
      float ly;
      lin_src(oh, H, OH, y0, y1, ly);
      float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
      if (wy == 0.f) continue;
      float rs = 0.f;
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        int x0, x1;
        float lx;
        lin_src(ow, W, OW, x0, x1, lx);
        float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
        if (wx != 0.f) rs += wx * dout[(((size_t)n * OH + oh) * OW + ow) * C + c];
      }
      g += wy * rs;
    }
    din[i] = g;
  }
}

extern "C" int tpgsr_resize_bilinear_fwd(const float* in, int N, int H, int W, int C, int OH, int OW, float* out, void* stream) {
  TPGSR_CHECK_ARG(in && out && N > 0 && H > 0 && W > 0 && C > 0 && OH > 0 && OW > 0, "tpgsr_resize_bilinear_fwd: bad arguments");
  long long total = (long long)N * OH * OW * C;
  hipLaunchKernelGGL(resize_bilinear_fwd_kernel, dim3((int)min((long long)8192, (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, N,
                     H, W, C, OH, OW, out);
  TPGSR_LAUNCH_CHECK("tpgsr_resize_bilinear_fwd");
}
extern "C" int tpgsr_resize_bilinear_bwd(const float* dout, int N, int H, int W, int C, int OH, int OW, float* din, void* stream) {
  TPGSR_CHECK_ARG(dout && din && N > 0 && H > 0 && W > 0 && C > 0 && OH > 0 && OW > 0, "tpgsr_resize_bilinear_bwd: bad arguments");
  long long total = (long long)N * H * W * C;
  hipLaunchKernelGGL(resize_bilinear_bwd_kernel, dim3((int)min((long long)8192, (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dout,
                     N, H, W, C, OH, OW, din);
  TPGSR_LAUNCH_CHECK("tpgsr_resize_bilinear_bwd");
}

// zero-dilation: out [N][(H-1)*sh+1][(W-1)*sw+1][C], out[h*sh][w*sw] = in[h][w], 0 elsewhere; sub-sampling is its adjoint:
// out [N][ceil(H/sh)][ceil(W/sw)][C] = in[h*sh][w*sw]
__global__ __launch_bounds__(256) void dilate2d_kernel(const float* __restrict__ in, int N, int H, int W, int C, int sh, int sw,
                                                       float* __restrict__ out) {
  const int OH = (H - 1) * sh + 1, OW = (W - 1) * sw + 1;
  long long total = (long long)N * OH * OW * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    long long p = i / C;
    int ow = (int)(p % OW);
    p /= OW;
    int oh = (int)(p % OH);
    int n = (int)(p / OH);
    out[i] = (oh % sh == 0 && ow % sw == 0) ? in[(((size_t)n * H + oh / sh) * W + ow / sw) * C + c] : 0.f;
  }
}
__global__ __launch_bounds__(256) void subsample2d_kernel(const float* __restrict__ in, int N, int H, int W, int C, int sh, int sw,
                                                          float* __restrict__ out) {
  const int OH = (H + sh - 1) / sh, OW = (W + sw - 1) / sw;
  long long total = (long long)N * OH * OW * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    long long p = i / C;
    int ow = (int)(p % OW);
    p /= OW;
    int oh = (int)(p % OH);
    int n = (int)(p / OH);
    out[i] = in[(((size_t)n * H + oh * sh) * W + ow * sw) * C + c];
  }
}

extern "C" int tpgsr_dilate2d(const float* in, int N, int H, int W, int C, int sh, int sw, float* out, void* stream) {
  TPGSR_CHECK_ARG(in && out && N > 0 && H > 0 && W > 0 && C > 0 && sh >= 1 && sw >= 1, "tpgsr_dilate2d: bad arguments");
  long long total = (long long)N * ((H - 1) * sh + 1) * ((W - 1) * sw + 1) * C;
  hipLaunchKernelGGL(dilate2d_kernel, dim3((int)min((long long)8192, (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, N, H, W, C,
                     sh, sw, out);
  TPGSR_LAUNCH_CHECK("tpgsr_dilate2d");
}
extern "C" int tpgsr_subsample2d(const float* in, int N, int H, int W, int C, int sh, int sw, float* out, void* stream) {
  TPGSR_CHECK_ARG(in && out && N > 0 && H > 0 && W > 0 && C > 0 && sh >= 1 && sw >= 1, "tpgsr_subsample2d: bad arguments");
  long long total = (long long)N * ((H + sh - 1) / sh) * ((W + sw - 1) / sw) * C;
  hipLaunchKernelGGL(subsample2d_kernel, dim3((int)min((long long)8192, (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, N, H, W,
                     C, sh, sw, out);
  TPGSR_LAUNCH_CHECK("tpgsr_subsample2d");
}

// out[n][w][c] = scale * sum_h in[n][h][w][c]  (mean over H with scale = 1/H); broadcast back: din[n][h][w][c] = scale * dout[n][w][c]
__global__ __launch_bounds__(256) void hreduce_kernel(const float* __restrict__ in, int N, int H, int W, int C, float scale,
                                                      float* __restrict__ out) {
  long long total = (long long)N * W * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    long long wc = i % ((long long)W * C);
    int n = (int)(i / ((long long)W * C));
    float s = 0.f;
    for (int h = 0; h < H; ++h) s += in[((size_t)n * H + h) * W * C + wc];
    out[i] = s * scale;
  }
}
__global__ __launch_bounds__(256) void hbroadcast_kernel(const float* __restrict__ dout, int N, int H, int W, int C, float scale,
                                                         float* __restrict__ din) {
  long long total = (long long)N * H * W * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    long long wc = i % ((long long)W * C);
    int n = (int)(i / ((long long)H * W * C));
    din[i] = scale * dout[(size_t)n * W * C + wc];
  }
}

extern "C" int tpgsr_hreduce(const float* in, int N, int H, int W, int C, float scale, float* out, void* stream) {
  TPGSR_CHECK_ARG(in && out && N > 0 && H > 0 && W > 0 && C > 0, "tpgsr_hreduce: bad arguments");
  long long total = (long long)N * W * C;
  hipLaunchKernelGGL(hreduce_kernel, dim3((int)min((long long)8192, (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, N, H, W, C,
                     scale, out);
  TPGSR_LAUNCH_CHECK("tpgsr_hreduce");
}
extern "C" int tpgsr_hbroadcast(const float* dout, int N, int H, int W, int C, float scale, float* din, void* stream) {
  TPGSR_CHECK_ARG(dout && din && N > 0 && H > 0 && W > 0 && C > 0, "tpgsr_hbroadcast: bad arguments");
  long long total = (long long)N * H * W * C;
  hipLaunchKernelGGL(hbroadcast_kernel, dim3((int)min((long long)8192, (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dout, N, H, W,
                     C, scale, din);
  TPGSR_LAUNCH_CHECK("tpgsr_hbroadcast");
}
