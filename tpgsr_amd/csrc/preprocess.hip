// Input pipeline kernel: the reference's `resizeNormalize` (dataset/dataset.py:615-632) --
// Pillow bicubic resize of a uint8 HWC image to the network size, ToTensor (/255, CHW) and the luminance-threshold mask
// channel -- for a whole batch of variable-size images on the device, bit-exact against Pillow's 8-bit resampling
// (two separable passes in 22-bit fixed point with a clamp after each, src/libImaging/Resample.c: restated in
// oracle/input_pipeline.py and pinned against the installed Pillow).  The reference runs this per image with PIL on ONE
// DataLoader worker (`workers: 1`); here the host only concatenates the encoded-size uint8 pixels and the coefficient tables.
#include "common.h"
#include <math.h>

#define RS_PRECISION_BITS 22

// ---- host: Pillow's precompute_coeffs + normalize_coeffs_8bpc for the bicubic filter ---------------------------------------------
static double rs_bicubic(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

extern "C" int tpgsr_resample_ksize(int in_size, int out_size) {
  double filterscale = (double)in_size / (double)out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  return (int)ceil(2.0 * filterscale) * 2 + 1;
}

// bounds [out_size][2] = (first input index, tap count); kk [out_size][ksize] = 22-bit fixed-point coefficients (0 padded)
#pragma clang fp contract(off)
extern "C" int tpgsr_resample_coeffs(int in_size, int out_size, int* bounds, int* kk) {
  if (in_size <= 0 || out_size <= 0 || !bounds || !kk) {
    tpgsr_set_error("tpgsr_resample_coeffs: bad arguments");
    return TPGSR_ERR_ARG;
  }
  const double scale = (double)in_size / (double)out_size;
  double filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 2.0 * filterscale;
  const int ksize = (int)ceil(support) * 2 + 1;
  double w[1024];
  if (ksize > 1024) {
    tpgsr_set_error("tpgsr_resample_coeffs: shrink factor too large");
    return TPGSR_ERR_ARG;
  }
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = (xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
      w[x] = rs_bicubic((x + xmin - center + 0.5) * ss);
      ww += w[x];
    }
    int* k = kk + (size_t)xx * ksize;
    for (int x = 0; x < xmax; ++x) {
      double v = ww != 0.0 ? w[x] / ww : w[x];
      k[x] = v < 0 ? (int)(-0.5 + v * (double)(1 << RS_PRECISION_BITS)) : (int)(0.5 + v * (double)(1 << RS_PRECISION_BITS));
    }
    for (int x = xmax; x < ksize; ++x) k[x] = 0;
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
  }
  return 0;
}

// ---- device -----------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned char rs_clip8(int acc) {
  int v = acc >> RS_PRECISION_BITS;
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: tmp[n][y][xx][c], y < H_n
__global__ __launch_bounds__(256) void resize_h_kernel(const unsigned char* __restrict__ pix, const tpgsr_image_desc* __restrict__ descs,
                                                       const int* __restrict__ tab, int N, int OW, int maxH, unsigned char* __restrict__ tmp) {
  const long long total = (long long)N * maxH * OW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int xx = (int)(i % OW);
    const long long r = i / OW;
    const int y = (int)(r % maxH);
    const int n = (int)(r / maxH);
    const tpgsr_image_desc d = descs[n];
    if (y >= d.H) continue;
    const int* b = tab + d.xb_off + 2 * xx;
    const int* k = tab + d.xk_off + xx * d.kx;
    const unsigned char* row = pix + d.offset + ((size_t)y * d.W + b[0]) * 3;
    int a0 = 1 << (RS_PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int x = 0; x < b[1]; ++x) {
      const int kv = k[x];
      a0 += row[3 * x] * kv;
      a1 += row[3 * x + 1] * kv;
      a2 += row[3 * x + 2] * kv;
    }
    unsigned char* o = tmp + (((size_t)n * maxH + y) * OW + xx) * 3;
    o[0] = rs_clip8(a0);
    o[1] = rs_clip8(a1);
    o[2] = rs_clip8(a2);
  }
}

// vertical pass + ToTensor: out [N][C][OH][OW] (channels 0..2), res8 [N][OH][OW][3]
__global__ __launch_bounds__(256) void resize_v_kernel(const unsigned char* __restrict__ tmp, const tpgsr_image_desc* __restrict__ descs,
                                                       const int* __restrict__ tab, int N, int OH, int OW, int maxH, int C,
                                                       unsigned char* __restrict__ res8, float* __restrict__ out) {
  const long long total = (long long)N * OH * OW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int xx = (int)(i % OW);
    const long long r = i / OW;
    const int yy = (int)(r % OH);
    const int n = (int)(r / OH);
    const tpgsr_image_desc d = descs[n];
    const int* b = tab + d.yb_off + 2 * yy;
    const int* k = tab + d.yk_off + yy * d.ky;
    const unsigned char* col = tmp + (((size_t)n * maxH + b[0]) * OW + xx) * 3;
    int a0 = 1 << (RS_PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int y = 0; y < b[1]; ++y) {
      const int kv = k[y];
      const unsigned char* p = col + (size_t)y * OW * 3;
      a0 += p[0] * kv;
      a1 += p[1] * kv;
      a2 += p[2] * kv;
    }
    const unsigned char v0 = rs_clip8(a0), v1 = rs_clip8(a1), v2 = rs_clip8(a2);
    unsigned char* o8 = res8 + i * 3;
    o8[0] = v0; o8[1] = v1; o8[2] = v2;
    const size_t plane = (size_t)OH * OW;
    float* o = out + (size_t)n * C * plane + (size_t)yy * OW + xx;
    o[0] = (float)v0 / 255.f;
    o[plane] = (float)v1 / 255.f;
    o[2 * plane] = (float)v2 / 255.f;
  }
}

// mask channel: L = (19595 R + 38470 G + 7471 B + 32768) >> 16; 1.0 where L <= mean(L) over the image, else 0.0 (one block per image)
__global__ __launch_bounds__(256) void luma_mask_kernel(const unsigned char* __restrict__ res8, int OH, int OW, int C, float* __restrict__ out) {
  __shared__ long long red[4];
  const int n = blockIdx.x;
  const int P = OH * OW;
  const unsigned char* p = res8 + (size_t)n * P * 3;
  long long s = 0;
  for (int i = threadIdx.x; i < P; i += 256) s += (p[3 * i] * 19595 + p[3 * i + 1] * 38470 + p[3 * i + 2] * 7471 + 0x8000) >> 16;
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  const long long sum = (red[0] + red[1]) + (red[2] + red[3]);
  float* m = out + ((size_t)n * C + 3) * P;
  for (int i = threadIdx.x; i < P; i += 256) {
    const long long L = (p[3 * i] * 19595 + p[3 * i + 1] * 38470 + p[3 * i + 2] * 7471 + 0x8000) >> 16;
    m[i] = (L * P > sum) ? 0.f : 1.f;          // L > mean  <=>  L * P > sum, in integers
  }
}

extern "C" int tpgsr_resize_normalize(const unsigned char* pixels, const tpgsr_image_desc* descs_dev, const int* tables_dev, int N, int OH,
                                      int OW, int maxH, int mask, unsigned char* tmp, unsigned char* res8, float* out, void* stream) {
  TPGSR_CHECK_ARG(pixels && descs_dev && tables_dev && tmp && res8 && out && N > 0 && OH > 0 && OW > 0 && maxH > 0,
                  "tpgsr_resize_normalize: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int C = mask ? 4 : 3;
  long long t1 = (long long)N * maxH * OW, t2 = (long long)N * OH * OW;
  hipLaunchKernelGGL(resize_h_kernel, dim3((int)min((long long)4096, (t1 + 255) / 256)), dim3(256), 0, st, pixels, descs_dev, tables_dev, N, OW,
                     maxH, tmp);
  hipLaunchKernelGGL(resize_v_kernel, dim3((int)min((long long)4096, (t2 + 255) / 256)), dim3(256), 0, st, tmp, descs_dev, tables_dev, N, OH, OW,
                     maxH, C, res8, out);
  if (mask) hipLaunchKernelGGL(luma_mask_kernel, dim3(N), dim3(256), 0, st, res8, OH, OW, C, out);
  TPGSR_LAUNCH_CHECK("tpgsr_resize_normalize");
}
