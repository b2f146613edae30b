// Evaluation-path reductions: on-device CTC greedy decoding of the recogniser's logits
// (utils/metrics.py:71-88 get_string_crnn), PSNR (utils/ssim_psnr.py:9-15) and SSIM (utils/ssim_psnr.py:18-78) of the SR image
// against the HR image -- the three things interfaces/super_resolution.py:770-900 computes per evaluation batch.  Deterministic
// (fixed-order block partials, fp64 combine).
#include "common.h"

// logits [N][T][C] batch-major.  labels [N][T] (collapsed, blank-free class indices, -1 padded), lengths [N].
// Per time step the FIRST maximum wins (torch.max); a class is emitted when it differs from the previously emitted one and is not
// the blank 0; a blank resets "previous" (so "a-a" decodes to "aa", "aa" to "a").
__global__ __launch_bounds__(64) void ctc_greedy_kernel(const float* __restrict__ logits, int N, int T, int C, int* __restrict__ labels,
                                                        int* __restrict__ lengths) {
  __shared__ int amax[1024];
  const int n = blockIdx.x;
  for (int t = threadIdx.x; t < T; t += 64) {
    const float* p = logits + ((size_t)n * T + t) * C;
    float best = p[0];
    int bi = 0;
    for (int c = 1; c < C; ++c) {
      float v = p[c];
      if (v > best) {
        best = v;
        bi = c;
      }
    }
    amax[t] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int last = -1, len = 0;
    for (int t = 0; t < T; ++t) {
      const int i = amax[t];
      if (i != last) {
        if (i != 0) {
          labels[(size_t)n * T + len++] = i;
          last = i;
        } else {
          last = -1;
        }
      }
    }
    lengths[n] = len;
    for (int t = len; t < T; ++t) labels[(size_t)n * T + t] = -1;
  }
}

extern "C" int tpgsr_ctc_greedy_decode(const float* logits, int N, int T, int C, int* labels, int* lengths, void* stream) {
  TPGSR_CHECK_ARG(logits && labels && lengths && N > 0 && T > 0 && T <= 1024 && C > 0, "tpgsr_ctc_greedy_decode: bad arguments (T <= 1024)");
  hipLaunchKernelGGL(ctc_greedy_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, logits, N, T, C, labels, lengths);
  TPGSR_LAUNCH_CHECK("tpgsr_ctc_greedy_decode");
}

// partial[blk] = sum over this block's share of ((a - b) * 255)^2 over the first min(Ctot, 3) channels of NCHW images
__global__ __launch_bounds__(256) void psnr_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, int N, int Ctot, int HW,
                                                           double* __restrict__ partial) {
  __shared__ double red[4];
  const int Cc = Ctot < 3 ? Ctot : 3;
  const long long total = (long long)N * Cc * HW;
  double s = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int hw = (int)(i % HW);
    const long long r = i / HW;
    const int c = (int)(r % Cc);
    const int n = (int)(r / Cc);
    const size_t idx = ((size_t)n * Ctot + c) * HW + hw;
    const float d = a[idx] * 255.f - b[idx] * 255.f;
    s += (double)(d * d);
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void psnr_finalize_kernel(const double* __restrict__ partial, int nblk, double count, float* out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 64) s += partial[i];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) {
    const double mse = s / count;
    out[0] = mse == 0.0 ? INFINITY : (float)(20.0 * log10(255.0 / sqrt(mse)));
  }
}

extern "C" int tpgsr_psnr(const float* a, const float* b, int N, int Ctot, int H, int W, double* partial, int nblk, float* out, void* stream) {
  TPGSR_CHECK_ARG(a && b && partial && out && N > 0 && Ctot > 0 && H > 0 && W > 0 && nblk > 0, "tpgsr_psnr: bad arguments");
  const int Cc = Ctot < 3 ? Ctot : 3;
  hipLaunchKernelGGL(psnr_partial_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, a, b, N, Ctot, H * W, partial);
  hipLaunchKernelGGL(psnr_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, nblk, (double)N * Cc * H * W, out);
  TPGSR_LAUNCH_CHECK("tpgsr_psnr");
}

// SSIM with a KS x KS window (the reference: 11 x 11 Gaussian, sigma 1.5, zero padding, per channel, first min(Ctot,3) channels):
// partial[blk] = sum of the ssim map over this block's pixels; mean = sum / (N * Cc * H * W)
__global__ __launch_bounds__(256) void ssim_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ win,
                                                           int KS, int N, int Ctot, int H, int W, double* __restrict__ partial) {
  __shared__ double red[4];
  __shared__ float w_s[33 * 33];
  for (int i = threadIdx.x; i < KS * KS; i += 256) w_s[i] = win[i];
  __syncthreads();
  const int Cc = Ctot < 3 ? Ctot : 3;
  const long long total = (long long)N * Cc * H * W;
  const int R = KS / 2;
  const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  double s = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    long long r = i / W;
    const int y = (int)(r % H);
    r /= H;
    const int c = (int)(r % Cc);
    const int n = (int)(r / Cc);
    const float* pa = a + ((size_t)n * Ctot + c) * H * W;
    const float* pb = b + ((size_t)n * Ctot + c) * H * W;
    float m1 = 0.f, m2 = 0.f, s11 = 0.f, s22 = 0.f, s12 = 0.f;
    for (int ky = 0; ky < KS; ++ky) {
      const int yy = y + ky - R;
      if ((unsigned)yy >= (unsigned)H) continue;
      for (int kx = 0; kx < KS; ++kx) {
        const int xx = x + kx - R;
        if ((unsigned)xx >= (unsigned)W) continue;
        const float w = w_s[ky * KS + kx];
        const float u = pa[(size_t)yy * W + xx], v = pb[(size_t)yy * W + xx];
        m1 += w * u;
        m2 += w * v;
        s11 += w * u * u;
        s22 += w * v * v;
        s12 += w * u * v;
      }
    }
    const float m11 = m1 * m1, m22 = m2 * m2, m12 = m1 * m2;
    const float v = ((2.f * m12 + C1) * (2.f * (s12 - m12) + C2)) / ((m11 + m22 + C1) * ((s11 - m11) + (s22 - m22) + C2));
    s += (double)v;
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void mean_finalize_kernel(const double* __restrict__ partial, int nblk, double count, float* out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 64) s += partial[i];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) out[0] = (float)(s / count);
}

extern "C" int tpgsr_ssim(const float* a, const float* b, const float* window, int KS, int N, int Ctot, int H, int W, double* partial, int nblk,
                          float* out, void* stream) {
  TPGSR_CHECK_ARG(a && b && window && partial && out && KS > 0 && KS <= 33 && (KS & 1) && N > 0 && Ctot > 0 && nblk > 0,
                  "tpgsr_ssim: bad arguments (odd window <= 33)");
  const int Cc = Ctot < 3 ? Ctot : 3;
  hipLaunchKernelGGL(ssim_partial_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, a, b, window, KS, N, Ctot, H, W, partial);
  hipLaunchKernelGGL(mean_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, nblk, (double)N * Cc * H * W, out);
  TPGSR_LAUNCH_CHECK("tpgsr_ssim");
}

// ------------------------------------------------------------------------------------------------------
// SSIM as a LOSS (round 6): `--ssim_loss`, interfaces/super_resolution.py:388-391: loss_ssim = (1 - ssim(sr, hr).mean()) * 10 -- the
// gradient of the mean SSIM with respect to the first image.  With mu1 = w * a, mu2 = w * b, e11 = w * a^2, e22 = w * b^2, e12 = w * a b
// (w * . = the KS x KS window sum with zero padding) and S = A1 A2 / (B1 B2), A1 = 2 mu1 mu2 + C1, A2 = 2 (e12 - mu1 mu2) + C2,
// B1 = mu1^2 + mu2^2 + C1, B2 = (e11 - mu1^2) + (e22 - mu2^2) + C2:
//   pass 1 (every map pixel): G0 = dS/dmu1, G1 = dS/de11, G2 = dS/de12
//   pass 2 (every image pixel q): dS_total/da(q) = (w * G0)(q) + 2 a(q) (w * G1)(q) + b(q) (w * G2)(q)       (the window is symmetric)
// utils/ssim_psnr.py:30-50 (_ssim) is what autograd differentiates in the reference.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ssim_grad_maps_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ win,
                                                             int KS, int N, int Ctot, int H, int W, float* __restrict__ gm) {
  __shared__ float w_s[33 * 33];
  for (int i = threadIdx.x; i < KS * KS; i += 256) w_s[i] = win[i];
  __syncthreads();
  const int Cc = Ctot < 3 ? Ctot : 3;
  const long long total = (long long)N * Cc * H * W;
  const int R = KS / 2;
  const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    long long r = i / W;
    const int y = (int)(r % H);
    r /= H;
    const int c = (int)(r % Cc);
    const int n = (int)(r / Cc);
    const float* pa = a + ((size_t)n * Ctot + c) * H * W;
    const float* pb = b + ((size_t)n * Ctot + c) * H * W;
    float m1 = 0.f, m2 = 0.f, s11 = 0.f, s22 = 0.f, s12 = 0.f;
    for (int ky = 0; ky < KS; ++ky) {
      const int yy = y + ky - R;
      if ((unsigned)yy >= (unsigned)H) continue;
      for (int kx = 0; kx < KS; ++kx) {
        const int xx = x + kx - R;
        if ((unsigned)xx >= (unsigned)W) continue;
        const float w = w_s[ky * KS + kx];
        const float u = pa[(size_t)yy * W + xx], v = pb[(size_t)yy * W + xx];
        m1 += w * u;
        m2 += w * v;
        s11 += w * u * u;
        s22 += w * v * v;
        s12 += w * u * v;
      }
    }
    const float m11 = m1 * m1, m22 = m2 * m2, m12 = m1 * m2;
    const float A1 = 2.f * m12 + C1, A2 = 2.f * (s12 - m12) + C2, B1 = m11 + m22 + C1, B2 = (s11 - m11) + (s22 - m22) + C2;
    const float inv = 1.f / (B1 * B2), S = A1 * A2 * inv;
    // d/dmu1: dA1 = 2 mu2, dA2 = -2 mu2, dB1 = 2 mu1, dB2 = -2 mu1
    const float g0 = (2.f * m2 * A2 - 2.f * m2 * A1) * inv - S * (2.f * m1 / B1 - 2.f * m1 / B2);
    const float g1 = -S / B2;                 // d/de11: dB2 = 1
    const float g2 = 2.f * A1 * inv;          // d/de12: dA2 = 2
    gm[i] = g0;
    gm[total + i] = g1;
    gm[2 * total + i] = g2;
  }
}

__global__ __launch_bounds__(256) void ssim_grad_apply_kernel(const float* __restrict__ gm, const float* __restrict__ a, const float* __restrict__ b,
                                                              const float* __restrict__ win, int KS, int N, int Ctot, int H, int W,
                                                              const float* __restrict__ coef, float mult, float* __restrict__ da, int accumulate) {
  __shared__ float w_s[33 * 33];
  for (int i = threadIdx.x; i < KS * KS; i += 256) w_s[i] = win[i];
  __syncthreads();
  const int Cc = Ctot < 3 ? Ctot : 3;
  const long long total = (long long)N * Cc * H * W;
  const int R = KS / 2;
  const float k = mult * (coef ? coef[0] : 1.f);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    long long r = i / W;
    const int y = (int)(r % H);
    r /= H;
    const int c = (int)(r % Cc);
    const int n = (int)(r / Cc);
    const float* g0 = gm + ((size_t)n * Cc + c) * H * W;
    const float* g1 = g0 + total;
    const float* g2 = g1 + total;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
    for (int ky = 0; ky < KS; ++ky) {
      const int yy = y - (ky - R);             // map pixel p = q - (tap offset): the adjoint of the window sum
      if ((unsigned)yy >= (unsigned)H) continue;
      for (int kx = 0; kx < KS; ++kx) {
        const int xx = x - (kx - R);
        if ((unsigned)xx >= (unsigned)W) continue;
        const float w = w_s[ky * KS + kx];
        const size_t o = (size_t)yy * W + xx;
        t0 += w * g0[o];
        t1 += w * g1[o];
        t2 += w * g2[o];
      }
    }
    const size_t q = (((size_t)n * Ctot + c) * H + y) * W + x;
    const float v = k * (t0 + 2.f * a[q] * t1 + b[q] * t2);
    da[q] = accumulate ? da[q] + v : v;
  }
}

/* da[:, :min(Ctot,3)] (+)= mult * coef[0] * d(sum of the SSIM map)/da  (NCHW, Ctot channels per image; channels >= 3 of `da` are left alone);
 * gm: scratch of 3 * N * min(Ctot,3) * H * W floats; coef: optional device scalar (the upstream gradient).  For the mean SSIM pass
 * mult = 1 / (N min(Ctot,3) H W); for the reference's loss (1 - ssim.mean()) * 10, mult = -10 / (N min(Ctot,3) H W). */
extern "C" int tpgsr_ssim_bwd(const float* a, const float* b, const float* window, int KS, int N, int Ctot, int H, int W, float* gm,
                              const float* coef, float mult, float* da, int accumulate, void* stream) {
  TPGSR_CHECK_ARG(a && b && window && gm && da && KS > 0 && KS <= 33 && (KS & 1) && N > 0 && Ctot > 0 && H > 0 && W > 0,
                  "tpgsr_ssim_bwd: bad arguments (odd window <= 33)");
  const int Cc = Ctot < 3 ? Ctot : 3;
  const long long total = (long long)N * Cc * H * W;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(ssim_grad_maps_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, b, window, KS, N, Ctot, H, W, gm);
  hipLaunchKernelGGL(ssim_grad_apply_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, gm, a, b, window, KS, N, Ctot, H, W, coef, mult, da,
                     accumulate);
  TPGSR_LAUNCH_CHECK("tpgsr_ssim_bwd");
}
