// Tail of the SR network (shift-sum + tanh of the folded 9x9 conv), the image loss
// (loss/image_loss.py:10-51) and the optimiser step (clip_grad_norm_ + Adam over flat arenas).
#include "common.h"

// ------------------------------------------------------------------------------------------------------
// tail: out[n][co][h][w] = tanh(bias[co] + sum_kw P[n][h][w+kw-KS/2][kw*Co+co])   (NCHW output)
//
// Round 5: ONE WORKGROUP PER IMAGE ROW (n, h).  The first form gave every output element a thread that decoded its position with three
// 64-bit divisions and read its KS addends 144 bytes apart (33 us for 28 MB in, 3 MB out); here the row's [W][KS Co] block of P is staged
// into LDS by coalesced 16-byte loads (row pitch KS Co + 1 floats: the strided reads that follow are conflict-free) and summed from there;
// a row too wide for the LDS block is cut into chunks (one workgroup each).
// The same code without bias / tanh into an NHWC map is tpgsr_shiftsum_nhwc (block1's folded data gradient).  Sums run kw = 0 .. KS-1
// as before: same bits.
// ------------------------------------------------------------------------------------------------------
#define TAIL_MAX_ROW (160 * 37)      // floats of LDS per row: W (KS Co + 1) <= 5920 (the 32 x 128 output of the hot path: 128 x 37)

// rows wider than the LDS block are cut into chunks of WC output columns (a workgroup per (row, chunk) stages WC + KS - 1 columns)
static inline int tail_chunk(int W, int pitch, int half) {
  if (W * pitch <= TAIL_MAX_ROW) return W;
  return TAIL_MAX_ROW / pitch - 2 * half;
}

template <bool NCHW_TANH>
__global__ __launch_bounds__(256) void shiftsum_row_kernel(const float* __restrict__ P, const float* __restrict__ bias, int H, int W, int Co,
                                                           int KS, int WC, int nchunk, float* __restrict__ out) {
  __shared__ float row[TAIL_MAX_ROW];
  const int NP = KS * Co, pitch = NP + 1, half = KS / 2;
  const int rid = blockIdx.x / nchunk, x0 = (blockIdx.x - rid * nchunk) * WC;      // (nchunk == 1 on the hot path)
  const int n = rid / H, h = rid - n * H;
  const int xs0 = max(0, x0 - half), xs1 = min(W, x0 + WC + half), wc = min(WC, W - x0);
  const float* src = P + ((size_t)rid * W + xs0) * NP;
  if ((NP & 3) == 0) {      // (workgroup-uniform) the hot path: KS Co = 36, rows start 16-byte aligned and a 16-byte group never straddles two pixels
    for (int e = threadIdx.x; e < ((xs1 - xs0) * NP) >> 2; e += 256) {
      const float4 v = *reinterpret_cast<const float4*>(src + 4 * e);
      const int x = (4 * e) / NP, c = 4 * e - x * NP;
      float* d = row + x * pitch + c;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
  } else {                  // any other column count (three-channel networks, mask=False: KS Co = 27): element by element, still coalesced
    for (int e = threadIdx.x; e < (xs1 - xs0) * NP; e += 256) {
      const int x = e / NP;
      row[x * pitch + (e - x * NP)] = src[e];
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < wc * Co; e += 256) {
    int co, w;
    if (NCHW_TANH) {
      co = e / wc;
      w = e - co * wc;
    } else {
      w = e / Co;
      co = e - w * Co;
    }
    w += x0;
    float s = (NCHW_TANH && bias) ? bias[co] : 0.f;
    for (int kw = 0; kw < KS; ++kw) {
      const int x = w + kw - half;
      if ((unsigned)x < (unsigned)W) s += row[(x - xs0) * pitch + kw * Co + co];
    }
    if (NCHW_TANH) out[(((size_t)n * Co + co) * H + h) * W + w] = tanh_f(s);
    else out[((size_t)rid * W + w) * Co + co] = s;
  }
}

static int shiftsum_launch(const char* who, bool tail, const float* P, const float* bias, int N, int H, int W, int Co, int KS, float* out,
                           void* stream) {
  TPGSR_CHECK_ARG(P && out && N > 0 && H > 0 && W > 0 && Co > 0 && (KS & 1), "%s: bad arguments", who);
  const int pitch = KS * Co + 1, half = KS / 2, WC = tail_chunk(W, pitch, half);
  TPGSR_CHECK_ARG(WC >= 1 && (((uintptr_t)P) & 15) == 0,
                  "%s: needs a 16-byte aligned P and KS (KS Co + 1) <= %d (got KS %d, Co %d)", who, TAIL_MAX_ROW, KS, Co);
  const int nchunk = (W + WC - 1) / WC;
  if (tail)
    hipLaunchKernelGGL(shiftsum_row_kernel<true>, dim3(N * H * nchunk), dim3(256), 0, (hipStream_t)stream, P, bias, H, W, Co, KS, WC, nchunk,
                       out);
  else
    hipLaunchKernelGGL(shiftsum_row_kernel<false>, dim3(N * H * nchunk), dim3(256), 0, (hipStream_t)stream, P, nullptr, H, W, Co, KS, WC,
                       nchunk, out);
  TPGSR_LAUNCH_CHECK(who);
}

extern "C" int tpgsr_tail_shiftsum_tanh(const float* P, const float* bias, int N, int H, int W, int Co, int KS,
                                        float* out_nchw, void* stream) {
  return shiftsum_launch("tpgsr_tail_shiftsum_tanh", true, P, bias, N, H, W, Co, KS, out_nchw, stream);
}

extern "C" int tpgsr_shiftsum_nhwc(const float* P, int N, int H, int W, int Co, int KS, float* out, void* stream) {
  return shiftsum_launch("tpgsr_shiftsum_nhwc", false, P, nullptr, N, H, W, Co, KS, out, stream);
}

// dP[n][h][x][kw*Co+co] = dpre[n][co][h][x-kw+KS/2], dpre = dout*(1-out^2); dbias partial per (n, h) row.
// One workgroup per image row as well: the row's Co x W values of dpre are computed once (coalesced NCHW reads) into LDS with KS/2 zeros
// on either side, the [W][KS Co] block of dP is written from there in coalesced order, and the bias-gradient partial of the row is each
// channel's sum over w in a FIXED order (a shuffle tree per wave over a fixed assignment, the waves added in order -- until round 4's fix an
// LDS atomicAdd: "the bit flip of round 3").  The first form spent its time in four 64-bit divisions per element (37 us).
#define TAIL_BWD_WC 160      // output columns per workgroup (wider rows are cut into chunks; the hot path's 128 is one)
__global__ __launch_bounds__(256) void tail_bwd_kernel(const float* __restrict__ out, const float* __restrict__ dout, int H, int W, int Co,
                                                       int KS, int WC, int nchunk, float* __restrict__ dP, float* __restrict__ dbp) {
  __shared__ float dpre[8 * (TAIL_BWD_WC + 16)];
  const int NP = KS * Co, half = KS / 2;
  const int rid = blockIdx.x / nchunk, x0 = (blockIdx.x - rid * nchunk) * WC, wc = min(WC, W - x0), pitch = wc + 2 * half;
  const int n = rid / H, h = rid - n * H;
  for (int e = threadIdx.x; e < Co * pitch; e += 256) {
    const int co = e / pitch, x = x0 + e - co * pitch - half;
    float v = 0.f;
    if ((unsigned)x < (unsigned)W) {
      const size_t o = (((size_t)n * Co + co) * H + h) * W + x;
      const float y = out[o];
      v = dout[o] * (1.f - y * y);
    }
    dpre[e] = v;
  }
  __syncthreads();
  float* dst = dP + ((size_t)rid * W + x0) * NP;
  for (int e = threadIdx.x; e < wc * NP; e += 256) {
    const int x = e / NP, np = e - x * NP;
    const int kw = np / Co, co = np - kw * Co;
    dst[e] = dpre[co * pitch + (x - kw + half) + half];      // (the margins: zeros outside the row, the neighbours' values inside it)
  }
  if (!dbp) return;                                    // (uniform)
  // bias-gradient partial of the (row, chunk): wave `wv` sums channel co over w = lane, lane + 64, ... for co = wv, wv + 4, ...
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int co = wave; co < Co; co += 4) {
    float s = 0.f;
    for (int w = lane; w < wc; w += 64) s += dpre[co * pitch + half + w];
    s = wave_sum(s);
    if (lane == 0) dbp[(size_t)blockIdx.x * Co + co] = s;
  }
}

static inline int tail_bwd_chunks(int W) { return (W + TAIL_BWD_WC - 1) / TAIL_BWD_WC; }

extern "C" int tpgsr_tail_bwd_blocks(int N, int H, int W, int Co, int KS) {
  (void)Co; (void)KS;
  return N * H * tail_bwd_chunks(W);      // one bias-gradient partial row per (image row, chunk)
}

extern "C" int tpgsr_tail_bwd(const float* out_nchw, const float* dout_nchw, int N, int H, int W, int Co, int KS, float* dP,
                              float* dbias_partial, int nblk, void* stream) {
  TPGSR_CHECK_ARG(out_nchw && dout_nchw && dP && N > 0 && H > 0 && W > 0 && Co > 0 && Co <= 8 && (KS & 1) && KS <= 17,
                  "tpgsr_tail_bwd: bad arguments (Co <= 8, odd KS <= 17)");
  const int nchunk = tail_bwd_chunks(W), grid = N * H * nchunk;
  TPGSR_CHECK_ARG(!dbias_partial || nblk == grid, "tpgsr_tail_bwd: dbias_partial needs nblk == %d blocks (got %d)", grid, nblk);
  hipLaunchKernelGGL(tail_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, out_nchw, dout_nchw, H, W, Co, KS,
                     nchunk == 1 ? W : TAIL_BWD_WC, nchunk, dP, dbias_partial);
  TPGSR_LAUNCH_CHECK("tpgsr_tail_bwd");
}

// ------------------------------------------------------------------------------------------------------
// image loss (NCHW): w0*MSE over all C channels + w1*L1(gradmag(out[:, :3]) - gradmag(tgt[:, :3]))
// gradmag(x)[h][w] = sqrt(((x[h][w+1]-x[h][w-1])/2)^2 + ((x[h-1][w]-x[h+1][w])/2)^2 + 1e-6), zero padding
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float at0(const float* p, int h, int w, int H, int W) {
  return ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) ? p[(size_t)h * W + w] : 0.f;
}
__device__ __forceinline__ void gradmag(const float* p, int h, int w, int H, int W, float& gm, float& dx, float& dy) {
  dx = at0(p, h, w + 1, H, W) - at0(p, h, w - 1, H, W);
  dy = at0(p, h - 1, w, H, W) - at0(p, h + 1, w, H, W);
  float a = dx * 0.5f, b = dy * 0.5f;
  gm = sqrtf(a * a + b * b + 1e-6f);
}

__global__ __launch_bounds__(256) void image_loss_fwd_kernel(const float* __restrict__ out, const float* __restrict__ tgt, int N,
                                                             int C, int H, int W, int gradient, float* __restrict__ partial) {
  __shared__ float red[2][4];
  long long total = (long long)N * C * H * W;
  float sq = 0.f, ab = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    float d = out[i] - tgt[i];
    sq += d * d;
    if (gradient) {
      int w = (int)(i % W);
      long long r = i / W;
      int h = (int)(r % H);
      r /= H;
      int c = (int)(r % C);
      if (c < 3) {
        size_t plane = (size_t)r * H * W;
        float g1, g2, t0, t1;
        gradmag(out + plane, h, w, H, W, g1, t0, t1);
        gradmag(tgt + plane, h, w, H, W, g2, t0, t1);
        ab += fabsf(g1 - g2);
      }
    }
  }
  sq = wave_sum(sq);
  ab = wave_sum(ab);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = sq;
    red[1][threadIdx.x >> 6] = ab;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[blockIdx.x * 2] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    partial[blockIdx.x * 2 + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

extern "C" int tpgsr_image_loss_fwd(const float* out, const float* tgt, int N, int C, int H, int W, int gradient, float* partial,
                                    int nblk, void* stream) {
  TPGSR_CHECK_ARG(out && tgt && partial && nblk > 0 && N > 0 && C > 0, "tpgsr_image_loss_fwd: bad arguments");
  hipLaunchKernelGGL(image_loss_fwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, out, tgt, N, C, H, W, gradient,
                     partial);
  TPGSR_LAUNCH_CHECK("tpgsr_image_loss_fwd");
}

__global__ void image_loss_finalize_kernel(const float* __restrict__ partial, int nblk, long long n_mse, long long n_gp, float w0,
                                           float w1, float* loss) {
  double sq = 0.0, ab = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 64) {
    sq += (double)partial[b * 2];
    ab += (double)partial[b * 2 + 1];
  }
  sq = wave_sum_d(sq);
  ab = wave_sum_d(ab);
  if (threadIdx.x == 0) {
    double l = (double)w0 * sq / (double)n_mse;
    if (n_gp > 0) l += (double)w1 * ab / (double)n_gp;
    loss[0] = (float)l;
  }
}

extern "C" int tpgsr_image_loss_finalize(const float* partial, int nblk, long long n_mse, long long n_gp, float w0, float w1,
                                         float* loss, void* stream) {
  TPGSR_CHECK_ARG(partial && loss && nblk > 0 && n_mse > 0, "tpgsr_image_loss_finalize: bad arguments");
  hipLaunchKernelGGL(image_loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, nblk, n_mse, n_gp, w0, w1,
                     loss);
  TPGSR_LAUNCH_CHECK("tpgsr_image_loss_finalize");
}

__global__ __launch_bounds__(256) void image_loss_bwd_kernel(const float* __restrict__ out, const float* __restrict__ tgt,
                                                             const float* __restrict__ dloss, int N, int C, int H, int W,
                                                             int gradient, float cm, float cg, float* __restrict__ dout) {
  long long total = (long long)N * C * H * W;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  float dl = dloss[0];
  float g = cm * 2.f * (out[i] - tgt[i]);
  if (gradient) {
    int w = (int)(i % W);
    long long r = i / W;
    int h = (int)(r % H);
    r /= H;
    int c = (int)(r % C);
    if (c < 3) {
      const float* po = out + (size_t)r * H * W;
      const float* pt = tgt + (size_t)r * H * W;
      float acc = 0.f;
      // out[h][w] is the `r` neighbour of (h, w-1), the `l` neighbour of (h, w+1),
      // the `t` neighbour of (h+1, w) and the `b` neighbour of (h-1, w)
      auto contrib = [&](int qh, int qw, int which) {
        if ((unsigned)qh >= (unsigned)H || (unsigned)qw >= (unsigned)W) return;
        float go, gt, dx, dy, t0, t1;
        gradmag(po, qh, qw, H, W, go, dx, dy);
        gradmag(pt, qh, qw, H, W, gt, t0, t1);
        float diff = go - gt;
        float s = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
        float d = (which == 0) ? dx : (which == 1) ? -dx : (which == 2) ? dy : -dy;
        acc += s * d * 0.25f / go;
      };
      contrib(h, w - 1, 0);
      contrib(h, w + 1, 1);
      contrib(h + 1, w, 2);
      contrib(h - 1, w, 3);
      g += cg * acc;
    }
  }
  dout[i] = dl * g;
}

extern "C" int tpgsr_image_loss_bwd(const float* out, const float* tgt, const float* dloss, int N, int C, int H, int W, int gradient,
                                    float w0, float w1, float* dout, void* stream) {
  TPGSR_CHECK_ARG(out && tgt && dloss && dout, "tpgsr_image_loss_bwd: null pointer");
  long long total = (long long)N * C * H * W;
  long long n_gp = (long long)N * (C < 3 ? C : 3) * H * W;
  float cm = w0 / (float)total, cg = gradient ? w1 / (float)n_gp : 0.f;
  hipLaunchKernelGGL(image_loss_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, out, tgt, dloss, N, C, H, W,
                     gradient, cm, cg, dout);
  TPGSR_LAUNCH_CHECK("tpgsr_image_loss_bwd");
}

// ------------------------------------------------------------------------------------------------------
// optimiser: global L2 norm, clip coefficient, Adam
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, long long n, float* __restrict__ partial) {
  __shared__ double red[4];
  double s = 0.0;
  // 16-byte loads, two in flight per thread (the scalar form read a 14 MB gradient arena in 23 us); x is 16-byte aligned (arena slices are)
  const long long n4 = n >> 2, stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += 2 * stride) {
    const float4 u = *reinterpret_cast<const float4*>(x + i * 4);
    const float4 v = i + stride < n4 ? *reinterpret_cast<const float4*>(x + (i + stride) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    s += ((double)u.x * (double)u.x + (double)u.y * (double)u.y) + ((double)u.z * (double)u.z + (double)u.w * (double)u.w);
    s += ((double)v.x * (double)v.x + (double)v.y * (double)v.y) + ((double)v.z * (double)v.z + (double)v.w * (double)v.w);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float v = x[n4 * 4 + threadIdx.x];
    s += (double)v * (double)v;
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (float)(red[0] + red[1] + red[2] + red[3]);
}

extern "C" int tpgsr_sumsq_partial(const float* x, long long n, float* partial, int nblk, void* stream) {
  TPGSR_CHECK_ARG(x && partial && n > 0 && nblk > 0 && (((uintptr_t)x) & 15) == 0, "tpgsr_sumsq_partial: bad arguments (x must be 16-byte aligned)");
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, n, partial);
  TPGSR_LAUNCH_CHECK("tpgsr_sumsq_partial");
}

__global__ void clip_coef_kernel(const float* __restrict__ partial, int nblk, float max_norm, float* coef, float* norm_out) {
  double s = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 64) s += (double)partial[b];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) {
    float total = (float)sqrt(s);
    float c = max_norm / (total + 1e-6f);
    coef[0] = c < 1.f ? c : 1.f;
    if (norm_out) norm_out[0] = total;
  }
}

extern "C" int tpgsr_clip_coef(const float* partial, int nblk, float max_norm, float* coef, float* norm_out, void* stream) {
  TPGSR_CHECK_ARG(partial && coef && nblk > 0, "tpgsr_clip_coef: bad arguments");
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, nblk, max_norm, coef, norm_out);
  TPGSR_LAUNCH_CHECK("tpgsr_clip_coef");
}

__global__ __launch_bounds__(256) void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, long long n, const float* __restrict__ gscale,
                                                        float lr, float beta1, float beta2, float eps,
                                                        const int* __restrict__ step_dev) {
  __shared__ float s_step_size, s_bc2_sqrt;
  if (threadIdx.x == 0) {
    int t = step_dev[0];
    double bc1 = 1.0 - pow((double)beta1, (double)t);
    double bc2 = 1.0 - pow((double)beta2, (double)t);
    s_step_size = (float)((double)lr / bc1);
    s_bc2_sqrt = (float)sqrt(bc2);
  }
  __syncthreads();
  const float step_size = s_step_size, bc2s = s_bc2_sqrt;
  const float gs = gscale ? gscale[0] : 1.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float gi = g[i] * gs;
    float mi = beta1 * m[i] + (1.f - beta1) * gi;
    float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    float denom = sqrtf(vi) / bc2s + eps;
    p[i] = p[i] - step_size * (mi / denom);
  }
}

extern "C" int tpgsr_adam_step(float* p, const float* g, float* m, float* v, long long n, const float* gscale, float lr,
                               float beta1, float beta2, float eps, const int* step_dev, void* stream) {
  TPGSR_CHECK_ARG(p && g && m && v && step_dev && n > 0, "tpgsr_adam_step: bad arguments");
  int grid = (int)min((long long)4096, (n + 255) / 256);
  hipLaunchKernelGGL(adam_step_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, gscale, lr, beta1, beta2, eps,
                     step_dev);
  TPGSR_LAUNCH_CHECK("tpgsr_adam_step");
}

// clip coefficient of ONE module + the step counters of up to eight modules in one single-wave launch: the optimiser runs on the step's
// critical tail, after the last gradient, where every launch boundary is exposed (round 6: sumsq, THIS, Adam, Adam instead of
// sumsq, clip_coef, step_inc, Adam, step_inc, Adam)
struct StepPtrs {
  int* s[8];
};
__global__ void clip_coef_steps_kernel(const float* __restrict__ partial, int nblk, float max_norm, float* coef, float* norm_out, StepPtrs sp) {
  if (partial) {
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) s += (double)partial[b];
    s = wave_sum_d(s);
    if (threadIdx.x == 0) {
      float total = (float)sqrt(s);
      float c = max_norm / (total + 1e-6f);
      coef[0] = c < 1.f ? c : 1.f;
      if (norm_out) norm_out[0] = total;
    }
  }
  if (threadIdx.x < 8 && sp.s[threadIdx.x]) sp.s[threadIdx.x][0] += 1;
}
extern "C" int tpgsr_clip_coef_steps(const float* partial, int nblk, float max_norm, float* coef, float* norm_out, int* const* steps, int nsteps,
                                     void* stream) {
  TPGSR_CHECK_ARG((partial == nullptr || (coef && nblk > 0)) && nsteps >= 0 && nsteps <= 8 && (nsteps == 0 || steps),
                  "tpgsr_clip_coef_steps: needs coef with partial, and at most eight step counters (HOST array of device pointers)");
  StepPtrs sp;
  for (int i = 0; i < 8; ++i) sp.s[i] = i < nsteps ? steps[i] : nullptr;
  for (int i = 0; i < nsteps; ++i)
    for (int j = 0; j < i; ++j) TPGSR_CHECK_ARG(sp.s[i] != sp.s[j] && sp.s[i], "tpgsr_clip_coef_steps: step counters must be distinct and non-null");
  hipLaunchKernelGGL(clip_coef_steps_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, nblk, max_norm, coef, norm_out, sp);
  TPGSR_LAUNCH_CHECK("tpgsr_clip_coef_steps");
}

__global__ void step_inc_kernel(int* s) { s[0] += 1; }
extern "C" int tpgsr_step_inc(int* step_dev, void* stream) {
  TPGSR_CHECK_ARG(step_dev, "tpgsr_step_inc: null pointer");
  hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev);
  TPGSR_LAUNCH_CHECK("tpgsr_step_inc");
}

__global__ __launch_bounds__(256) void scale_kernel(float* x, long long n, const float* coef) {
  float c = coef[0];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) x[i] *= c;
}
extern "C" int tpgsr_scale_(float* x, long long n, const float* coef, void* stream) {
  TPGSR_CHECK_ARG(x && coef && n > 0, "tpgsr_scale_: bad arguments");
  int grid = (int)min((long long)4096, (n + 255) / 256);
  hipLaunchKernelGGL(scale_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, n, coef);
  TPGSR_LAUNCH_CHECK("tpgsr_scale_");
}
