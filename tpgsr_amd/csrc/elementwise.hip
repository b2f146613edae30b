// HBM-bound elementwise / per-channel reduction kernels: train-mode BatchNorm (statistics, backward),
// activation + max-pool materialisation (STN head), PReLU, layout transposes, small reductions.
// All tensors are [M][C] row-major (NHWC flattened); channel loops are float4-vectorised when C % 4 == 0.
#include "common.h"
#include "bn_derive.h"
#include <type_traits>

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// ------------------------------------------------------------------------------------------------------
// BatchNorm statistics
// ------------------------------------------------------------------------------------------------------
// partial[b][0][c] = sum_m x, partial[b][1][c] = sum_m x^2 over the rows of block b
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, long long M, int C, int ld,
                                                       float* __restrict__ partial, int nblk) {
  long long rows_per = (M + nblk - 1) / nblk;
  long long r0 = blockIdx.x * rows_per, r1 = min(M, r0 + rows_per);
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f, ss = 0.f;
    for (long long r = r0; r < r1; ++r) {
      float v = x[r * ld + c];
      s += v;
      ss += v * v;
    }
    partial[((size_t)blockIdx.x * 2 + 0) * C + c] = s;
    partial[((size_t)blockIdx.x * 2 + 1) * C + c] = ss;
  }
}

extern "C" int tpgsr_bn_stats(const float* x, long long M, int C, int ld, float* partial, int nblk, void* stream) {
  TPGSR_CHECK_ARG(x && partial && M > 0 && C > 0 && nblk > 0, "tpgsr_bn_stats: bad arguments");
  hipLaunchKernelGGL(bn_stats_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, M, C, ld, partial, nblk);
  TPGSR_LAUNCH_CHECK("tpgsr_bn_stats");
}

// per-channel sum of the two partial rows [nblk][2][C] in fp64: one workgroup per channel, 256 row lanes (every lane issues
// its <= nblk/256 loads back to back), wave shuffle + 4-entry LDS combine in a fixed order.  (16 channels x 16 row
// slices per workgroup = 4 workgroups for C = 64 took 12-16 us: each thread walked 48 partial rows serially.)
__device__ __forceinline__ void channel_sums(const float* __restrict__ partial, int nblk, int C, int c, double& s, double& ss) {
  __shared__ double red[2][4];
  double a = 0.0, b2 = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) {
    a += (double)partial[((size_t)b * 2 + 0) * C + c];
    b2 += (double)partial[((size_t)b * 2 + 1) * C + c];
  }
  a = wave_sum_d(a);
  b2 = wave_sum_d(b2);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = a;
    red[1][threadIdx.x >> 6] = b2;
  }
  __syncthreads();
  s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
  ss = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partial, int nblk, int C,
                                                          long long count, const float* conv_bias,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* running_mean, float* running_var, float momentum,
                                                          float eps, int eval, float* scale, float* shift,
                                                          float* save_mean, float* save_rstd) {
  const int c = blockIdx.x;
  if (eval) {
    if (threadIdx.x == 0) {
      float rstd = 1.f / sqrtf(running_var[c] + eps);
      float sc = gamma[c] * rstd;
      scale[c] = sc;
      shift[c] = beta[c] - running_mean[c] * sc;
    }
    return;
  }
  double s, ss;
  channel_sums(partial, nblk, C, c, s, ss);
  if (threadIdx.x == 0) {
    double mean_raw = s / (double)count;
    double var = ss / (double)count - mean_raw * mean_raw;
    if (var < 0.0) var = 0.0;
    double mean = mean_raw + (conv_bias ? (double)conv_bias[c] : 0.0);
    double rstd = 1.0 / sqrt(var + (double)eps);
    float sc = (float)((double)gamma[c] * rstd);
    scale[c] = sc;
    shift[c] = (float)((double)beta[c] - mean * (double)gamma[c] * rstd);
    if (save_mean) save_mean[c] = (float)mean;
    if (save_rstd) save_rstd[c] = (float)rstd;
    if (running_mean) {
      double unbiased = count > 1 ? var * (double)count / (double)(count - 1) : var;
      running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
      running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
    }
  }
}

extern "C" int tpgsr_bn_finalize(const float* partial, int nblk, int C, long long count, const float* conv_bias,
                                 const float* gamma, const float* beta, float* running_mean, float* running_var,
                                 float momentum, float eps, int eval, float* scale, float* shift, float* save_mean,
                                 float* save_rstd, void* stream) {
  TPGSR_CHECK_ARG(gamma && beta && scale && shift && C > 0, "tpgsr_bn_finalize: null pointer");
  TPGSR_CHECK_ARG(eval || (partial && nblk > 0 && count > 0), "tpgsr_bn_finalize: training mode needs partial statistics");
  TPGSR_CHECK_ARG(!eval || (running_mean && running_var), "tpgsr_bn_finalize: eval mode needs running statistics");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, partial, nblk, C, count,
                     conv_bias, gamma, beta, running_mean, running_var, momentum, eps, eval, scale, shift, save_mean,
                     save_rstd);
  TPGSR_LAUNCH_CHECK("tpgsr_bn_finalize");
}

// ------------------------------------------------------------------------------------------------------
// BatchNorm (+activation) backward
//   z = scale*y + shift ; a = act(z) ; given da (+da2): dz = da*act'(z)
//   pass 1: partial[b][0][c] = sum dz ; partial[b][1][c] = sum dz*xhat, xhat = (y-mean)*rstd
//   pass 2: dy = coef0*dz + coef1*y + coef2
// thread layout: C4 = C/4 channel quads across threads, 256/C4 row lanes; requires C%4==0, 256%(C/4)==0
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ da, const float* __restrict__ da2,
                                                            const float* __restrict__ y, long long M, int C,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ save_mean,
                                                            const float* __restrict__ save_rstd, int act,
                                                            float* __restrict__ partial, int nblk) {
  extern __shared__ float sm[];  // [rowlanes][2][C]
  const int C4 = C >> 2;
  const int rl = 256 / C4;
  const int q = threadIdx.x % C4, lane_r = threadIdx.x / C4;
  const int c = q * 4;
  long long rows_per = (M + nblk - 1) / nblk;
  long long r0 = blockIdx.x * rows_per, r1 = min(M, r0 + rows_per);
  float4 sc = ld4(scale + c), sh = ld4(shift + c), mu = ld4(save_mean + c), rs = ld4(save_rstd + c);
  float4 s = make_float4(0, 0, 0, 0), sx = make_float4(0, 0, 0, 0);
  // four rows' loads in flight per thread (the loop is latency-bound: a block owns ~64 rows, 4 per thread); the sums run in the
  // same order as a row-at-a-time loop, rows past the end contribute +0.  Two instances (with / without the second gradient
  // operand) so that no branch -- and with it a full s_waitcnt -- sits between the loads.
  constexpr int U = 4;
  auto sweep = [&](auto has2_tag) __attribute__((always_inline)) {
    constexpr bool HAS2 = decltype(has2_tag)::value;
    for (long long r = r0 + lane_r; r < r1; r += (long long)U * rl) {
      float4 g[U], g2[HAS2 ? U : 1], yv[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long rr = r + (long long)u * rl;
        ok[u] = rr < r1;
        const long long o = (ok[u] ? rr : r) * C + c;
        g[u] = ld4(da + o);
        if (HAS2) g2[u] = ld4(da2 + o);
        yv[u] = ld4(y + o);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (HAS2) {
          g[u].x += g2[u].x; g[u].y += g2[u].y; g[u].z += g2[u].z; g[u].w += g2[u].w;
        }
        if (!ok[u]) g[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (act) {
          g[u].x *= act_grad(yv[u].x * sc.x + sh.x, act);
          g[u].y *= act_grad(yv[u].y * sc.y + sh.y, act);
          g[u].z *= act_grad(yv[u].z * sc.z + sh.z, act);
          g[u].w *= act_grad(yv[u].w * sc.w + sh.w, act);
        }
        s.x += g[u].x; s.y += g[u].y; s.z += g[u].z; s.w += g[u].w;
        sx.x += g[u].x * (yv[u].x - mu.x) * rs.x;
        sx.y += g[u].y * (yv[u].y - mu.y) * rs.y;
        sx.z += g[u].z * (yv[u].z - mu.z) * rs.z;
        sx.w += g[u].w * (yv[u].w - mu.w) * rs.w;
      }
    }
  };
  if (da2) sweep(std::true_type{}); else sweep(std::false_type{});
  float* a0 = sm + ((size_t)lane_r * 2 + 0) * C + c;
  float* a1 = sm + ((size_t)lane_r * 2 + 1) * C + c;
  *reinterpret_cast<float4*>(a0) = s;
  *reinterpret_cast<float4*>(a1) = sx;
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    float t = 0.f;
    for (int l = 0; l < rl; ++l) t += sm[(size_t)l * 2 * C + i];
    partial[(size_t)blockIdx.x * 2 * C + i] = t;
  }
}

static int bn_vec_ok(int C) { return (C & 3) == 0 && C >= 4 && C <= 1024 && (256 % (C >> 2)) == 0; }

extern "C" int tpgsr_bn_bwd_reduce(const float* da, const float* da2, const float* y, long long M, int C,
                                   const float* scale, const float* shift, const float* save_mean,
                                   const float* save_rstd, int act, float* partial, int nblk, void* stream) {
  TPGSR_CHECK_ARG(da && y && scale && shift && save_mean && save_rstd && partial && nblk > 0, "tpgsr_bn_bwd_reduce: null pointer");
  TPGSR_CHECK_ARG(bn_vec_ok(C), "tpgsr_bn_bwd_reduce: unsupported channel count %d", C);
  size_t smem = (size_t)(256 / (C >> 2)) * 2 * C * sizeof(float);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nblk), dim3(256), smem, (hipStream_t)stream, da, da2, y, M, C, scale, shift,
                     save_mean, save_rstd, act, partial, nblk);
  TPGSR_LAUNCH_CHECK("tpgsr_bn_bwd_reduce");
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, int C,
                                                              long long count, const float* __restrict__ gamma,
                                                              const float* __restrict__ save_mean,
                                                              const float* __restrict__ save_rstd, float* dgamma,
                                                              float* dbeta, int accumulate, float* coef) {
  const int c = blockIdx.x;
  double s, sx;
  channel_sums(partial, nblk, C, c, s, sx);
  if (threadIdx.x == 0) {
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)sx : (float)sx;
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)s : (float)s;
    double rstd = save_rstd[c], mu = save_mean[c], g = gamma[c];
    double mdz = s / (double)count, mdzx = sx / (double)count;
    double c0 = g * rstd;
    coef[c] = (float)c0;
    coef[C + c] = (float)(-c0 * mdzx * rstd);
    coef[2 * C + c] = (float)(-c0 * (mdz - mu * rstd * mdzx));
  }
}

extern "C" int tpgsr_bn_bwd_finalize(const float* partial, int nblk, int C, long long count, const float* gamma,
                                     const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta,
                                     int accumulate, float* coef, void* stream) {
  TPGSR_CHECK_ARG(partial && gamma && save_mean && save_rstd && coef && nblk > 0 && count > 0, "tpgsr_bn_bwd_finalize: bad arguments");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, partial, nblk, C, count,
                     gamma, save_mean, save_rstd, dgamma, dbeta, accumulate, coef);
  TPGSR_LAUNCH_CHECK("tpgsr_bn_bwd_finalize");
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ da, const float* __restrict__ da2,
                                                           const float* __restrict__ y, long long total4, int C,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           int act, const float* __restrict__ coef, float* __restrict__ dy) {
  const int C4 = C >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    int c = (int)(i % C4) * 4;
    float4 g = ld4(da + i * 4);
    if (da2) {
      float4 g2 = ld4(da2 + i * 4);
      g.x += g2.x; g.y += g2.y; g.z += g2.z; g.w += g2.w;
    }
    float4 yv = ld4(y + i * 4);
    if (act) {
      float4 sc = ld4(scale + c), sh = ld4(shift + c);
      g.x *= act_grad(yv.x * sc.x + sh.x, act);
      g.y *= act_grad(yv.y * sc.y + sh.y, act);
      g.z *= act_grad(yv.z * sc.z + sh.z, act);
      g.w *= act_grad(yv.w * sc.w + sh.w, act);
    }
    float4 c0 = ld4(coef + c), c1 = ld4(coef + C + c), c2 = ld4(coef + 2 * C + c);
    float4 o;
    o.x = c0.x * g.x + c1.x * yv.x + c2.x;
    o.y = c0.y * g.y + c1.y * yv.y + c2.y;
    o.z = c0.z * g.z + c1.z * yv.z + c2.z;
    o.w = c0.w * g.w + c1.w * yv.w + c2.w;
    *reinterpret_cast<float4*>(dy + i * 4) = o;
  }
}

extern "C" int tpgsr_bn_bwd_apply(const float* da, const float* da2, const float* y, long long M, int C, const float* scale,
                                  const float* shift, int act, const float* coef, float* dy, void* stream) {
  TPGSR_CHECK_ARG(da && y && coef && dy && (C & 3) == 0, "tpgsr_bn_bwd_apply: bad arguments");
  TPGSR_CHECK_ARG(!act || (scale && shift), "tpgsr_bn_bwd_apply: activation needs scale/shift");
  long long total4 = M * C / 4;
  int grid = (int)min((long long)4096, (total4 + 255) / 256);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, da, da2, y, total4, C, scale, shift,
                     act, coef, dy);
  TPGSR_LAUNCH_CHECK("tpgsr_bn_bwd_apply");
}

// out = act(scale[c]*x + shift[c]) materialised once (float4-vectorised): used where the consumer would otherwise
// re-apply an expensive activation per filter tap (mish in front of a 3x3 / 9x1 conv)
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ x, long long total4, int C,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         int act, float* __restrict__ out) {
  const int C4 = C >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    float4 v = ld4(x + i * 4);
    if (scale) {
      int c = (int)(i % C4) * 4;
      float4 s = ld4(scale + c), t = ld4(shift + c);
      v.x = v.x * s.x + t.x; v.y = v.y * s.y + t.y; v.z = v.z * s.z + t.z; v.w = v.w * s.w + t.w;
    }
    v.x = apply_act(v.x, act); v.y = apply_act(v.y, act); v.z = apply_act(v.z, act); v.w = apply_act(v.w, act);
    *reinterpret_cast<float4*>(out + i * 4) = v;
  }
}

__global__ void affine_act_pool_kernel(const float* __restrict__ x, int N, int H, int W, int C, const float* __restrict__ scale,
                                       const float* __restrict__ shift, int act, int ph, int pw, float* __restrict__ out);

extern "C" int tpgsr_affine_act(const float* x, long long M, int C, const float* scale, const float* shift, int act, float* out,
                                void* stream) {
  TPGSR_CHECK_ARG(x && out && M > 0 && C > 0, "tpgsr_affine_act: bad arguments");
  TPGSR_CHECK_ARG((scale == nullptr) == (shift == nullptr), "tpgsr_affine_act: scale/shift must come together");
  if (C & 3) {   // channel counts that are no multiple of 4 (the one-channel offset map of MORAN's rectifier): the scalar 1x1 "pool"
    TPGSR_CHECK_ARG(M < (1ll << 31), "tpgsr_affine_act: M too large for the scalar path");
    const long long total = M * C;
    const int grid = (int)min((long long)4096, (total + 255) / 256);
    hipLaunchKernelGGL(affine_act_pool_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, 1, 1, (int)M, C, scale, shift, act, 1, 1, out);
    TPGSR_LAUNCH_CHECK("tpgsr_affine_act");
  }
  long long total4 = M * C / 4;
  int grid = (int)min((long long)8192, (total4 + 255) / 256);
  hipLaunchKernelGGL(affine_act_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, total4, C, scale, shift, act, out);
  TPGSR_LAUNCH_CHECK("tpgsr_affine_act");
}

// ---- the same with the BatchNorm finalized by the launch itself (bn_derive.h): the grid's first ceil(C / 16) workgroups derive and
// publish, everybody waits on the flag with its first loads already in flight, scale / shift out of LDS ----
#define BND_LDS_DECL                                              \
  __shared__ double bnd_scr[32 * 32];                             \
  __shared__ double bnd_sums[32];                                 \
  __shared__ __attribute__((aligned(16))) float bnd_a[512], bnd_b[512]

static int bnd_check(const tpgsr_bn_derive* d, const char* who, bool fwd, long long grid) {
  TPGSR_CHECK_ARG(d && d->rows && d->nrows > 0 && d->count > 0 && d->gamma && d->flag, "%s: incomplete BatchNorm descriptor (rows, count, gamma, flag)", who);
  TPGSR_CHECK_ARG(bnd_shape_ok(d->C), "%s: channel count %d not supported by the in-launch finalize (a multiple of 16 up to 512, or 8)", who, d->C);
  TPGSR_CHECK_ARG((((uintptr_t)d->rows) & 15) == 0, "%s: partial rows must be 16-byte aligned", who);
  TPGSR_CHECK_ARG(grid >= bnd_derivers(d->C), "%s: the launch has %lld workgroups, the finalize needs %d derivers", who, grid, bnd_derivers(d->C));
  if (fwd) TPGSR_CHECK_ARG(d->beta && d->scale && d->shift && (!d->running_mean == !d->running_var), "%s: forward descriptor needs beta / scale / shift", who);
  else TPGSR_CHECK_ARG(d->save_mean && d->save_rstd && d->coef, "%s: backward descriptor needs the saved statistics and coef", who);
  return 0;
}

__global__ __launch_bounds__(256) void affine_act_bnd_kernel(const tpgsr_bn_derive d, const float* __restrict__ x, long long total4,
                                                             int act, float* __restrict__ out) {
  BND_LDS_DECL;
  const long long stride = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  float4 v = i < total4 ? ld4(x + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);      // in flight while the BatchNorm is being finalized
  bnd_forward(d, threadIdx.x, bnd_scr, bnd_sums, bnd_a, bnd_b);
  const int C4 = d.C >> 2;
  while (i < total4) {
    const long long nx = i + stride;
    const float4 vn = nx < total4 ? ld4(x + nx * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int c = (int)(i % C4) * 4;
    const float4 s = *reinterpret_cast<const float4*>(bnd_a + c), t = *reinterpret_cast<const float4*>(bnd_b + c);
    v.x = apply_act(v.x * s.x + t.x, act); v.y = apply_act(v.y * s.y + t.y, act);
    v.z = apply_act(v.z * s.z + t.z, act); v.w = apply_act(v.w * s.w + t.w, act);
    *reinterpret_cast<float4*>(out + i * 4) = v;
    v = vn;
    i = nx;
  }
}

extern "C" int tpgsr_affine_act_bnd(const tpgsr_bn_derive* d, const float* x, long long M, int act, float* out, void* stream) {
  TPGSR_CHECK_ARG(d && x && out && M > 0, "tpgsr_affine_act_bnd: bad arguments");
  const long long total4 = M * d->C / 4;
  const int grid = (int)min((long long)8192, (total4 + 255) / 256);
  if (int rc = bnd_check(d, "tpgsr_affine_act_bnd", true, grid)) return rc;
  hipLaunchKernelGGL(affine_act_bnd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *d, x, total4, act, out);
  TPGSR_LAUNCH_CHECK("tpgsr_affine_act_bnd");
}

__global__ __launch_bounds__(256) void bn_bwd_apply_bnd_kernel(const tpgsr_bn_derive d, const float* __restrict__ da,
                                                               const float* __restrict__ da2, const float* __restrict__ y,
                                                               long long total4, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, int act, float* __restrict__ dy) {
  __shared__ double bnd_scr[32 * 32];
  __shared__ double bnd_sums[32];
  __shared__ __attribute__((aligned(16))) float coef[3 * 512];
  const int C = d.C, C4 = C >> 2;
  const long long stride = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const bool has2 = da2 != nullptr;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 g = zero, g2 = zero, yv = zero;
  if (i < total4) {      // in flight while the coefficients are being derived
    g = ld4(da + i * 4);
    yv = ld4(y + i * 4);
    if (has2) g2 = ld4(da2 + i * 4);
  }
  bnd_backward(d, threadIdx.x, bnd_scr, bnd_sums, coef);
  while (i < total4) {
    const long long nx = i + stride;
    float4 gn = zero, g2n = zero, yn = zero;
    if (nx < total4) {
      gn = ld4(da + nx * 4);
      yn = ld4(y + nx * 4);
      if (has2) g2n = ld4(da2 + nx * 4);
    }
    const int c = (int)(i % C4) * 4;
    if (has2) {
      g.x += g2.x; g.y += g2.y; g.z += g2.z; g.w += g2.w;
    }
    if (act) {
      const float4 sc = ld4(scale + c), sh = ld4(shift + c);
      g.x *= act_grad(yv.x * sc.x + sh.x, act);
      g.y *= act_grad(yv.y * sc.y + sh.y, act);
      g.z *= act_grad(yv.z * sc.z + sh.z, act);
      g.w *= act_grad(yv.w * sc.w + sh.w, act);
    }
    const float4 c0 = *reinterpret_cast<const float4*>(coef + c), c1 = *reinterpret_cast<const float4*>(coef + C + c),
                 c2 = *reinterpret_cast<const float4*>(coef + 2 * C + c);
    float4 o;
    o.x = c0.x * g.x + c1.x * yv.x + c2.x;
    o.y = c0.y * g.y + c1.y * yv.y + c2.y;
    o.z = c0.z * g.z + c1.z * yv.z + c2.z;
    o.w = c0.w * g.w + c1.w * yv.w + c2.w;
    *reinterpret_cast<float4*>(dy + i * 4) = o;
    g = gn; g2 = g2n; yv = yn;
    i = nx;
  }
}

extern "C" int tpgsr_bn_bwd_apply_bnd(const tpgsr_bn_derive* d, const float* da, const float* da2, const float* y, long long M,
                                      const float* scale, const float* shift, int act, float* dy, void* stream) {
  TPGSR_CHECK_ARG(d && da && y && dy && M > 0, "tpgsr_bn_bwd_apply_bnd: bad arguments");
  TPGSR_CHECK_ARG(!act || (scale && shift), "tpgsr_bn_bwd_apply_bnd: activation needs scale/shift");
  const long long total4 = M * d->C / 4;
  const int grid = (int)min((long long)4096, (total4 + 255) / 256);
  if (int rc = bnd_check(d, "tpgsr_bn_bwd_apply_bnd", false, grid)) return rc;
  hipLaunchKernelGGL(bn_bwd_apply_bnd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *d, da, da2, y, total4, scale,
                     shift, act, dy);
  TPGSR_LAUNCH_CHECK("tpgsr_bn_bwd_apply_bnd");
}

// ------------------------------------------------------------------------------------------------------
// act(scale*x+shift) + max-pool (ph x pw, stride = window, floor) -- STN head / CRNN pooling stages
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void affine_act_pool_kernel(const float* __restrict__ x, int N, int H, int W, int C,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ shift, int act, int ph, int pw,
                                                              float* __restrict__ out) {
  int OH = H / ph, OW = W / pw;
  long long total = (long long)N * OH * OW * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    long long p = i / C;
    int ow = (int)(p % OW);
    p /= OW;
    int oh = (int)(p % OH);
    int n = (int)(p / OH);
    float sc = scale ? scale[c] : 1.f, sh = shift ? shift[c] : 0.f;
    float best = -INFINITY;
    for (int a = 0; a < ph; ++a)
      for (int b = 0; b < pw; ++b) {
        float v = x[((size_t)(n * H + oh * ph + a) * W + ow * pw + b) * C + c];
        v = apply_act(v * sc + sh, act);
        if (v > best || v != v) best = v;
      }
    out[i] = best;
  }
}

extern "C" int tpgsr_affine_act_pool(const float* x, int N, int H, int W, int C, const float* scale, const float* shift,
                                     int act, int pool_h, int pool_w, float* out, void* stream) {
  TPGSR_CHECK_ARG(x && out && pool_h >= 1 && pool_w >= 1 && H >= pool_h && W >= pool_w, "tpgsr_affine_act_pool: bad arguments");
  long long total = (long long)N * (H / pool_h) * (W / pool_w) * C;
  int grid = (int)min((long long)4096, (total + 255) / 256);
  hipLaunchKernelGGL(affine_act_pool_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, N, H, W, C, scale, shift, act,
                     pool_h, pool_w, out);
  TPGSR_LAUNCH_CHECK("tpgsr_affine_act_pool");
}

__global__ __launch_bounds__(256) void affine_act_pool_bnd_kernel(const tpgsr_bn_derive d, const float* __restrict__ x, int N, int H, int W,
                                                                  int act, int ph, int pw, float* __restrict__ out) {
  BND_LDS_DECL;
  bnd_forward(d, threadIdx.x, bnd_scr, bnd_sums, bnd_a, bnd_b);
  const int C = d.C, OH = H / ph, OW = W / pw;
  const long long total = (long long)N * OH * OW * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long long p = i / C;
    const int ow = (int)(p % OW);
    p /= OW;
    const int oh = (int)(p % OH);
    const int n = (int)(p / OH);
    const float sc = bnd_a[c], sh = bnd_b[c];
    float best = -INFINITY;
    for (int a = 0; a < ph; ++a)
      for (int b = 0; b < pw; ++b) {
        float v = x[((size_t)(n * H + oh * ph + a) * W + ow * pw + b) * C + c];
        v = apply_act(v * sc + sh, act);
        if (v > best || v != v) best = v;
      }
    out[i] = best;
  }
}

extern "C" int tpgsr_affine_act_pool_bnd(const tpgsr_bn_derive* d, const float* x, int N, int H, int W, int act, int pool_h, int pool_w,
                                         float* out, void* stream) {
  TPGSR_CHECK_ARG(d && x && out && pool_h >= 1 && pool_w >= 1 && H >= pool_h && W >= pool_w, "tpgsr_affine_act_pool_bnd: bad arguments");
  const long long total = (long long)N * (H / pool_h) * (W / pool_w) * d->C;
  const int grid = (int)min((long long)4096, (total + 255) / 256);
  if (int rc = bnd_check(d, "tpgsr_affine_act_pool_bnd", true, grid)) return rc;
  hipLaunchKernelGGL(affine_act_pool_bnd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *d, x, N, H, W, act, pool_h,
                     pool_w, out);
  TPGSR_LAUNCH_CHECK("tpgsr_affine_act_pool_bnd");
}

// dz[n][h][w][c] = (h,w is the first arg-max of its window) ? dout * act'(scale*x+shift) : 0
__global__ __launch_bounds__(256) void affine_act_pool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dout,
                                                                  int N, int H, int W, int C,
                                                                  const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, int act, int ph, int pw,
                                                                  float* __restrict__ dz) {
  int OH = H / ph, OW = W / pw;
  long long total = (long long)N * OH * OW * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    long long p = i / C;
    int ow = (int)(p % OW);
    p /= OW;
    int oh = (int)(p % OH);
    int n = (int)(p / OH);
    float sc = scale ? scale[c] : 1.f, sh = shift ? shift[c] : 0.f;
    float best = -INFINITY, bestpre = 0.f;
    int ba = 0, bb = 0;
    for (int a = 0; a < ph; ++a)
      for (int b = 0; b < pw; ++b) {
        float pre = x[((size_t)(n * H + oh * ph + a) * W + ow * pw + b) * C + c] * sc + sh;
        float v = apply_act(pre, act);
        if (v > best || v != v) {
          best = v;
          bestpre = pre;
          ba = a;
          bb = b;
        }
      }
    float g = dout[i] * act_grad(bestpre, act);
    for (int a = 0; a < ph; ++a)
      for (int b = 0; b < pw; ++b)
        dz[((size_t)(n * H + oh * ph + a) * W + ow * pw + b) * C + c] = (a == ba && b == bb) ? g : 0.f;
  }
  // rows/cols dropped by the floor (H % ph, W % pw) receive no gradient
  if (H % ph || W % pw) {
    long long tot2 = (long long)N * H * W * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < tot2; i += (long long)gridDim.x * blockDim.x) {
      long long p = i / C;
      int w = (int)(p % W);
      int h = (int)((p / W) % H);
      if (h >= OH * ph || w >= OW * pw) dz[i] = 0.f;
    }
  }
}

extern "C" int tpgsr_affine_act_pool_bwd(const float* x, const float* dout, int N, int H, int W, int C, const float* scale,
                                         const float* shift, int act, int pool_h, int pool_w, float* dz, void* stream) {
  TPGSR_CHECK_ARG(x && dout && dz && pool_h >= 1 && pool_w >= 1, "tpgsr_affine_act_pool_bwd: bad arguments");
  long long total = (long long)N * (H / pool_h) * (W / pool_w) * C;
  int grid = (int)min((long long)4096, (total + 255) / 256);
  hipLaunchKernelGGL(affine_act_pool_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, dout, N, H, W, C, scale,
                     shift, act, pool_h, pool_w, dz);
  TPGSR_LAUNCH_CHECK("tpgsr_affine_act_pool_bwd");
}

// ------------------------------------------------------------------------------------------------------
// text-prior strip: bilinear (align_corners=True) resample along W of act(scale*in+shift), and its backward
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void strip_src(int ow, int Win, int Wout, int& x0, int& x1, float& l) {
  float sx = Wout > 1 ? (float)ow * ((float)(Win - 1) / (float)(Wout - 1)) : 0.f;
  x0 = (int)sx;
  if (x0 > Win - 1) x0 = Win - 1;
  x1 = x0 + 1 < Win ? x0 + 1 : x0;
  l = sx - (float)x0;
}

__global__ __launch_bounds__(256) void strip_resample_fwd_kernel(const float* __restrict__ in, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, int act, int N, int Win,
                                                                 int Wout, int C, float* __restrict__ out) {
  long long total = (long long)N * Wout * C;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int c = (int)(i % C);
  long long r = i / C;
  int ow = (int)(r % Wout);
  int n = (int)(r / Wout);
  int x0, x1;
  float l;
  strip_src(ow, Win, Wout, x0, x1, l);
  float sc = scale ? scale[c] : 1.f, sh = shift ? shift[c] : 0.f;
  float a = apply_act(in[((size_t)n * Win + x0) * C + c] * sc + sh, act);
  float b = apply_act(in[((size_t)n * Win + x1) * C + c] * sc + sh, act);
  out[i] = a * (1.f - l) + b * l;
}

extern "C" int tpgsr_strip_resample_fwd(const float* in, const float* scale, const float* shift, int act, int N, int Win, int Wout,
                                        int C, float* out, void* stream) {
  TPGSR_CHECK_ARG(in && out && N > 0 && Win > 0 && Wout > 0 && C > 0, "tpgsr_strip_resample_fwd: bad arguments");
  long long total = (long long)N * Wout * C;
  hipLaunchKernelGGL(strip_resample_fwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, in, scale, shift, act, N,
                     Win, Wout, C, out);
  TPGSR_LAUNCH_CHECK("tpgsr_strip_resample_fwd");
}

// gather form (deterministic): every input column sums the output columns that sampled it
__global__ __launch_bounds__(256) void strip_resample_bwd_kernel(const float* __restrict__ in, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, int act,
                                                                 const float* __restrict__ dout, int N, int Win, int Wout, int C,
                                                                 float* __restrict__ dz) {
  long long total = (long long)N * Win * C;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int c = (int)(i % C);
  long long r = i / C;
  int iw = (int)(r % Win);
  int n = (int)(r / Win);
  float g = 0.f;
  // only outputs whose two source columns include iw contribute: sx(ow) = ow (Win-1)/(Wout-1) in [iw - 1, iw + 1).  The loop visits that
  // window (+ one output either side against rounding) in ascending ow with the forward pass's own strip_src -- the same addends in the
  // same order as the full 0 .. Wout-1 sweep it replaces (27 us for a 1 MB tensor: Wout dependent loads and divisions per element)
  int lo = 0, hi = Wout - 1;
  if (Win > 1 && Wout > 1) {
    const float inv = (float)(Wout - 1) / (float)(Win - 1);
    lo = max(0, (int)floorf((float)(iw - 1) * inv) - 1);
    hi = min(Wout - 1, (int)ceilf((float)(iw + 1) * inv) + 1);
  }
  for (int ow = lo; ow <= hi; ++ow) {
    int x0, x1;
    float l;
    strip_src(ow, Win, Wout, x0, x1, l);
    float d = dout[((size_t)n * Wout + ow) * C + c];
    if (x0 == iw) g += d * (1.f - l);
    if (x1 == iw) g += d * l;
  }
  float sc = scale ? scale[c] : 1.f, sh = shift ? shift[c] : 0.f;
  dz[i] = g * act_grad(in[i] * sc + sh, act);
}

extern "C" int tpgsr_strip_resample_bwd(const float* in, const float* scale, const float* shift, int act, const float* dout, int N,
                                        int Win, int Wout, int C, float* dz, void* stream) {
  TPGSR_CHECK_ARG(in && dout && dz && N > 0 && Win > 0 && Wout > 0 && C > 0, "tpgsr_strip_resample_bwd: bad arguments");
  long long total = (long long)N * Win * C;
  hipLaunchKernelGGL(strip_resample_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, in, scale, shift, act,
                     dout, N, Win, Wout, C, dz);
  TPGSR_LAUNCH_CHECK("tpgsr_strip_resample_bwd");
}

// dstrip[n][w][c] (+)= sum_h d[n][h][w][c]: one thread per float4 of a strip row, the H loads issued together (the sum runs h = 0 .. H-1)
__global__ __launch_bounds__(256) void hsum_kernel(const float* __restrict__ d, int N, int H, int WC4, float* dstrip, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * WC4) return;
  const int n = i / WC4, wc = i - n * WC4;
  const float4* src = reinterpret_cast<const float4*>(d) + (size_t)n * H * WC4 + wc;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  int h = 0;
  for (; h + 8 <= H; h += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(h + u) * WC4];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w;
    }
  }
  for (; h < H; ++h) {
    const float4 v = src[(size_t)h * WC4];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  float4* dst = reinterpret_cast<float4*>(dstrip) + i;
  if (accumulate) {
    const float4 o = *dst;
    s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
  }
  *dst = s;
}

extern "C" int tpgsr_hsum(const float* d, int N, int H, int W, int C, float* dstrip, int accumulate, void* stream) {
  TPGSR_CHECK_ARG(d && dstrip && N > 0 && H > 0 && W > 0 && C > 0, "tpgsr_hsum: bad arguments");
  TPGSR_CHECK_ARG(((W * C) & 3) == 0 && ((((uintptr_t)d) | ((uintptr_t)dstrip)) & 15) == 0 && (long long)N * W * C < (1ll << 31),
                  "tpgsr_hsum: needs W C %% 4 == 0, 16-byte aligned operands and fewer than 2^31 strip elements");
  const int WC4 = (W * C) >> 2;
  hipLaunchKernelGGL(hsum_kernel, dim3(cdiv((long long)N * WC4, 256)), dim3(256), 0, (hipStream_t)stream, d, N, H, WC4, dstrip, accumulate);
  TPGSR_LAUNCH_CHECK("tpgsr_hsum");
}

// ------------------------------------------------------------------------------------------------------
// PReLU (single shared slope), add, activation backward, transposes, small reductions
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prelu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ alpha,
                                                        long long n4, float* __restrict__ y) {
  float a = alpha[0];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 v = ld4(x + i * 4);
    v.x = v.x > 0.f ? v.x : a * v.x;
    v.y = v.y > 0.f ? v.y : a * v.y;
    v.z = v.z > 0.f ? v.z : a * v.z;
    v.w = v.w > 0.f ? v.w : a * v.w;
    *reinterpret_cast<float4*>(y + i * 4) = v;
  }
}

extern "C" int tpgsr_prelu_fwd(const float* x, const float* alpha, long long n, float* y, void* stream) {
  TPGSR_CHECK_ARG(x && alpha && y && (n & 3) == 0, "tpgsr_prelu_fwd: bad arguments (n must be a multiple of 4)");
  int grid = (int)min((long long)4096, (n / 4 + 255) / 256);
  hipLaunchKernelGGL(prelu_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, alpha, n / 4, y);
  TPGSR_LAUNCH_CHECK("tpgsr_prelu_fwd");
}

__global__ __launch_bounds__(256) void prelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ alpha,
                                                        const float* __restrict__ dy, const float* __restrict__ dy2,
                                                        long long n4, float* __restrict__ dx, float* __restrict__ dap) {
  __shared__ double red[4];
  float a = alpha[0];
  double acc = 0.0;      // d alpha = sum over the negative side of dy * x: a heavily cancelling sum, accumulated in fp64
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 v = ld4(x + i * 4);
    float4 g = ld4(dy + i * 4);
    if (dy2) {
      float4 g2 = ld4(dy2 + i * 4);
      g.x += g2.x; g.y += g2.y; g.z += g2.z; g.w += g2.w;
    }
    float4 o;
    o.x = v.x > 0.f ? g.x : a * g.x;
    o.y = v.y > 0.f ? g.y : a * g.y;
    o.z = v.z > 0.f ? g.z : a * g.z;
    o.w = v.w > 0.f ? g.w : a * g.w;
    acc += (double)((v.x > 0.f ? 0.f : g.x * v.x) + (v.y > 0.f ? 0.f : g.y * v.y)) +
           (double)((v.z > 0.f ? 0.f : g.z * v.z) + (v.w > 0.f ? 0.f : g.w * v.w));
    *reinterpret_cast<float4*>(dx + i * 4) = o;
  }
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) dap[blockIdx.x] = (float)((red[0] + red[1]) + (red[2] + red[3]));
}

extern "C" int tpgsr_prelu_bwd(const float* x, const float* alpha, const float* dy, const float* dy2, long long n, float* dx,
                               float* dalpha_partial, int nblk, void* stream) {
  TPGSR_CHECK_ARG(x && alpha && dy && dx && dalpha_partial && nblk > 0 && (n & 3) == 0, "tpgsr_prelu_bwd: bad arguments");
  hipLaunchKernelGGL(prelu_bwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, alpha, dy, dy2, n / 4, dx,
                     dalpha_partial);
  TPGSR_LAUNCH_CHECK("tpgsr_prelu_bwd");
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n4,
                                                  long long n, float* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 u = ld4(a + i * 4), v = ld4(b + i * 4);
    u.x += v.x; u.y += v.y; u.z += v.z; u.w += v.w;
    *reinterpret_cast<float4*>(out + i * 4) = u;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    long long i = n4 * 4 + threadIdx.x;
    out[i] = a[i] + b[i];
  }
}

extern "C" int tpgsr_add(const float* a, const float* b, long long n, float* out, void* stream) {
  TPGSR_CHECK_ARG(a && b && out && n > 0, "tpgsr_add: bad arguments");
  int grid = (int)max((long long)1, min((long long)4096, (n / 4 + 255) / 256));
  hipLaunchKernelGGL(add_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, b, n / 4, n, out);
  TPGSR_LAUNCH_CHECK("tpgsr_add");
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const float* x, const float* dy, long long n, int act, float* dx) {
  long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 xv = ld4(x + i * 4), g = ld4(dy + i * 4);
    g.x *= act_grad(xv.x, act); g.y *= act_grad(xv.y, act); g.z *= act_grad(xv.z, act); g.w *= act_grad(xv.w, act);
    *reinterpret_cast<float4*>(dx + i * 4) = g;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    long long i = n4 * 4 + threadIdx.x;
    dx[i] = dy[i] * act_grad(x[i], act);
  }
}

extern "C" int tpgsr_act_bwd(const float* x, const float* dy, long long n, int act, float* dx, void* stream) {
  TPGSR_CHECK_ARG(x && dy && dx && n > 0, "tpgsr_act_bwd: bad arguments");
  int grid = (int)max((long long)1, min((long long)8192, (n / 4 + 255) / 256));
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, dy, n, act, dx);
  TPGSR_LAUNCH_CHECK("tpgsr_act_bwd");
}

__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ in, int N, int C, int HW,
                                                           float* __restrict__ out) {
  long long total = (long long)N * C * HW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    long long p = i / C;
    int hw = (int)(p % HW);
    int n = (int)(p / HW);
    out[i] = in[((size_t)n * C + c) * HW + hw];
  }
}
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ in, int N, int C, int HW,
                                                           float* __restrict__ out) {
  long long total = (long long)N * C * HW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int hw = (int)(i % HW);
    long long p = i / HW;
    int c = (int)(p % C);
    int n = (int)(p / C);
    out[i] = in[((size_t)n * HW + hw) * C + c];
  }
}

extern "C" int tpgsr_nchw_to_nhwc(const float* in, int N, int C, int H, int W, float* out, void* stream) {
  TPGSR_CHECK_ARG(in && out && N > 0 && C > 0 && H > 0 && W > 0, "tpgsr_nchw_to_nhwc: bad arguments");
  long long total = (long long)N * C * H * W;
  int grid = (int)min((long long)8192, (total + 255) / 256);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, N, C, H * W, out);
  TPGSR_LAUNCH_CHECK("tpgsr_nchw_to_nhwc");
}
extern "C" int tpgsr_nhwc_to_nchw(const float* in, int N, int C, int H, int W, float* out, void* stream) {
  TPGSR_CHECK_ARG(in && out && N > 0 && C > 0 && H > 0 && W > 0, "tpgsr_nhwc_to_nchw: bad arguments");
  long long total = (long long)N * C * H * W;
  int grid = (int)min((long long)8192, (total + 255) / 256);
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, N, C, H * W, out);
  TPGSR_LAUNCH_CHECK("tpgsr_nhwc_to_nchw");
}

// one block per output: 256 lanes stride over the Z partials, fixed-order tree combine (deterministic)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, int Z, int n, float* out,
                                                              int accumulate) {
  __shared__ double red[4];
  const int i = blockIdx.x;
  double s = 0.0;
  for (int z = threadIdx.x; z < Z; z += 256) s += (double)part[(size_t)z * n + i];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float v = (float)((red[0] + red[1]) + (red[2] + red[3]));
    out[i] = accumulate ? out[i] + v : v;
  }
}

extern "C" int tpgsr_reduce_partials(const float* part, int Z, int n, float* out, int accumulate, void* stream) {
  TPGSR_CHECK_ARG(part && out && Z > 0 && n > 0, "tpgsr_reduce_partials: bad arguments");
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, part, Z, n, out, accumulate);
  TPGSR_LAUNCH_CHECK("tpgsr_reduce_partials");
}
