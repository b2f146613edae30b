// Consumer-side BatchNorm finalize ("derive", round 5).
//
// Train-mode BatchNorm (model/tsrn.py:376,380: conv -> bn -> mish -> conv -> bn; model/stn_head.py:15; model/crnn/crnn.py:47) needs a
// reduction over the whole batch between the convolution that produces its input and whatever consumes its output.  The producing
// convolution's epilogue already leaves per-row-block partial sums behind (tpgsr_conv_args.bn_partial: [nrows][2][C]); until round 4 a
// separate launch (tpgsr_bn_finalize / tpgsr_bn_bwd_finalize: 3 us of work behind a ~5 us launch boundary, 53 times per C3 step) turned
// them into the folded scale / shift (forward) or the three backward coefficients.  Here the FIRST CONSUMER does it itself: every
// workgroup of the consuming kernel sums the rows in its prologue -- the rows are L2-resident, the order of the additions is fixed by
// (nrows, C, blockDim) alone, so every workgroup and every run gets the same bits -- keeps the result in LDS for its own use, and
// workgroup 0 additionally publishes it (scale / shift / saved statistics / running statistics, or dgamma / dbeta / coef) for the
// launches that follow.  No finalize launch, no launch boundary, no grid-wide synchronisation.
//
// Layout of the sum: the [2][C] floats of a row are C/2 float4 column quads; thread (rl, q) = (tid / Q, tid % Q), Q = C / 2, sums rows
// rl, rl + RL, ... (RL = blockDim / Q) of quad q in fp64 with BND_U loads in flight, the RL lane sums of a column are added in lane order.
// Needs Q to divide blockDim (C a power of two, 8 <= C <= 2 * blockDim).
#pragma once
#include "common.h"

#define BND_U 8

__host__ __device__ inline bool bnd_shape_ok(int C, int nthreads) {
  return C >= 8 && (C & (C - 1)) == 0 && (C >> 1) <= nthreads && nthreads % (C >> 1) == 0;
}

// sums[0][c] = sum over rows of row[0][c], sums[1][c] likewise: written to `sums` (LDS, 2 * C doubles); `scr` = LDS scratch of
// 4 * nthreads doubles.  Called by all threads; ends behind a barrier.
__device__ __forceinline__ void bnd_row_sums(const float* __restrict__ rows, const int nrows, const int C, const int tid, const int nthreads,
                                             double* __restrict__ scr, double* __restrict__ sums) {
  const int Q = C >> 1, RL = nthreads / Q, rl = tid / Q, q = tid - rl * Q;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  const size_t ld = (size_t)2 * C;
  for (int r = rl; r < nrows; r += BND_U * RL) {
    float4 v[BND_U];
#pragma unroll
    for (int u = 0; u < BND_U; ++u) {
      const int rr = r + u * RL;
      v[u] = rr < nrows ? *reinterpret_cast<const float4*>(rows + (size_t)rr * ld + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < BND_U; ++u) {
      a0 += (double)v[u].x;
      a1 += (double)v[u].y;
      a2 += (double)v[u].z;
      a3 += (double)v[u].w;
    }
  }
  double* mine = scr + ((size_t)rl * Q + q) * 4;
  mine[0] = a0;
  mine[1] = a1;
  mine[2] = a2;
  mine[3] = a3;
  __syncthreads();
  for (int col = tid; col < 2 * C; col += nthreads) {
    double s = 0.0;
    for (int l = 0; l < RL; ++l) s += scr[(size_t)l * 2 * C + col];
    sums[col] = s;
  }
  __syncthreads();
}

// forward: folded scale / shift of every channel into LDS (ssc / ssh, C floats each); workgroup 0 publishes them with the saved batch
// statistics and the running-statistics update -- the arithmetic of bn_finalize_kernel (elementwise.hip), value for value.
__device__ __forceinline__ void bnd_forward(const tpgsr_bn_derive& d, const int tid, const int nthreads, const bool publish,
                                            double* __restrict__ scr, double* __restrict__ sums, float* __restrict__ ssc,
                                            float* __restrict__ ssh) {
  bnd_row_sums(d.rows, d.nrows, d.C, tid, nthreads, scr, sums);
  const int C = d.C;
  for (int c = tid; c < C; c += nthreads) {
    const double count = (double)d.count;
    const double mean_raw = sums[c] / count;
    double var = sums[C + c] / count - mean_raw * mean_raw;
    if (var < 0.0) var = 0.0;
    const double mean = mean_raw + (d.bias ? (double)d.bias[c] : 0.0);
    const double rstd = 1.0 / sqrt(var + (double)d.eps);
    const float sc = (float)((double)d.gamma[c] * rstd);
    const float sh = (float)((double)d.beta[c] - mean * (double)d.gamma[c] * rstd);
    ssc[c] = sc;
    ssh[c] = sh;
    if (publish) {
      d.scale[c] = sc;
      d.shift[c] = sh;
      if (d.save_mean) d.save_mean[c] = (float)mean;
      if (d.save_rstd) d.save_rstd[c] = (float)rstd;
      if (d.running_mean) {
        const double unbiased = d.count > 1 ? var * count / (double)(d.count - 1) : var;
        d.running_mean[c] = (float)((1.0 - d.momentum) * (double)d.running_mean[c] + d.momentum * mean);
        d.running_var[c] = (float)((1.0 - d.momentum) * (double)d.running_var[c] + d.momentum * unbiased);
      }
    }
  }
  __syncthreads();
}

// backward: dy = coef0 * dz + coef1 * y + coef2; the three coefficients of every channel into LDS (scoef [3][C]); workgroup 0 adds the
// two sums to dgamma / dbeta and publishes coef -- the arithmetic of bn_bwd_finalize_kernel, value for value.
__device__ __forceinline__ void bnd_backward(const tpgsr_bn_derive& d, const int tid, const int nthreads, const bool publish,
                                             double* __restrict__ scr, double* __restrict__ sums, float* __restrict__ scoef) {
  bnd_row_sums(d.rows, d.nrows, d.C, tid, nthreads, scr, sums);
  const int C = d.C;
  for (int c = tid; c < C; c += nthreads) {
    const double s = sums[c], sx = sums[C + c];
    const double rstd = d.save_rstd[c], mu = d.save_mean[c], g = d.gamma[c];
    const double mdz = s / (double)d.count, mdzx = sx / (double)d.count;
    const double c0 = g * rstd;
    const float k0 = (float)c0, k1 = (float)(-c0 * mdzx * rstd), k2 = (float)(-c0 * (mdz - mu * rstd * mdzx));
    scoef[c] = k0;
    scoef[C + c] = k1;
    scoef[2 * C + c] = k2;
    if (publish) {
      if (d.dgamma) d.dgamma[c] = d.accumulate ? d.dgamma[c] + (float)sx : (float)sx;
      if (d.dbeta) d.dbeta[c] = d.accumulate ? d.dbeta[c] + (float)s : (float)s;
      if (d.coef) {
        d.coef[c] = k0;
        d.coef[C + c] = k1;
        d.coef[2 * C + c] = k2;
      }
    }
  }
  __syncthreads();
}
