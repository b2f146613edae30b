// BatchNorm finalize inside the launch that consumes it ("derive", round 5).
//
// Train-mode BatchNorm (model/tsrn.py:376,380: conv -> bn -> mish -> conv -> bn; model/stn_head.py:15; model/crnn/crnn.py:47) needs a
// reduction over the whole batch between the convolution that produces its input and whatever consumes its output.  The producing
// convolution's epilogue already leaves per-row-block partial sums behind (tpgsr_conv_args.bn_partial: [nrows][2][C]); until round 4 a
// separate launch (tpgsr_bn_finalize / tpgsr_bn_bwd_finalize: 3 us of work behind a launch boundary, 53 times per C3 step) turned them
// into the folded scale / shift (forward) or the three backward coefficients.  Here the FIRST CONSUMER's launch does it:
//   * the first D = ceil(C / 16) workgroups of the grid are DERIVERS: workgroup k sums the rows of channels 16 k .. 16 k + 15 (fp64, an
//     order fixed by (nrows, C) alone), finishes the BatchNorm arithmetic for them, publishes what THIS launch needs (scale / shift, or the
//     backward coefficients) by write-through (`sc1`) stores, waits for their acknowledgement (`s_waitcnt vmcnt(0)`), and arrives on a
//     flag with ONE relaxed agent-scope atomic; what only LATER launches read (saved statistics, running statistics, dgamma / dbeta) goes
//     out as plain stores;
//   * every workgroup -- the derivers included -- issues nothing that depends on the BatchNorm before one lane has seen flag == D
//     (relaxed agent-scope polls, sleeping in between), then reads the published values with `sc1` loads (they bypass this CU's vector L1)
//     into LDS.  This is the "{sc1 stores, sc1 loads}" hand-off of MI355X_MICROARCH.md (inter-workgroup visibility): correct for any
//     placement of the workgroups on XCDs, no fence, no L2 write-back.
// The derivers are the FIRST workgroups of the grid and wait for nobody, so they make progress as soon as they are resident; workgroups
// are dispatched in index order, so no consumer can be resident before them.  Should that ever not hold, a poll that sees nothing for
// ~0.5 s gives up and POISONS its outputs with NaN (as the persistent LSTM kernels do): a broken hand-off ends in a NaN loss, never in a
// silently wrong one.  The flag is zeroed by a tpgsr_zero launch earlier on the stream (the recorded plans zero all their flags at once).
//
// (The first form of this file let EVERY workgroup sum all rows itself: alone on the chip that beat the separate launch by 5-9 us per
//  BatchNorm, inside the three-stream train step it lost -- profiles/r05e_bn_derive_ab.md.)
#pragma once
#include "common.h"

#define BND_CB 16          // channels per deriver workgroup
#define BND_U 8            // row loads in flight per deriver thread
#define BND_SPINS (1 << 19)

__host__ __device__ inline int bnd_derivers(int C) { return (C + BND_CB - 1) / BND_CB; }
__host__ __device__ inline bool bnd_shape_ok(int C) { return C >= 8 && C <= 512 && (C & 7) == 0 && (C <= BND_CB || C % BND_CB == 0); }

// deriver workgroup k (256 threads): per-channel sums of its <= 16 channels over all rows -> sums[0][i], sums[1][i] (LDS, 2 x 16 doubles);
// scr: LDS scratch of 32 x 32 doubles.  Thread (rl, q) = (tid >> 3, tid & 7): row lane rl of 32, column quad q: q < nq the quads of
// [.][0][c0 ..], q >= 4 those of [.][1][c0 ..].  Ends behind a barrier.
__device__ __forceinline__ void bnd_slice_sums(const float* __restrict__ rows, const int nrows, const int C, const int k, const int tid,
                                               double* __restrict__ scr, double* __restrict__ sums) {
  const int c0 = k * BND_CB, cb = min(BND_CB, C - c0), nq = cb >> 2;
  const int rl = tid >> 3, q = tid & 7, qq = q & 3;
  const bool active = qq < nq;
  const float* col = rows + (q >= 4 ? C : 0) + c0 + 4 * qq;
  const size_t ld = (size_t)2 * C;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  if (active) {
    for (int r = rl; r < nrows; r += BND_U * 32) {
      float4 v[BND_U];
#pragma unroll
      for (int u = 0; u < BND_U; ++u) {
        const int rr = r + u * 32;
        v[u] = rr < nrows ? *reinterpret_cast<const float4*>(col + (size_t)rr * ld) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < BND_U; ++u) {
        a0 += (double)v[u].x;
        a1 += (double)v[u].y;
        a2 += (double)v[u].z;
        a3 += (double)v[u].w;
      }
    }
  }
  double* mine = scr + (rl * 8 + q) * 4;
  mine[0] = a0;
  mine[1] = a1;
  mine[2] = a2;
  mine[3] = a3;
  __syncthreads();
  if (tid < 32) {      // column tid = stat (tid >> 4), channel (tid & 15): the 32 row lanes added in lane order
    double s = 0.0;
    for (int l = 0; l < 32; ++l) s += scr[l * 32 + tid];
    sums[tid] = s;
  }
  __syncthreads();
}

__device__ __forceinline__ void bnd_store_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float bnd_load_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// all threads: make the deriver's published stores globally visible, then one arrival on the flag
__device__ __forceinline__ void bnd_arrive(unsigned* flag, const int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// all threads: wait until `target` derivers have arrived; false when the wait was given up (the caller poisons its outputs)
__device__ __forceinline__ bool bnd_wait(unsigned* flag, const unsigned target, const int tid) {
  __shared__ int s_ok;
  if (tid == 0) {
    int spins = 0, ok = 1;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > BND_SPINS) {
        ok = 0;
        break;
      }
    }
    s_ok = ok;
  }
  __syncthreads();
  return s_ok != 0;
}

// ---- forward: scale / shift of all C channels into LDS (ssc / ssh) ----
__device__ __forceinline__ void bnd_forward(const tpgsr_bn_derive& d, const int tid, double* __restrict__ scr, double* __restrict__ sums,
                                            float* __restrict__ ssc, float* __restrict__ ssh) {
  const int C = d.C, D = bnd_derivers(C);
  if ((int)blockIdx.x < D) {
    const int k = blockIdx.x;
    bnd_slice_sums(d.rows, d.nrows, C, k, tid, scr, sums);
    const int c = k * BND_CB + tid;
    if (tid < BND_CB && c < C) {      // the arithmetic of bn_finalize_kernel (elementwise.hip), value for value
      const double count = (double)d.count;
      const double mean_raw = sums[tid] / count;
      double var = sums[16 + tid] / count - mean_raw * mean_raw;
      if (var < 0.0) var = 0.0;
      const double mean = mean_raw + (d.bias ? (double)d.bias[c] : 0.0);
      const double rstd = 1.0 / sqrt(var + (double)d.eps);
      bnd_store_sc1(d.scale + c, (float)((double)d.gamma[c] * rstd));
      bnd_store_sc1(d.shift + c, (float)((double)d.beta[c] - mean * (double)d.gamma[c] * rstd));
      if (d.save_mean) d.save_mean[c] = (float)mean;
      if (d.save_rstd) d.save_rstd[c] = (float)rstd;
      if (d.running_mean) {
        const double unbiased = d.count > 1 ? var * count / (double)(d.count - 1) : var;
        d.running_mean[c] = (float)((1.0 - d.momentum) * (double)d.running_mean[c] + d.momentum * mean);
        d.running_var[c] = (float)((1.0 - d.momentum) * (double)d.running_var[c] + d.momentum * unbiased);
      }
    }
    bnd_arrive(d.flag, tid);
  }
  const bool ok = bnd_wait(d.flag, (unsigned)D, tid);
  for (int c = tid; c < C; c += blockDim.x) {
    ssc[c] = ok ? bnd_load_sc1(d.scale + c) : __builtin_nanf("");
    ssh[c] = ok ? bnd_load_sc1(d.shift + c) : __builtin_nanf("");
  }
  __syncthreads();
}

// ---- backward: dy = coef0 * dz + coef1 * y + coef2; the coefficients of all channels into LDS (scoef [3][C]) ----
__device__ __forceinline__ void bnd_backward(const tpgsr_bn_derive& d, const int tid, double* __restrict__ scr, double* __restrict__ sums,
                                             float* __restrict__ scoef) {
  const int C = d.C, D = bnd_derivers(C);
  if ((int)blockIdx.x < D) {
    const int k = blockIdx.x;
    bnd_slice_sums(d.rows, d.nrows, C, k, tid, scr, sums);
    const int c = k * BND_CB + tid;
    if (tid < BND_CB && c < C) {      // the arithmetic of bn_bwd_finalize_kernel, value for value
      const double s = sums[tid], sx = sums[16 + tid];
      const double rstd = d.save_rstd[c], mu = d.save_mean[c], g = d.gamma[c];
      const double mdz = s / (double)d.count, mdzx = sx / (double)d.count;
      const double c0 = g * rstd;
      bnd_store_sc1(d.coef + c, (float)c0);
      bnd_store_sc1(d.coef + C + c, (float)(-c0 * mdzx * rstd));
      bnd_store_sc1(d.coef + 2 * C + c, (float)(-c0 * (mdz - mu * rstd * mdzx)));
      if (d.dgamma) d.dgamma[c] = d.accumulate ? d.dgamma[c] + (float)sx : (float)sx;
      if (d.dbeta) d.dbeta[c] = d.accumulate ? d.dbeta[c] + (float)s : (float)s;
    }
    bnd_arrive(d.flag, tid);
  }
  const bool ok = bnd_wait(d.flag, (unsigned)D, tid);
  for (int i = tid; i < 3 * C; i += blockDim.x) scoef[i] = ok ? bnd_load_sc1(d.coef + i) : __builtin_nanf("");
  __syncthreads();
}
