// STN rectification: thin-plate-spline grid generation and bilinear grid_sample (forward + backward).
// model/tps_spatial_transformer.py:97-112 (grid), :10-18 (F.grid_sample: bilinear, zeros padding,
// align_corners False on torch >= 1.3 / True on the authors' torch 1.2 -- both selectable).
#include "common.h"

#define TPS_MAXC 32  // >= num_control_points + 3

// grid generation: one block per (image, 256-pixel chunk); the 23x2 mapping matrix is rebuilt per block
__global__ __launch_bounds__(256) void tps_grid_fwd_kernel(const float* __restrict__ ctrl, const float* __restrict__ inv_kernel,
                                                           const float* __restrict__ repr, int HW, int NC,
                                                           float* __restrict__ grid, float* __restrict__ src) {
  // The TPS system is ill-conditioned (|inverse_kernel| entries >> 1, heavy cancellation): the two small
  // contractions are accumulated in fp64 so this kernel adds no rounding of its own on top of the fp32 inputs.
  __shared__ double map[TPS_MAXC][2];
  const int n = blockIdx.y;
  const int K = NC + 3;
  if (threadIdx.x < 2 * K) {
    int k = threadIdx.x >> 1, c = threadIdx.x & 1;
    double s = 0.0;
    for (int i = 0; i < NC; ++i) s += (double)inv_kernel[k * K + i] * (double)ctrl[((size_t)n * NC + i) * 2 + c];
    map[k][c] = s;  // rows NC..NC+2 of Y are the zero padding_matrix
  }
  __syncthreads();
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= HW) return;
  double dsx = 0.0, dsy = 0.0;
  for (int k = 0; k < K; ++k) {
    double r = (double)repr[(size_t)p * K + k];
    dsx += r * map[k][0];
    dsy += r * map[k][1];
  }
  float sx = (float)dsx, sy = (float)dsy;
  size_t o = ((size_t)n * HW + p) * 2;
  if (src) {
    src[o] = sx;
    src[o + 1] = sy;
  }
  grid[o] = 2.f * fminf(fmaxf(sx, 0.f), 1.f) - 1.f;
  grid[o + 1] = 2.f * fminf(fmaxf(sy, 0.f), 1.f) - 1.f;
}

extern "C" int tpgsr_tps_grid_fwd(const float* ctrl, const float* inv_kernel, const float* coord_repr, int N, int HW, int NC,
                                  float* grid, float* src, void* stream) {
  TPGSR_CHECK_ARG(ctrl && inv_kernel && coord_repr && grid, "tpgsr_tps_grid_fwd: null pointer");
  TPGSR_CHECK_ARG(NC + 3 <= TPS_MAXC && N > 0 && HW > 0, "tpgsr_tps_grid_fwd: bad sizes");
  hipLaunchKernelGGL(tps_grid_fwd_kernel, dim3(cdiv(HW, 256), N), dim3(256), 0, (hipStream_t)stream, ctrl, inv_kernel,
                     coord_repr, HW, NC, grid, src);
  TPGSR_LAUNCH_CHECK("tpgsr_tps_grid_fwd");
}

// dctrl = (inv_kernel^T (repr^T dsrc))[:NC],  dsrc = 2*dgrid where 0 <= src <= 1 (clamp gradient)
__global__ __launch_bounds__(256) void tps_grid_bwd_kernel(const float* __restrict__ dgrid, const float* __restrict__ src,
                                                           const float* __restrict__ inv_kernel, const float* __restrict__ repr,
                                                           int HW, int NC, float* __restrict__ dctrl) {
  __shared__ float red[4][TPS_MAXC][2];
  __shared__ float dmap[TPS_MAXC][2];
  const int n = blockIdx.x;
  const int K = NC + 3;
  float ax[TPS_MAXC], ay[TPS_MAXC];
#pragma unroll
  for (int k = 0; k < TPS_MAXC; ++k) ax[k] = ay[k] = 0.f;
  for (int p = threadIdx.x; p < HW; p += blockDim.x) {
    size_t o = ((size_t)n * HW + p) * 2;
    float sx = src[o], sy = src[o + 1];
    float gx = (sx >= 0.f && sx <= 1.f) ? 2.f * dgrid[o] : 0.f;
    float gy = (sy >= 0.f && sy <= 1.f) ? 2.f * dgrid[o + 1] : 0.f;
#pragma unroll
    for (int k = 0; k < TPS_MAXC; ++k)
      if (k < K) {
        float r = repr[(size_t)p * K + k];
        ax[k] = fmaf(r, gx, ax[k]);
        ay[k] = fmaf(r, gy, ay[k]);
      }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < TPS_MAXC; ++k) {
    float vx = wave_sum(ax[k]), vy = wave_sum(ay[k]);
    if (lane == 0) {
      red[wave][k][0] = vx;
      red[wave][k][1] = vy;
    }
  }
  __syncthreads();
  if (threadIdx.x < 2 * K) {
    int k = threadIdx.x >> 1, c = threadIdx.x & 1;
    dmap[k][c] = red[0][k][c] + red[1][k][c] + red[2][k][c] + red[3][k][c];
  }
  __syncthreads();
  if (threadIdx.x < 2 * NC) {
    int i = threadIdx.x >> 1, c = threadIdx.x & 1;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s = fmaf(inv_kernel[k * K + i], dmap[k][c], s);
    dctrl[((size_t)n * NC + i) * 2 + c] = s;
  }
}

extern "C" int tpgsr_tps_grid_bwd(const float* dgrid, const float* src, const float* inv_kernel, const float* coord_repr,
                                  int N, int HW, int NC, float* dctrl, void* stream) {
  TPGSR_CHECK_ARG(dgrid && src && inv_kernel && coord_repr && dctrl, "tpgsr_tps_grid_bwd: null pointer");
  TPGSR_CHECK_ARG(NC + 3 <= TPS_MAXC && N > 0 && HW > 0, "tpgsr_tps_grid_bwd: bad sizes");
  hipLaunchKernelGGL(tps_grid_bwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, dgrid, src, inv_kernel, coord_repr, HW,
                     NC, dctrl);
  TPGSR_LAUNCH_CHECK("tpgsr_tps_grid_bwd");
}

// ------------------------------------------------------------------------------------------------------
// bilinear grid_sample, NHWC, C <= 4 channels per pixel handled by one thread
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float unnorm(float g, int size, int align) {
  return align ? (g + 1.f) * 0.5f * (float)(size - 1) : ((g + 1.f) * (float)size - 1.f) * 0.5f;
}

template <int C>
__global__ __launch_bounds__(256) void grid_sample_fwd_kernel(const float* __restrict__ in, const float* __restrict__ grid,
                                                              int N, int H, int W, int OHW, int align, float* __restrict__ out) {
  long long total = (long long)N * OHW;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int n = (int)(i / OHW);
  float ix = unnorm(grid[i * 2], W, align), iy = unnorm(grid[i * 2 + 1], H, align);
  float fx = floorf(ix), fy = floorf(iy);
  int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
  float wx1 = ix - fx, wx0 = 1.f - wx1, wy1 = iy - fy, wy0 = 1.f - wy1;
  float acc[C];
#pragma unroll
  for (int c = 0; c < C; ++c) acc[c] = 0.f;
  auto tap = [&](int y, int x, float w) {
    if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
      const float* p = in + ((size_t)(n * H + y) * W + x) * C;
#pragma unroll
      for (int c = 0; c < C; ++c) acc[c] = fmaf(w, p[c], acc[c]);
    }
  };
  tap(y0, x0, wy0 * wx0);
  tap(y0, x1, wy0 * wx1);
  tap(y1, x0, wy1 * wx0);
  tap(y1, x1, wy1 * wx1);
#pragma unroll
  for (int c = 0; c < C; ++c) out[i * C + c] = acc[c];
}

extern "C" int tpgsr_grid_sample_fwd(const float* in, const float* grid, int N, int H, int W, int C, int OH, int OW,
                                     int align_corners, float* out, void* stream) {
  TPGSR_CHECK_ARG(in && grid && out && N > 0 && H > 0 && W > 0, "tpgsr_grid_sample_fwd: bad arguments");
  TPGSR_CHECK_ARG(C == 3 || C == 4 || C == 1, "tpgsr_grid_sample_fwd: C must be 1, 3 or 4 (got %d)", C);
  long long total = (long long)N * OH * OW;
  dim3 g(cdiv(total, 256)), b(256);
  hipStream_t s = (hipStream_t)stream;
  if (C == 4) hipLaunchKernelGGL(grid_sample_fwd_kernel<4>, g, b, 0, s, in, grid, N, H, W, OH * OW, align_corners, out);
  else if (C == 3) hipLaunchKernelGGL(grid_sample_fwd_kernel<3>, g, b, 0, s, in, grid, N, H, W, OH * OW, align_corners, out);
  else hipLaunchKernelGGL(grid_sample_fwd_kernel<1>, g, b, 0, s, in, grid, N, H, W, OH * OW, align_corners, out);
  TPGSR_LAUNCH_CHECK("tpgsr_grid_sample_fwd");
}

template <int C>
__global__ __launch_bounds__(256) void grid_sample_bwd_kernel(const float* __restrict__ in, const float* __restrict__ grid,
                                                              const float* __restrict__ dout, int N, int H, int W, int OHW,
                                                              int align, float* __restrict__ dgrid) {
  long long total = (long long)N * OHW;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int n = (int)(i / OHW);
  float ix = unnorm(grid[i * 2], W, align), iy = unnorm(grid[i * 2 + 1], H, align);
  float fx = floorf(ix), fy = floorf(iy);
  int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
  float wx1 = ix - fx, wx0 = 1.f - wx1, wy1 = iy - fy, wy0 = 1.f - wy1;
  float g[C];
#pragma unroll
  for (int c = 0; c < C; ++c) g[c] = dout[i * C + c];
  float gix = 0.f, giy = 0.f;
  // corner value I, its weight w, and d w/d ix = sx * (y-weight), d w/d iy = sy * (x-weight)
  auto tap = [&](int y, int x, float w, float dwx, float dwy) {
    if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
      size_t o = ((size_t)(n * H + y) * W + x) * C;
      float dot = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) dot = fmaf(in[o + c], g[c], dot);
      (void)w;
      gix = fmaf(dwx, dot, gix);
      giy = fmaf(dwy, dot, giy);
    }
  };
  tap(y0, x0, wy0 * wx0, -wy0, -wx0);
  tap(y0, x1, wy0 * wx1, wy0, -wx1);
  tap(y1, x0, wy1 * wx0, -wy1, wx0);
  tap(y1, x1, wy1 * wx1, wy1, wx1);
  if (dgrid) {
    float mx = align ? 0.5f * (float)(W - 1) : 0.5f * (float)W;
    float my = align ? 0.5f * (float)(H - 1) : 0.5f * (float)H;
    dgrid[i * 2] = gix * mx;
    dgrid[i * 2 + 1] = giy * my;
  }
}

// gradient with respect to the sampled IMAGE in GATHER form (round 5; it was the library's last floating-point atomicAdd): a thread
// owns one input pixel and walks ALL output positions of its image in index order (staged through LDS in chunks: every lane of a wave
// reads the same entry, a broadcast), adding weight * dout for those whose bilinear footprint {x0, x0+1} x {y0, y0+1} contains the
// pixel -- the same weights the forward pass used, summed in a fixed order: deterministic, no memset.  O(HW * OHW) compares per image
// (1 M for the 16 x 64 STN input: microseconds); no recorded train step asks for this gradient (the image is data), the functional
// operator layer does.
#define GS_CHUNK 512
template <int C>
__global__ __launch_bounds__(256) void grid_sample_bwd_din_kernel(const float* __restrict__ grid, const float* __restrict__ dout,
                                                                  int H, int W, int OHW, int align, float* __restrict__ din) {
  __shared__ int sx0[GS_CHUNK], sy0[GS_CHUNK];
  __shared__ float swx[GS_CHUNK], swy[GS_CHUNK], sg[GS_CHUNK][C];
  const int n = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int y = p / W, x = p - y * W;
  float acc[C];
#pragma unroll
  for (int c = 0; c < C; ++c) acc[c] = 0.f;
  for (int base = 0; base < OHW; base += GS_CHUNK) {
    const int cnt = min(GS_CHUNK, OHW - base);
    __syncthreads();
    for (int j = threadIdx.x; j < cnt; j += 256) {
      const size_t i = (size_t)n * OHW + base + j;
      const float ix = unnorm(grid[i * 2], W, align), iy = unnorm(grid[i * 2 + 1], H, align);
      const float fx = floorf(ix), fy = floorf(iy);
      sx0[j] = (int)fx;
      sy0[j] = (int)fy;
      swx[j] = ix - fx;
      swy[j] = iy - fy;
#pragma unroll
      for (int c = 0; c < C; ++c) sg[j][c] = dout[i * C + c];
    }
    __syncthreads();
    for (int j = 0; j < cnt; ++j) {
      const unsigned dx = (unsigned)(x - sx0[j]), dy = (unsigned)(y - sy0[j]);
      if (dx < 2u && dy < 2u) {
        const float wx1 = swx[j], wy1 = swy[j];
        const float w = (dy ? wy1 : 1.f - wy1) * (dx ? wx1 : 1.f - wx1);
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = fmaf(w, sg[j][c], acc[c]);
      }
    }
  }
  if (p < H * W) {
#pragma unroll
    for (int c = 0; c < C; ++c) din[((size_t)n * H * W + p) * C + c] = acc[c];
  }
}

extern "C" int tpgsr_grid_sample_bwd(const float* in, const float* grid, const float* dout, int N, int H, int W, int C, int OH,
                                     int OW, int align_corners, float* din, float* dgrid, void* stream) {
  TPGSR_CHECK_ARG(in && grid && dout && (din || dgrid), "tpgsr_grid_sample_bwd: null pointer");
  TPGSR_CHECK_ARG(C == 3 || C == 4 || C == 1, "tpgsr_grid_sample_bwd: C must be 1, 3 or 4 (got %d)", C);
  hipStream_t s = (hipStream_t)stream;
  if (din) {
    dim3 gd(cdiv((long long)H * W, 256), N), bd(256);
    if (C == 4) hipLaunchKernelGGL(grid_sample_bwd_din_kernel<4>, gd, bd, 0, s, grid, dout, H, W, OH * OW, align_corners, din);
    else if (C == 3) hipLaunchKernelGGL(grid_sample_bwd_din_kernel<3>, gd, bd, 0, s, grid, dout, H, W, OH * OW, align_corners, din);
    else hipLaunchKernelGGL(grid_sample_bwd_din_kernel<1>, gd, bd, 0, s, grid, dout, H, W, OH * OW, align_corners, din);
  }
  if (dgrid) {
    long long total = (long long)N * OH * OW;
    dim3 g(cdiv(total, 256)), b(256);
    if (C == 4) hipLaunchKernelGGL(grid_sample_bwd_kernel<4>, g, b, 0, s, in, grid, dout, N, H, W, OH * OW, align_corners, dgrid);
    else if (C == 3) hipLaunchKernelGGL(grid_sample_bwd_kernel<3>, g, b, 0, s, in, grid, dout, N, H, W, OH * OW, align_corners, dgrid);
    else hipLaunchKernelGGL(grid_sample_bwd_kernel<1>, g, b, 0, s, in, grid, dout, N, H, W, OH * OW, align_corners, dgrid);
  }
  TPGSR_LAUNCH_CHECK("tpgsr_grid_sample_bwd");
}
