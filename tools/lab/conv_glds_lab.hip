// Lab: 3x3 NHWC conv 64 -> 64 (the RRB conv of the C2 step: N = 48, 16 x 64) with the tile loop built around direct
// global -> LDS loads (global_load_lds b128) instead of register staging:
//   * A image in LDS is [64 pixels][8 quads] with the quad position XOR-swizzled by (m >> 1) & 7 on the SOURCE side
//     (the LDS side of an LDS-DMA is lane-linear), A fragments by ds_read_b128 (4 per chunk instead of 16 b32 reads)
//   * the K index of MFMA j for lane half h is 4 * (2 * (j >> 2) + h) + (j & 3), B fragments follow the same order
//   * two LDS buffers, one barrier per chunk, padding pixels read a zero page
// Checks a sample of outputs against a host fp64 reference, then times it.
//   hipcc --offload-arch=gfx950 -O3 -w -o conv_glds_lab conv_glds_lab.hip && ./conv_glds_lab
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
#define C 64
#define KC 32
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ void glds16(const float* src, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)lds_wave_base, 16, 0, 0);
}

template <int NBUF>   // 2: one chunk in flight, __syncthreads(); 3: two chunks in flight, counted vmcnt + raw s_barrier
__global__ __launch_bounds__(256) void conv3x3_glds(const float* __restrict__ in, const float* __restrict__ wt,
                                                    const float* __restrict__ zero, float* __restrict__ out, int N, int H,
                                                    int W, int M) {
  __shared__ __attribute__((aligned(16))) float As[NBUF][64 * 32];
  __shared__ __attribute__((aligned(16))) float Bs[NBUF][32 * 64];
  const int tid = threadIdx.x, L = tid & 63, w = tid >> 6;
  const int wm = w & 1, wn = w >> 1;
  const int m0 = blockIdx.x * 64;
  // the two A pixels this lane feeds (fixed over all chunks) and its swizzled source quad
  int pm[2], pn[2], poh[2], pow_[2], pq[2];
  bool pv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int ml = (w * 2 + i) * 8 + (L >> 3);
    pm[i] = ml;
    int m = m0 + ml;
    pv[i] = m < M;
    int mm = pv[i] ? m : 0;
    pn[i] = mm / (H * W);
    int r = mm - pn[i] * H * W;
    poh[i] = r / W;
    pow_[i] = r - poh[i] * W;
    pq[i] = (L & 7) ^ ((ml >> 1) & 7);
  }
  const int brow = (L >> 4), bcol4 = (L & 15) * 4;
  auto issue = [&](int ch, int buf) {
    const int tap = ch >> 1, c0 = (ch & 1) * 32;
    const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int ih = poh[i] + kh - 1, iw = pow_[i] + kw - 1;
      bool ok = pv[i] && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
      const float* src = ok ? in + ((size_t)(pn[i] * H + ih) * W + iw) * C + c0 + pq[i] * 4 : zero + (L & 7) * 4;
      glds16(src, &As[buf][(w * 2 + i) * 8 * 32]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int k = (w * 2 + i) * 4 + brow;
      glds16(wt + (size_t)(ch * KC + k) * C + bcol4, &Bs[buf][(w * 2 + i) * 4 * 64]);
    }
  };
  floatx16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int nchunks = 18;
  issue(0, 0);
  if (NBUF == 3) {
    issue(1, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  } else {
    __syncthreads();
  }
  const int am = wm * 32 + (L & 31), h = L >> 5;
  const int bn = wn * 32 + (L & 31);
  for (int ch = 0; ch < nchunks; ++ch) {
    const int buf = ch % NBUF;
    if (ch + NBUF - 1 < nchunks) issue(ch + NBUF - 1, (ch + NBUF - 1) % NBUF);
    float4 a4[4];
    float b[16];
#pragma unroll
    for (int qi = 0; qi < 4; ++qi) {
      int pos = (2 * qi + h) ^ ((am >> 1) & 7);
      a4[qi] = *reinterpret_cast<const float4*>(&As[buf][am * 32 + pos * 4]);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) b[j] = Bs[buf][(4 * (2 * (j >> 2) + h) + (j & 3)) * 64 + bn];
#pragma unroll
    for (int qi = 0; qi < 4; ++qi) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[qi].x, b[4 * qi + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[qi].y, b[4 * qi + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[qi].z, b[4 * qi + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[qi].w, b[4 * qi + 3], acc, 0, 0, 0);
    }
    if (NBUF == 3) {   // chunk ch+1 must have landed; chunk ch+2 (4 LDS-DMAs per wave) may stay in flight across the barrier
      if (ch + 2 < nchunks) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    } else {
      __syncthreads();   // drains the LDS-DMA of chunk ch+1 (vmcnt(0)) and orders buffer reuse
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (L >> 5);
    int m = m0 + wm * 32 + row;
    if (m < M) out[(size_t)m * C + bn] = acc[r];
  }
}


// ---- halo variant -----------------------------------------------------------------------------------------------------
// One workgroup = one image row (W = 64 pixels = the 64-pixel tile).  Instead of gathering the A tile of every
// (tap, 32-channel) chunk from global memory (9 fetches of every input pixel), the 3 x 66 pixel halo of the row is brought
// into LDS once per 32-channel pass and the 9 taps are walked out of LDS; only the weight chunks still stream (two LDS
// buffers, one barrier per chunk).  A global traffic per workgroup: 2 x 25 KB instead of 147 KB.
__global__ __launch_bounds__(256) void conv3x3_halo(const float* __restrict__ in, const float* __restrict__ wt,
                                                    const float* __restrict__ zero, float* __restrict__ out, int N, int H,
                                                    int W, int M) {
  constexpr int HP = 3 * 66;                                    // halo pixels
  constexpr int HPAD = (HP + 7) / 8 * 8;                        // 8 pixels per LDS-DMA instruction
  __shared__ __attribute__((aligned(16))) float Hs[HPAD * 32];  // [halo pixel][8 quads of 32 channels], quad pos = q ^ ((P >> 1) & 7)
  __shared__ __attribute__((aligned(16))) float Bs[2][32 * 64];
  const int tid = threadIdx.x, L = tid & 63, w = tid >> 6;
  const int wm = w & 1, wn = w >> 1;
  const int m0 = blockIdx.x * 64;                               // = (n * H + oh) * W
  const int n = m0 / (H * W), oh = (m0 - n * H * W) / W;
  const int brow = (L >> 4), bcol4 = (L & 15) * 4;
  auto issue_b = [&](int kchunk, int buf) {                     // rows kchunk*32 .. +31 of the [576][64] weight matrix
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int k = (w * 2 + i) * 4 + brow;
      glds16(wt + (size_t)(kchunk * KC + k) * C + bcol4, &Bs[buf][(w * 2 + i) * 4 * 64]);
    }
  };
  auto issue_halo = [&](int c0) {
    for (int g = w; g < HPAD / 8; g += 4) {                     // 8 halo pixels per instruction
      int P = g * 8 + (L >> 3);
      int r = P / 66, cc = P - r * 66;                          // halo row 0..2 (ih = oh - 1 + r), halo col 0..65 (iw = cc - 1)
      int ih = oh - 1 + r, iw = cc - 1;
      bool ok = P < HP && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
      int q = (L & 7) ^ ((P >> 1) & 7);
      const float* src = ok ? in + ((size_t)(n * H + ih) * W + iw) * C + c0 + q * 4 : zero + (L & 7) * 4;
      glds16(src, &Hs[g * 8 * 32]);
    }
  };
  floatx16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int am = wm * 32 + (L & 31), h = L >> 5;
  const int bn = wn * 32 + (L & 31);
  int bbuf = 0;
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();                                            // everyone is done reading the previous pass's halo / B buffer
    issue_halo(pass * 32);
    issue_b((0 * 2 + pass), bbuf);                              // weight rows of (tap 0, this pass's channels): k = tap*64 + pass*32
    __syncthreads();                                            // (drains the LDS-DMAs: vmcnt(0))
    for (int tap = 0; tap < 9; ++tap) {
      if (tap + 1 < 9) issue_b(((tap + 1) * 2 + pass), bbuf ^ 1);
      const int kh = tap / 3, kw = tap - kh * 3;
      const int P = kh * 66 + am + kw;
      float4 a4[4];
      float b[16];
#pragma unroll
      for (int qi = 0; qi < 4; ++qi) {
        int pos = (2 * qi + h) ^ ((P >> 1) & 7);
        a4[qi] = *reinterpret_cast<const float4*>(&Hs[P * 32 + pos * 4]);
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) b[j] = Bs[bbuf][(4 * (2 * (j >> 2) + h) + (j & 3)) * 64 + bn];
#pragma unroll
      for (int qi = 0; qi < 4; ++qi) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[qi].x, b[4 * qi + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[qi].y, b[4 * qi + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[qi].z, b[4 * qi + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[qi].w, b[4 * qi + 3], acc, 0, 0, 0);
      }
      __syncthreads();
      bbuf ^= 1;
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (L >> 5);
    int m = m0 + wm * 32 + row;
    if (m < M) out[(size_t)m * C + bn] = acc[r];
  }
}


// ---- weights-stationary variant ---------------------------------------------------------------------------------------
// One workgroup per CU-sized slab: 12 waves = 3 groups x 4 waves, each group owns one image row (64 pixels); the 3 x 3 x 32 x 64
// weight slice of a 32-channel pass (72 KB) is brought into LDS ONCE per workgroup and shared by the three groups, each
// group brings its 3 x 66 pixel halo (25 KB).  Then 9 taps x 16 MFMAs per wave run straight out of LDS with no barrier and
// no staging in between.  Global traffic per launch: weights 256 x 147 KB + input 768 x 50 KB = 76 MB (im2col tiles: 226 MB).
__global__ __launch_bounds__(768) void conv3x3_wstat(const float* __restrict__ in, const float* __restrict__ wt,
                                                     const float* __restrict__ zero, float* __restrict__ out, int N, int H,
                                                     int W, int M) {
  constexpr int HP = 3 * 66, HPAD = (HP + 7) / 8 * 8;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;                         // [9][32][64]
  float* HsAll = smem + 9 * 32 * 64;        // [3 groups][HPAD * 32]
  const int tid = threadIdx.x, L = tid & 63, wv = tid >> 6;   // 12 waves
  const int grp = wv >> 2, w = wv & 3;
  const int wm = w & 1, wn = w >> 1;
  const int row = blockIdx.x * 3 + grp;                        // global image row (n * H + oh)
  const int m0 = row * 64;
  const bool rowok = m0 < M;
  const int n = rowok ? m0 / (H * W) : 0, oh = rowok ? (m0 - n * H * W) / W : 0;
  float* Hs = HsAll + grp * HPAD * 32;
  floatx16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int am = wm * 32 + (L & 31), h = L >> 5;
  const int bn = wn * 32 + (L & 31);
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
    // weights of this pass: rows tap*64 + pass*32 + r -> Ws[tap][r][:], 4 rows per LDS-DMA instruction, 72 instructions / 12 waves
    for (int g = wv; g < 72; g += 12) {
      int rr = g * 4 + (L >> 4);                               // 0..287 = tap*32 + r
      int tap = rr >> 5, r = rr & 31;
      glds16(wt + (size_t)(tap * 64 + pass * 32 + r) * C + (L & 15) * 4, &Ws[g * 4 * 64]);
    }
    for (int g = w; g < HPAD / 8; g += 4) {                    // this group's halo, 8 pixels per instruction
      int P = g * 8 + (L >> 3);
      int r = P / 66, cc = P - r * 66;
      int ih = oh - 1 + r, iw = cc - 1;
      bool ok = rowok && P < HP && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
      int q = (L & 7) ^ ((P >> 1) & 7);
      const float* src = ok ? in + ((size_t)(n * H + ih) * W + iw) * C + pass * 32 + q * 4 : zero + (L & 7) * 4;
      glds16(src, &Hs[g * 8 * 32]);
    }
    __syncthreads();
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      const int kh = tap / 3, kw = tap - kh * 3;
      const int P = kh * 66 + am + kw;
      const float* Wt = Ws + tap * 32 * 64;
      float4 a4[4];
      float b[16];
#pragma unroll
      for (int qi = 0; qi < 4; ++qi) {
        int pos = (2 * qi + h) ^ ((P >> 1) & 7);
        a4[qi] = *reinterpret_cast<const float4*>(&Hs[P * 32 + pos * 4]);
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) b[j] = Wt[(4 * (2 * (j >> 2) + h) + (j & 3)) * 64 + bn];
#pragma unroll
      for (int qi = 0; qi < 4; ++qi) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[qi].x, b[4 * qi + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[qi].y, b[4 * qi + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[qi].z, b[4 * qi + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[qi].w, b[4 * qi + 3], acc, 0, 0, 0);
      }
    }
  }
  if (rowok) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int rw = (r & 3) + 8 * (r >> 2) + 4 * (L >> 5);
      int m = m0 + wm * 32 + rw;
      if (m < M) out[(size_t)m * C + bn] = acc[r];
    }
  }
}


// ---- weights-stationary, 16-channel passes, double-buffered stages ----------------------------------------------------------
// Same ownership as conv3x3_wstat, but the K range is cut into four 16-channel passes whose LDS images (weights 36 KB + three
// 12.5 KB halos) are double-buffered: the LDS-DMAs of pass p+1 fly while pass p's 9 x 8 MFMAs per wave run.
__global__ __launch_bounds__(768) void conv3x3_wstat_db(const float* __restrict__ in, const float* __restrict__ wt,
                                                        const float* __restrict__ zero, float* __restrict__ out, int N, int H,
                                                        int W, int M) {
  constexpr int HP = 3 * 66, HPAD = (HP + 15) / 16 * 16;      // 16 halo pixels (64 B each) per LDS-DMA instruction
  constexpr int WSZ = 9 * 16 * 64, HSZ = HPAD * 16, STAGE = WSZ + 3 * HSZ;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, L = tid & 63, wv = tid >> 6;
  const int grp = wv >> 2, w = wv & 3;
  const int wm = w & 1, wn = w >> 1;
  const int row = blockIdx.x * 3 + grp;
  const int m0 = row * 64;
  const bool rowok = m0 < M;
  const int n = rowok ? m0 / (H * W) : 0, oh = rowok ? (m0 - n * H * W) / W : 0;
  auto issue = [&](int pass, int st) {
    float* Ws = smem + st * STAGE;
    float* Hs = Ws + WSZ + grp * HSZ;
    for (int g = wv; g < 36; g += 12) {                        // 144 weight rows, 4 per instruction
      int rr = g * 4 + (L >> 4);
      int tap = rr >> 4, r = rr & 15;
      glds16(wt + (size_t)(tap * 64 + pass * 16 + r) * C + (L & 15) * 4, &Ws[g * 4 * 64]);
    }
    for (int g = w; g < HPAD / 16; g += 4) {
      int P = g * 16 + (L >> 2);
      int r = P / 66, cc = P - r * 66;
      int ih = oh - 1 + r, iw = cc - 1;
      bool ok = rowok && P < HP && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
      int q = (L & 3) ^ ((P >> 2) & 3);
      const float* src = ok ? in + ((size_t)(n * H + ih) * W + iw) * C + pass * 16 + q * 4 : zero + (L & 3) * 4;
      glds16(src, &Hs[g * 16 * 16]);
    }
  };
  floatx16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int am = wm * 32 + (L & 31), h = L >> 5;
  const int bn = wn * 32 + (L & 31);
  issue(0, 0);
  __syncthreads();
  for (int pass = 0; pass < 4; ++pass) {
    const int st = pass & 1;
    if (pass + 1 < 4) issue(pass + 1, st ^ 1);
    const float* Ws = smem + st * STAGE;
    const float* Hs = Ws + WSZ + grp * HSZ;
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      const int kh = tap / 3, kw = tap - kh * 3;
      const int P = kh * 66 + am + kw;
      const float* Wt = Ws + tap * 16 * 64;
      float4 a4[2];
      float b[8];
#pragma unroll
      for (int qi = 0; qi < 2; ++qi) {
        int pos = (2 * qi + h) ^ ((P >> 2) & 3);
        a4[qi] = *reinterpret_cast<const float4*>(&Hs[P * 16 + pos * 4]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) b[j] = Wt[(4 * (2 * (j >> 2) + h) + (j & 3)) * 64 + bn];
#pragma unroll
      for (int qi = 0; qi < 2; ++qi) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[qi].x, b[4 * qi + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[qi].y, b[4 * qi + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[qi].z, b[4 * qi + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[qi].w, b[4 * qi + 3], acc, 0, 0, 0);
      }
    }
    __syncthreads();   // next stage landed (vmcnt(0)), this stage free for pass + 2
  }
  if (rowok) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int rw = (r & 3) + 8 * (r >> 2) + 4 * (L >> 5);
      int m = m0 + wm * 32 + rw;
      if (m < M) out[(size_t)m * C + bn] = acc[r];
    }
  }
}

int main() {
  const int N = 48, H = 16, W = 64, M = N * H * W;
  std::vector<float> hin((size_t)M * C), hwt(576 * C), hout((size_t)M * C);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : hin) v = rnd();
  for (auto& v : hwt) v = rnd() * 0.1f;
  float *in, *wt, *zero, *out;
  hipMalloc(&in, hin.size() * 4); hipMalloc(&wt, hwt.size() * 4); hipMalloc(&zero, 256); hipMalloc(&out, hout.size() * 4);
  hipMemcpy(in, hin.data(), hin.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(wt, hwt.data(), hwt.size() * 4, hipMemcpyHostToDevice);
  hipMemset(zero, 0, 256);
  hipMemset(out, 0xff, hout.size() * 4);
  const size_t ws_lds = (size_t)(9 * 32 * 64 + 3 * 200 * 32) * 4;
  hipFuncSetAttribute((const void*)conv3x3_wstat, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ws_lds);
  const size_t wsdb_lds = (size_t)2 * (9 * 16 * 64 + 3 * 208 * 16) * 4;
  hipFuncSetAttribute((const void*)conv3x3_wstat_db, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wsdb_lds);
  for (int variant = 4; variant <= 6; ++variant) {
    auto launch = [&]() {
      if (variant == 2) hipLaunchKernelGGL(conv3x3_glds<2>, dim3((M + 63) / 64), dim3(256), 0, 0, in, wt, zero, out, N, H, W, M);
      else if (variant == 3) hipLaunchKernelGGL(conv3x3_glds<3>, dim3((M + 63) / 64), dim3(256), 0, 0, in, wt, zero, out, N, H, W, M);
      else if (variant == 4) hipLaunchKernelGGL(conv3x3_halo, dim3((M + 63) / 64), dim3(256), 0, 0, in, wt, zero, out, N, H, W, M);   // 4 = halo
      else if (variant == 5) hipLaunchKernelGGL(conv3x3_wstat, dim3((M / 64 + 2) / 3), dim3(768), ws_lds, 0, in, wt, zero, out, N, H, W, M);  // 5 = weights-stationary
      else hipLaunchKernelGGL(conv3x3_wstat_db, dim3((M / 64 + 2) / 3), dim3(768), wsdb_lds, 0, in, wt, zero, out, N, H, W, M);   // 6 = + double-buffered 16-channel passes
    };
    hipMemset(out, 0xff, hout.size() * 4);
    launch();
    hipDeviceSynchronize();
    hipMemcpy(hout.data(), out, hout.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int t = 0; t < 4000; ++t) {
      int m = (int)(((unsigned long long)t * 2654435761ull) % M), n = (t * 7) % C;
      if (t < 200) m = t * 5;             // first rows: top padding
      int nn = m / (H * W), r = m % (H * W), oh = r / W, ow = r % W;
      double ref = 0;
      for (int kh = 0; kh < 3; ++kh)
        for (int kw = 0; kw < 3; ++kw) {
          int ih = oh + kh - 1, iw = ow + kw - 1;
          if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
          for (int c = 0; c < C; ++c)
            ref += (double)hin[((size_t)(nn * H + ih) * W + iw) * C + c] * hwt[(size_t)((kh * 3 + kw) * C + c) * C + n];
        }
      double e = fabs(ref - hout[(size_t)m * C + n]);
      if (e > maxerr) maxerr = e;
    }
    printf("NBUF=%d  max |err| over 4000 sampled outputs: %.3e %s\n", variant, maxerr, maxerr < 1e-4 ? "(ok)" : "(WRONG)");
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 50;
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double us = 1e3 * ms / reps;
    printf("NBUF=%d  conv3x3 64->64 glds: %.1f us  %.1f TFLOP/s  (product kernel: 45 us)\n", variant, us, 2.0 * M * 576 * 64 / us / 1e6);
  }
  return 0;
}
