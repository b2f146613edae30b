// Row-panel kernel for the 1x1 convolutions / linear maps with a SHORT contraction (K = Cin <= 192) over MANY pixels: the GruBlock
// projections of the SR network (model/tsrn.py:491-508: conv1 composed with the GRU's input projection, 64 or 96 -> 192, and their
// data gradients 192 -> 64 / 32), 25 launches per TPGSR training step at M = N H W = 49152 pixels.
//
// On the tile loop (conv_xbf.hip: 64 x 64 tiles, K chunks of 32 behind a barrier each) these launches have no steady state: two to six
// chunks per tile, every one a full L2 round trip, three column tiles re-loading and re-splitting the same pixels -- 20-39 us per launch
// for 50 MB of traffic and 3-7 us of matrix work (12-17 TFLOP/s fp32-equivalent, 4-14 % of the split-operand roof).  Here a
// workgroup owns 64 PIXELS x ALL output columns (<= 192):
//   * the whole [64 x K] activation panel is loaded at once (every 16-byte load of the workgroup is issued before the first is
//     waited for; 128 contiguous bytes per pixel and load instruction), the fused prologue applied, split into bf16 terms ONCE
//     and stored to LDS: rows of K bf16 padded by 16 B, so the 16-lane groups of a ds_read_b128 fragment read land in 16 distinct
//     bank quads (row stride = (K / 8 + 1) x 16 B with K / 8 even -> odd multiple of 16 B mod 256);
//   * the weights never touch LDS: fragment-ordered split planes straight into registers (as in the other xbf kernels), the first
//     k-block's fragments requested before the activation loads, the next k-block's under the current one's MFMAs;
//   * 4 waves = 2 row blocks x 2 column halves of NBW 32-column blocks each; one barrier in the whole kernel (two with BN statistics);
//   * epilogue shared with the tile loop (bias, activation, pixel-shuffle store, BN partial statistics).
// 768 workgroups at M = 49152: one resident round at three workgroups per CU.
#include "conv_xbf_common.h"
#include <stdlib.h>
#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>

// NBW: 32-column blocks per wave (the workgroup covers 64 NBW columns); NQ8 = K / 32 = quads per thread and pixel row
template <int LD, int T, int NBW, int NQ8>
__global__ __launch_bounds__(256, 2) void conv_panel_xbf_kernel(tpgsr_conv_args a, int M) {
  constexpr int row_bytes = NQ8 * 64 + 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char psm[];    // [T][64 rows][row_bytes]
  constexpr int PLANE = 64 * row_bytes;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1, g = lane >> 5;
  const int mblk = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = mblk * 64;
  constexpr int KB16 = NQ8 * 2;

  // ---- weights: fragment-ordered planes [term][n / 32][k / 16][lane][8] bf16 (tpgsr_split_bf_program) ----
  const int wrows = a.wt_ld > 0 ? a.wt_ld : a.Cout;
  const int NB32 = (wrows + 31) >> 5;
  const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(reinterpret_cast<const float*>(a.wt_bf), (size_t)T * NB32 * 32 * a.kp / 2);
  const unsigned plane_w = (unsigned)NB32 * KB16 * 1024u;
  const int ncol0 = wn * 32 * NBW;
  unsigned woff[NBW];
#pragma unroll
  for (int j = 0; j < NBW; ++j)      // a column block entirely past Cout: zeros (hardware zero fill of the out-of-range offset)
    woff[j] = ncol0 + 32 * j < a.Cout ? ((unsigned)((a.wt_coff + ncol0 + 32 * j) >> 5) * KB16) * 1024u + lane * 16u : OOB_OFF;
  u32x4 bw[2][NBW][T];
  auto fetch_w = [&](auto set_tag, const int kb) __attribute__((always_inline)) {
    constexpr int S = decltype(set_tag)::value;
#pragma unroll
    for (int j = 0; j < NBW; ++j)
#pragma unroll
      for (int t = 0; t < T; ++t)
        bw[S][j][t] = __builtin_amdgcn_raw_buffer_load_b128(
            rs_w, (woff[j] == OOB_OFF || kb >= KB16) ? (int)OOB_OFF : (int)(woff[j] + t * plane_w + (unsigned)kb * 1024u), 0, 0);
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  fetch_w(S0{}, 0);

  // ---- activations: 8 threads per pixel row (128 contiguous bytes per row and load), rows (tid >> 3) + 32 p, quads (tid & 7) + 8 j ----
  const int Wr_ = real_w(a);
  const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(a.in, (size_t)a.N * a.H * Wr_ * a.in_ld);
  const __amdgpu_buffer_rsrc_t rs_in2 = (LD & 16) ? make_rsrc(a.in_b, (size_t)a.N * Wr_ * a.in_b_ld)
                                                  : make_rsrc(a.in2 ? a.in2 : a.in, (size_t)a.N * a.H * Wr_ * a.in2_ld);
  ARaw qa[2][NQ8];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const PixelPos px = decode_pixel(a, m0 + (tid >> 3) + 32 * p, M);
#pragma unroll
    for (int j = 0; j < NQ8; ++j) qa[p][j] = load_a_raw<LD>(a, rs_in, rs_in2, px, KPos{0, 0, ((tid & 7) + 8 * j) * 4});
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int row = (tid >> 3) + 32 * p;
#pragma unroll
    for (int j = 0; j < NQ8; ++j) {
        const int c = ((tid & 7) + 8 * j) * 4;
        float4 qs = make_float4(1.f, 1.f, 1.f, 1.f), qt = make_float4(0.f, 0.f, 0.f, 0.f);
        if (LD & 1) {
          const int cc = c < a.Cin ? c : 0;
          qs = *reinterpret_cast<const float4*>(a.in_scale + cc);
          qt = *reinterpret_cast<const float4*>(a.in_shift + cc);
        }
        const float4 v = finish_a<LD>(a, qa[p][j], qs, qt);
        uint2 h[T];
        split4<T>(v, h);
#pragma unroll
        for (int t = 0; t < T; ++t) *reinterpret_cast<uint2*>(psm + t * PLANE + row * row_bytes + c * 2) = h[t];
    }
  }
  __syncthreads();

  // ---- matrix pipe: k-blocks of 16, W fragments one k-block ahead ----
  floatx16 acc[1][NBW];
#pragma unroll
  for (int j = 0; j < NBW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
  const unsigned char* arow = psm + (wm * 32 + (lane & 31)) * row_bytes + g * 16;
  auto kblock = [&](auto set_tag, const int kb) __attribute__((always_inline)) {
    constexpr int S = decltype(set_tag)::value;
    fetch_w(std::integral_constant<int, S ^ 1>{}, kb + 1);
    bf16x8 av[T];
#pragma unroll
    for (int t = 0; t < T; ++t) av[t] = *reinterpret_cast<const bf16x8*>(arow + t * PLANE + kb * 32);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
      bf16x8 bv[T];
#pragma unroll
      for (int t = 0; t < T; ++t) bv[t] = __builtin_bit_cast(bf16x8, bw[S][j][t]);
      acc[0][j] = mfma_terms<T>(av, bv, acc[0][j]);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
#pragma unroll
  for (int kb = 0; kb < KB16; kb += 2) {
    kblock(S0{}, kb);
    kblock(S1{}, kb + 1);
  }
  __syncthreads();                             // the panel is dead: its first bytes become the statistics scratch of the epilogue,
  float* red = reinterpret_cast<float*>(psm);  // then 4 KB per wave for the epilogue's transposition (the launcher sizes the LDS for both)
  xbf_epilogue<1, NBW>(a, acc, M, m0, 0, mblk, wm, wn, lane, tid, red, red + 4 * 64 * NBW + wave * 1024);
}

static int g_panel_on = [] { const char* e = getenv("TPGSR_XBF_PANEL"); return !(e && e[0] == '0') ? 1 : 0; }();
/* experiment / test switch: 0 sends the 1x1 convolutions back to the tile loop */
extern "C" void tpgsr_panel_set_enabled(int on) { g_panel_on = on ? 1 : 0; }

static long long g_panel_min_m = [] { const char* e = getenv("TPGSR_XBF_PANEL_MIN_M"); return e ? atoll(e) : 32768ll; }();
/* smallest pixel count the panel kernel takes (default 32768 = 512 workgroups; below that the tile loop's column split fills more of the chip) */
extern "C" void tpgsr_panel_set_min_m(long long m) { g_panel_min_m = m < 64 ? 64 : m; }

// K = 192 -> <= 64 columns (the projections' data gradients): measured on MI355X 22.0 us against 20.2 us on the tile loop in x3 arithmetic
// (two 77 KB workgroups per CU, no overlap inside a workgroup), 15.3 us in x2 -- off unless TPGSR_XBF_PANEL_K192=1
static int g_panel_k192 = [] { const char* e = getenv("TPGSR_XBF_PANEL_K192"); return (e && e[0] == '1') ? 1 : 0; }();
extern "C" void tpgsr_panel_set_k192(int on) { g_panel_k192 = on ? 1 : 0; }

// (K / 32, 32-column blocks per wave) pairs instantiated: 64 -> <= 192, 96 -> <= 192, 192 -> <= 64
#define PANEL_LD_CASES(X) X(0) X(1) X(4) X(17)

// 32-column blocks per wave when the shape is this kernel's, else 0
static int panel_takes(const tpgsr_conv_args* a, long long M, int ld) {
  const int T = a->terms;
  if (!g_panel_on || a->KH * a->KW != 1 || a->wt_bf_cin != 0 || a->stride_w > 1 || a->in_dil_w > 1 || a->in_ps || a->pad_h || a->pad_w ||
      a->OH != a->H || a->OW != a->W || M < g_panel_min_m || T < 1 || T > 3)
    return 0;
  const int nb32 = (a->Cout + 31) >> 5, nq8 = a->kp >> 5;
  int nbw = 0;
  if ((nq8 == 2 || nq8 == 3) && nb32 <= 6) nbw = 3;
  else if (nq8 == 6 && nb32 <= 2 && g_panel_k192) nbw = 1;
  else return 0;
  switch (ld) {
#define PANEL_OK(B) case B:
    PANEL_LD_CASES(PANEL_OK)
#undef PANEL_OK
    return nbw;
    default: return 0;
  }
}
/* (launcher-internal, conv_xbf.hip's split-K plan) */
extern "C" int tpgsr_conv_panel_would_take(const tpgsr_conv_args* a, long long M, int ld) { return panel_takes(a, M, ld) > 0 ? 1 : 0; }

// returns 1 when launched, 0 when the shape is not this kernel's, < 0 on error
extern "C" int tpgsr_conv_panel_xbf_launch(const tpgsr_conv_args* a, long long M, int ld, hipStream_t st) {
  const int T = a->terms;
  const int nbw = panel_takes(a, M, ld);
  if (nbw <= 0) return 0;
  const int nq8 = a->kp >> 5;
  size_t lds = (size_t)T * 64 * (nq8 * 64 + 16);
  if (lds < (size_t)(4 * 64 * nbw + 4 * 1024) * 4) lds = (size_t)(4 * 64 * nbw + 4 * 1024) * 4;      // (epilogue scratch: statistics + staging)
  const void* fn = nullptr;
#define PANEL_PICK(B, TT)                                                          \
  fn = nq8 == 2 ? (const void*)conv_panel_xbf_kernel<B, TT, 3, 2>                  \
     : nq8 == 3 ? (const void*)conv_panel_xbf_kernel<B, TT, 3, 3>                  \
                : (const void*)conv_panel_xbf_kernel<B, TT, 1, 6>;
#define PANEL_CASE(B)                       \
  case B:                                   \
    if (T == 1) { PANEL_PICK(B, 1) }        \
    else if (T == 2) { PANEL_PICK(B, 2) }   \
    else { PANEL_PICK(B, 3) }               \
    break;
  switch (ld) {
    PANEL_LD_CASES(PANEL_CASE)
    default: return 0;
  }
#undef PANEL_CASE
#undef PANEL_PICK
  if (lds > 64 * 1024) {   // opt-in to > 64 KB of dynamic LDS, per (kernel, device)
    static std::mutex mu;
    static std::vector<std::pair<const void*, int>> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
      tpgsr_set_error("tpgsr_conv_fwd(panel): hipGetDevice failed");
      return TPGSR_ERR_LAUNCH;
    }
    std::lock_guard<std::mutex> lock(mu);
    bool have = false;
    for (auto& d : done) have = have || (d.first == fn && d.second == dev);
    if (!have) {
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        tpgsr_set_error("tpgsr_conv_fwd(panel): LDS opt-in failed");
        return TPGSR_ERR_LAUNCH;
      }
      done.push_back({fn, dev});
    }
  }
  tpgsr_conv_args args = *a;
  int Mi = (int)M;
  void* params[] = {&args, &Mi};
  if (hipLaunchKernel(fn, dim3((unsigned)cdiv(M, 64)), dim3(256), params, lds, st) != hipSuccess) {
    tpgsr_set_error("tpgsr_conv_fwd(panel): launch failed: %s", hipGetErrorString(hipGetLastError()));
    return TPGSR_ERR_LAUNCH;
  }
  return 1;
}
