// Forward / data gradient of the KH x KW > 1 x 1 stride-1 convolutions over many pixels (the SR trunk's 3x3 64 -> 64 on 16 x 64 maps:
// model/tsrn.py:375-379, 33 launches of a C3 step; the upsample block's 64 -> 256): WHOLE-CU halo kernel.
//
// The halo kernel of conv_xbf.hip runs two 8-wave workgroups per CU, each on a 64-pixel x 64-channel tile with ONE accumulator per
// consumer wave: 768 tiles on 512 resident workgroups are a full round and a half-empty one, a wave's MFMAs are one dependent chain
// (64 cycles per instruction instead of 32), and every wave streams its own weight fragments out of L1 for one 32 x 32 block
// (22 - 25 us per trunk launch for 4.3 us of matrix work, round 3's profile).  Here ONE workgroup owns a CU and works on THREE
// consecutive 64-pixel tiles at once -- 768 tiles = exactly one round on 256 CUs:
//   * consumer wave (wm, wn) holds the SAME 32 x 32 position of all three tiles: three independent accumulator chains per wave,
//     one wave per SIMD, so the matrix pipe issues back to back, and a weight fragment read from L1 feeds three MFMA groups
//     (a third of the L1 bytes per MFMA: the 64 B / clk vector L1 was the binding pipe of the one-block layout);
//   * producers (waves 4..7) load the halo of the 192 output pixels -- every input pixel any tap touches, a contiguous range of the
//     virtual padded index space, 462 entries at most for a 3x3 on a 64-wide map (330 when the three rows stay inside one image) --
//     one 32-channel block at a time, fused prologue + split into bf16 terms on the way into a double-buffered swizzled LDS image;
//     the loads of block c + 1 are in flight while block c is split and while the consumers work on it.
// Arithmetic and summation order per output are those of conv_halo_xbf_kernel (channel block outer, taps inner, term pairs smallest
// first), so the two kernels agree bit for bit.  T <= 2 (three-term planes of 462 entries do not fit 160 KB twice): the fp32-equivalent
// policy stays on the two-workgroup kernel.  Epilogue (bias / activation / BatchNorm statistics / BatchNorm-backward sums): xbf_store_tile.
#include "conv_xbf_common.h"
#include <mutex>
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include <vector>

// diagnostic time line (tpgsr_halo3_trace, as tpgsr_halo_trace of conv_xbf.hip): wall-clock stamps (100 MHz) of the first 8 workgroups,
// [workgroup][8 wave rows][256 slots]; slot 4 j + k of item j -- producers: k = 0 loads issued, 1 split + stored, 2 past the barrier;
// consumers: k = 0 at the barrier, 1 past it, 2 MFMAs issued, 3 tiles stored (last item of a super-tile); slot 255: kernel entry
__device__ unsigned long long* g_halo3_trace = nullptr;
#define H3_STAMP(slot)                                                                                       \
  do {                                                                                                       \
    if (trace && lane == 0 && (slot) < 256)                                                                  \
      __builtin_nontemporal_store((unsigned long long)wall_clock64(), trace + (blockIdx.x * 8 + wave) * 256 + (slot)); \
  } while (0)

#define H3_TM 3                        // 64-pixel tiles per workgroup
#define H3_NE 15                       // halo entries per producer thread (capacity 32 * 15 = 480)
#define H3_PLANE (32 * H3_NE * 64)     // bytes per term plane of a halo buffer: a compile-time pitch, so a term is an immediate offset

// TAPS: KH * KW when the tap loop is fully unrolled (9: every 3x3), 0 = a run-time loop over the taps.  Unrolled because the compiler's
// s_waitcnt insertion merges the counter states of a loop's entry and back edge conservatively: with the two-tap loop every other
// group waited for ALL but its own two W loads (vmcnt(2) where vmcnt(6) was exact), i.e. for fragments requested 288 cycles earlier --
// an L2 round trip exposed per tap, 600 ns per tap for 290 of MFMAs (profiles/r04_halo3_trace.md).
template <int LD, int T, int TAPS>
__global__ __launch_bounds__(512, 2) void conv_halo3_xbf_kernel(tpgsr_conv_args a, int M, int Lcap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];   // [2 buffers][T][Lcap entries][64 B], 2 x 3 x 1 KB `red`, 16 KB staging
  constexpr int PLANE = H3_PLANE, BUF = T * PLANE;
  float* red_base = reinterpret_cast<float*>(hsm + 2 * BUF);
  float* stage_base = red_base + 2 * H3_TM * 256;      // 4 consumer waves x 2 slots x [32 rows][32 columns] fp32: the epilogue's transposition
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  typedef __attribute__((address_space(1))) unsigned long long* gptr_t;     // (a global, not a flat, store: see conv_xbf.hip)
  const gptr_t trace = blockIdx.x < 8 ? (gptr_t)g_halo3_trace : (gptr_t) nullptr;
  H3_STAMP(255);
  const int nbn = (a.Cout + 63) >> 6;
  const int mtiles = (M + 63) >> 6;
  const int nst = ((mtiles + H3_TM - 1) / H3_TM) * nbn;          // super-tiles: (192 pixels) x (64 channels)
  const int taps = a.KH * a.KW, NC = a.Cin >> 5;
  const int Hp = a.OH + a.KH - 1, Wp = a.OW + a.KW - 1, ohw = a.OH * a.OW;
  auto qbase = [&](int m) __attribute__((always_inline)) {
    const int n = m / ohw, r = m - n * ohw, oh = r / a.OW;
    return (n * Hp + oh) * Wp + (r - oh * a.OW);
  };

  if (wave >= 4) {
    // ------------------------------- producers -------------------------------
    // quad aq of the H3_NE CONSECUTIVE entries er * H3_NE ..: walking the virtual padded index space one entry at a time costs one
    // compare per entry (the stride-32 walk of conv_xbf.hip four; this prologue is on the critical path of a one-round kernel)
    const int pt = tid - 256, aq = pt & 7, er = pt >> 3;
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(a.in, (size_t)a.N * a.H * a.W * a.in_ld);
    const __amdgpu_buffer_rsrc_t rs_in2 = make_rsrc(a.in2 ? a.in2 : a.in, (size_t)a.N * a.H * a.W * a.in2_ld);
    int hpix[H3_NE];       // input pixel of entry er * H3_NE + i: >= 0, -1 = padding (stored as zeros), -2 = not part of the halo
    auto decode_tile = [&](const int t) __attribute__((always_inline)) {
      const int m0 = (xcd_remap(t, nst) / nbn) * (64 * H3_TM);
      const int q0 = qbase(m0);
      const int L = qbase(min(m0 + 64 * H3_TM - 1, M - 1)) - q0 + (a.KH - 1) * Wp + a.KW;     // <= Lcap (host bound)
      const int q = q0 + er * H3_NE;
      int n = q / (Hp * Wp);
      const int rem = q - n * (Hp * Wp);
      int r = rem / Wp, sx = rem - r * Wp;
#pragma unroll
      for (int i = 0; i < H3_NE; ++i) {
        const int ih = r - a.pad_h, iw = sx - a.pad_w;
        const bool in = n < a.N && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
        hpix[i] = er * H3_NE + i < L ? (in ? (n * a.H + ih) * a.W + iw : -1) : -2;
        const bool c1 = ++sx >= Wp;
        sx = c1 ? 0 : sx;
        r += c1 ? 1 : 0;
        const bool c2 = r >= Hp;
        r = c2 ? 0 : r;
        n += c2 ? 1 : 0;
      }
    };
    constexpr bool DB = !(LD & 4);       // (the residual-add loader carries two quads per entry: one register set only)
    ARaw hr[DB ? 2 : 1][H3_NE];
    float4 qs[2], qt[2], q2 = make_float4(1.f, 1.f, 1.f, 1.f);      // (q2: per-channel scale of the residual operand, LD bit 32 -- one register set)
    qs[0] = qs[1] = make_float4(1.f, 1.f, 1.f, 1.f);
    qt[0] = qt[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_item = [&](auto set_tag, const int cc) __attribute__((always_inline)) {
      constexpr int S = decltype(set_tag)::value;
      const int c = cc * 32 + aq * 4;
#pragma unroll
      for (int i = 0; i < H3_NE; ++i) {
        const bool ok = hpix[i] >= 0;
        // padding and out-of-halo entries load hardware zeros; only an affine / activation prologue (which maps 0 to f(0)) needs to
        // remember them -- thirty per-lane flags held across a barrier are what spilled the scalar registers of the first version
        hr[S][i].ok = (LD & 3) ? ok : true;
        hr[S][i].raw = false;
        hr[S][i].v = buf_load4(rs_in, ok ? ((unsigned)hpix[i] * (unsigned)a.in_ld + (unsigned)(a.in_coff + c)) * 4u : OOB_OFF);
        if (LD & 4) hr[S][i].v2 = buf_load4(rs_in2, ok ? ((unsigned)hpix[i] * (unsigned)a.in2_ld + (unsigned)c) * 4u : OOB_OFF);
      }
      if (LD & 1) {
        qs[S] = *reinterpret_cast<const float4*>(a.in_scale + c);
        qt[S] = *reinterpret_cast<const float4*>(a.in_shift + c);
      }
      if (LD & 32) q2 = *reinterpret_cast<const float4*>(a.in2_scale + c);
    };
    auto store_item = [&](auto set_tag, const int j) __attribute__((always_inline)) {
      constexpr int S = decltype(set_tag)::value;
      unsigned char* buf = hsm + (j & 1) * BUF;
#pragma unroll
      for (int i = 0; i < H3_NE; ++i) {
        const float4 v = finish_a<LD>(a, hr[S][i], qs[S], qt[S], q2);
        uint2 h[T];
        split4<T>(v, h);
        const int e = er * H3_NE + i;      // every one of the 480 entries of the plane is written (zeros past the halo's end)
        const int off = e * 64 + (((aq >> 1) ^ ((e >> 2) & 3)) << 4) + (aq & 1) * 8;
#pragma unroll
        for (int t = 0; t < T; ++t) *reinterpret_cast<uint2*>(buf + t * PLANE + off) = h[t];
      }
    };
    int t = blockIdx.x, cc = 0, j = 0;
    using P0 = std::integral_constant<int, 0>;
    using PN = std::integral_constant<int, DB ? 1 : 0>;
    decode_tile(t);
    if (DB) load_item(P0{}, 0);
    auto item = [&](auto cur_tag, auto nxt_tag) __attribute__((always_inline)) -> bool {
      int ncc = cc + 1, nt = t;
      if (ncc == NC) {
        ncc = 0;
        nt = t + gridDim.x;
      }
      const bool more = nt < nst;
      if (!DB) load_item(cur_tag, cc);
      if (more && ncc == 0) decode_tile(nt);       // (this item's entries are already captured in its register set)
      if (DB && more) load_item(nxt_tag, ncc);
      H3_STAMP(4 * j);
      store_item(cur_tag, j);
      H3_STAMP(4 * j + 1);
      __syncthreads();      // barrier j: item j is in LDS, and the consumers are done with item j - 1
      H3_STAMP(4 * j + 2);
      ++j;
      t = nt;
      cc = ncc;
      return more;
    };
    while (true) {
      if (!item(P0{}, PN{})) break;
      if (!item(PN{}, P0{})) break;
    }
    __syncthreads();        // the final barrier (the consumers' last statistics flush)
    goto fin_tail;
  }

  {
  // ------------------------------- consumers -------------------------------
  const int wm = wave & 1, wn = wave >> 1;
  const int wrows = a.wt_ld > 0 ? a.wt_ld : a.Cout;
  const int NB32 = (wrows + 31) >> 5, KB16 = a.kp >> 4;
  const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(reinterpret_cast<const float*>(a.wt_bf), (size_t)T * NB32 * 32 * a.kp / 2);
  const unsigned plane_w = (unsigned)NB32 * KB16 * 1024u;
  const int g = lane >> 5;
  unsigned woff = 0;
  int ebase[H3_TM];

  floatx16 acc[H3_TM][1][1];  // the same 32 x 32 position of the three tiles: three independent MFMA chains
  bf16x8 av[2][H3_TM][T];     // [k-block][tile][term]
  u32x4 bw[2][2][T];          // [set][k-block][term]: W fragments one TAP ahead of the matrix pipe
  int gtap = 0;
  auto fetch_w = [&](auto set_tag, auto kb_tag, const int tapidx) __attribute__((always_inline)) {
    constexpr int S = decltype(set_tag)::value, KB = decltype(kb_tag)::value;
    // per-lane part (column block, lane) in the vector offset -- OOB_OFF for a column block past Cout: hardware zeros --, the tap / term
    // part in the SCALAR offset: no address arithmetic in the matrix pipe's shadow, no register per tap
#pragma unroll
    for (int t = 0; t < T; ++t)
      bw[S][KB][t] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)woff, (int)(t * plane_w + (unsigned)(2 * tapidx + KB) * 1024u), 0);
  };
  // A fragments in two steps: the byte addresses of a group's six LDS reads are computed (in the matrix pipe's shadow) one group
  // BEFORE the reads are issued, so a group opens with its reads back to back and they have a whole group to land.  (Reads issued
  // one per MFMA inside the group -- the first version of this pipeline -- left the last of them ~100 cycles before their first
  // use: every group began with a wait, 620 ns per tap measured for 290 of MFMAs.)
  int an[H3_TM];
  auto addr_a = [&](auto kb_tag, const int bufoff, const int tapoff) __attribute__((always_inline)) {
    constexpr int KB = decltype(kb_tag)::value;
#pragma unroll
    for (int m = 0; m < H3_TM; ++m) {
      const int e = ebase[m] + tapoff;
      an[m] = bufoff + e * 64 + (((KB * 2 + g) ^ ((e >> 2) & 3)) << 4);
    }
  };
  auto read_a = [&](auto kb_tag) __attribute__((always_inline)) {
    constexpr int KB = decltype(kb_tag)::value;
#pragma unroll
    for (int m = 0; m < H3_TM; ++m)
#pragma unroll
      for (int t = 0; t < T; ++t) av[KB][m][t] = *reinterpret_cast<const bf16x8*>(hsm + an[m] + t * PLANE);   // (the term: an immediate offset)
  };
  // term-major over the three tiles: consecutive MFMAs go to different accumulators (per accumulator the order is mfma_terms<T>'s)
  auto multiply = [&](auto set_tag, auto kb_tag) __attribute__((always_inline)) {
    constexpr int S = decltype(set_tag)::value, KB = decltype(kb_tag)::value;
    bf16x8 bv[T];
#pragma unroll
    for (int t = 0; t < T; ++t) bv[t] = __builtin_bit_cast(bf16x8, bw[S][KB][t]);
    if (T == 2) {
#pragma unroll
      for (int m = 0; m < H3_TM; ++m) acc[m][0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[KB][m][0], bv[T - 1], acc[m][0][0], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < H3_TM; ++m) acc[m][0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[KB][m][T - 1], bv[0], acc[m][0][0], 0, 0, 0);
    }
#pragma unroll
    for (int m = 0; m < H3_TM; ++m) acc[m][0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[KB][m][0], bv[0], acc[m][0][0], 0, 0, 0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  int kw = 0, tapoff = 0;
  auto next_tap = [&]() __attribute__((always_inline)) {
    const bool cw = ++kw == a.KW;
    tapoff += cw ? Wp - a.KW + 1 : 1;
    kw = cw ? 0 : kw;
    asm volatile("" : "+s"(tapoff));   // opaque: with the taps unrolled the compiler otherwise precomputes every tap's three swizzled
                                       // addresses per super-tile (27 registers; the spill that followed drained the W loads)
  };
  // The consumer's software pipeline, shifted by HALF a tap.  A tap = two groups of 3 (2 T - 1) MFMAs (k-block 0, k-block 1), and
  // every group also issues the loads whose destination registers the PREVIOUS group has just released:
  //   group 0 of tap t (W set S = t & 1):  MFMAs (t, kb0)  |  A(t, kb1) -> av[1],      W(t + 1, kb1) -> bw[S ^ 1][1]
  //   group 1 of tap t:                    MFMAs (t, kb1)  |  A(t + 1, kb0) -> av[0],  W(t + 2, kb0) -> bw[S][0]
  // so an A fragment has one group (288 cycles of matrix work at T = 2) between its LDS read and its use, a W fragment 2.5 groups
  // between its L2 read and its use, and -- the point -- the loads are interleaved WITH the MFMAs by sched_group_barrier instead of
  // sitting between two MFMA bursts: one workgroup per CU means one consumer wave per SIMD, nobody else fills the matrix pipe while a
  // wave issues 20 load / address instructions (the burst form measured 1150 cycles per tap for 576 of MFMAs).
#define H3_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
  // group = [T * 3 LDS reads] [T W loads] then MFMAs with two address instructions in each gap
  // group = MFMAs with the group's loads in the gaps behind the FIRST ones (a burst of eight load instructions in front of the group
  // left the matrix pipe idle for their issue time), address arithmetic behind the rest
  auto interleave = [&](const bool ds) __attribute__((always_inline)) {
    constexpr int NM = 3 * (2 * T - 1);
    int used = 0;
    if (ds) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        H3_SGB(0x008, 1);
        H3_SGB(0x100, T);
      }
      used = 3;
    }
    H3_SGB(0x008, 1);
    H3_SGB(0x020, T);
    ++used;
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      if (i >= NM - used) break;
      H3_SGB(0x008, 1);
      H3_SGB(0x002, 3);
    }
  };
  auto tap = [&](auto set_tag, auto next_tag, const int bufoff) __attribute__((always_inline)) {
    constexpr int S = decltype(set_tag)::value;
    constexpr bool NEXT = decltype(next_tag)::value != 0;
    using SX = std::integral_constant<int, S ^ 1>;
    __builtin_amdgcn_sched_barrier(0);
    read_a(I1{});                               // A(t, kb1): addresses from the previous group
    fetch_w(SX{}, I1{}, gtap + 1);              // W(t + 1, kb1)
    if (NEXT) {
      next_tap();
      addr_a(I0{}, bufoff, tapoff);             // where A(t + 1, kb0) lives
    }
    multiply(set_tag, I0{});
    interleave(true);
    __builtin_amdgcn_sched_barrier(0);
    if (NEXT) read_a(I0{});                     // A(t + 1, kb0)
    fetch_w(set_tag, I0{}, gtap + 2);           // W(t + 2, kb0)
    if (NEXT) addr_a(I1{}, bufoff, tapoff);     // where A(t + 1, kb1) lives
    multiply(set_tag, I1{});
    interleave(NEXT);
    __builtin_amdgcn_sched_barrier(0);
    ++gtap;
  };
  // all taps of one channel block; ODD tap counts only (launcher), so consecutive blocks start on alternating W sets: S0 = block & 1
  auto run_block = [&](auto s0_tag, const int bufoff) __attribute__((always_inline)) {
    constexpr int S0 = decltype(s0_tag)::value;
    using A0 = std::integral_constant<int, S0>;
    using A1 = std::integral_constant<int, S0 ^ 1>;
    kw = 0;
    tapoff = 0;
    addr_a(I0{}, bufoff, 0);
    read_a(I0{});                                // A(first tap, kb0): the one LDS latency a block exposes
    addr_a(I1{}, bufoff, 0);
    if constexpr (TAPS > 0) {
      auto unrolled = [&](auto seq) __attribute__((always_inline)) {
        [&]<int... I>(std::integer_sequence<int, I...>) __attribute__((always_inline)) {
          (tap(std::integral_constant<int, S0 ^ (I & 1)>{}, std::integral_constant<int, (I + 1 < TAPS) ? 1 : 0>{}, bufoff), ...);
        }(seq);
      };
      unrolled(std::make_integer_sequence<int, TAPS>{});
    } else {
      for (int tp = 0; tp + 2 < taps; tp += 2) {
        tap(A0{}, I1{}, bufoff);
        tap(A1{}, I1{}, bufoff);
      }
      tap(A0{}, I0{}, bufoff);
    }
  };
  int j = 0, pend_mblk = -1, pend_n0 = 0, ndone = 0;
  auto flush_pending = [&]() __attribute__((always_inline)) {   // the previous super-tile's BN statistics: behind a barrier now
    if (pend_mblk >= 0) {
      if (a.bn_row_tiles == H3_TM) {
        // ONE row per super-tile (tpgsr_conv_args.bn_row_tiles): the tiles' sums -- each exactly what the per-tile row would hold --
        // added in tile order
        if (tid < 64 && pend_n0 + tid < a.Cout) {
          float v0 = 0.f, v1 = 0.f;
#pragma unroll
          for (int m = 0; m < H3_TM; ++m)
            if ((pend_mblk + m) * 64 < M) {
              const float* red = red_base + (((ndone - 1) & 1) * H3_TM + m) * 256;
              v0 += red[tid] + red[128 + tid];
              v1 += red[64 + tid] + red[192 + tid];
            }
          float* dst = a.bn_partial + (size_t)(pend_mblk / H3_TM) * 2 * a.Cout;
          dst[pend_n0 + tid] = v0;
          dst[a.Cout + pend_n0 + tid] = v1;
        }
      } else {
#pragma unroll
        for (int m = 0; m < H3_TM; ++m)
          if ((pend_mblk + m) * 64 < M)
            xbf_bn_flush<1, 1>(a, M, pend_n0, pend_mblk + m, tid, red_base + (((ndone - 1) & 1) * H3_TM + m) * 256, a.fin_mode != 0);
      }
      pend_mblk = -1;
    }
  };
  for (int st = blockIdx.x; st < nst; st += gridDim.x) {
    const int tile = xcd_remap(st, nst);
    const int mblk3 = tile / nbn;
    const int m0 = mblk3 * (64 * H3_TM), n0 = (tile - mblk3 * nbn) * 64;
    const int q0 = qbase(m0);
    const int ncol0 = n0 + wn * 32;
    woff = ncol0 < a.Cout ? ((unsigned)((a.wt_coff + ncol0) >> 5) * KB16) * 1024u + lane * 16u : OOB_OFF;
#pragma unroll
    for (int m = 0; m < H3_TM; ++m) {
      ebase[m] = qbase(min(m0 + 64 * m + wm * 32 + (lane & 31), M - 1)) - q0;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][0][0][r] = 0.f;
    }
    gtap = 0;
    fetch_w(I0{}, I0{}, 0);
    fetch_w(I0{}, I1{}, 0);
    fetch_w(I1{}, I0{}, 1);
    for (int cc = 0; cc < NC; cc += 2) {
      H3_STAMP(4 * j);
      __syncthreads();        // barrier j
      H3_STAMP(4 * j + 1);
      flush_pending();
      run_block(I0{}, (j & 1) * BUF);
      H3_STAMP(4 * j + 2);
      ++j;
      if (cc + 1 < NC) {
        H3_STAMP(4 * j);
        __syncthreads();      // barrier j
        H3_STAMP(4 * j + 1);
        run_block(I1{}, (j & 1) * BUF);
        H3_STAMP(4 * j + 2);
        ++j;
      }
    }
    // Epilogue.  xbf_store_tile issues 16 four-byte stores per lane and tile (a lane owns one column of 16 rows): 48 per super-tile, and
    // store ISSUE is what bounds it (6.8 us of a 22 us launch in the first version's trace).  Fast path -- plain dense store, no
    // BatchNorm-backward epilogue, 16-byte aligned rows: every wave transposes its 32 x 32 block through 4 KB of LDS of its own
    // (wave-local: LDS operations of one wave execute in order, no barrier) and stores 4 x 16 bytes per lane and tile; bias,
    // activation and the BatchNorm statistics exactly as xbf_store_tile computes them (same values, same summation order).
    const bool bnb = a.bnb_y != nullptr;
    const bool fast = !a.out_ps && (a.Cout & 3) == 0 && (a.out_ld & 3) == 0 && (a.out_coff & 3) == 0 && ((uintptr_t)a.out & 15) == 0 &&
                      (!bnb || ((uintptr_t)a.bnb_y & 15) == 0);
    const bool full3 = m0 + 64 * (H3_TM - 1) < M;
    if (fast && !bnb && a.out_act == TPGSR_ACT_NONE && full3) {
      // the common case (every trunk convolution): three tiles through TWO staging slots per wave, so a tile's 16 LDS writes overlap
      // the previous tile's reads and wide stores (LDS operations of one wave execute in order: slot 0 is rewritten by tile 2 only
      // behind tile 0's reads)
      const int cloc = wn * 32 + (lane & 31), n = n0 + cloc;
      const bool nvalid = n < a.Cout;
      const float bias = (a.bias && nvalid) ? a.bias[n] : 0.f;
      const int quad = lane & 7, nq = n0 + wn * 32 + quad * 4;
      auto stage = [&](auto m_tag, float* stg) __attribute__((always_inline)) {
        constexpr int m = decltype(m_tag)::value;
        float sum = 0.f, sq = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float raw = acc[m][0][0][r];
          if (nvalid) {
            sum += raw;
            sq = __builtin_fmaf(raw, raw, sq);
          }
          stg[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = raw + bias;
        }
        if (a.bn_partial) {
          float* red = red_base + ((ndone & 1) * H3_TM + m) * 256;
          sum += __shfl_xor(sum, 32);
          sq += __shfl_xor(sq, 32);
          if (lane < 32) {
            red[(wm * 2 + 0) * 64 + cloc] = sum;
            red[(wm * 2 + 1) * 64 + cloc] = sq;
          }
        }
      };
      auto drain = [&](auto m_tag, const float* stg) __attribute__((always_inline)) {
        constexpr int m = decltype(m_tag)::value;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int row = 8 * k + (lane >> 3);
          const float4 q4 = *reinterpret_cast<const float4*>(stg + row * 32 + quad * 4);
          if (nq < a.Cout) *reinterpret_cast<float4*>(a.out + (size_t)(m0 + 64 * m + wm * 32 + row) * a.out_ld + a.out_coff + nq) = q4;
        }
      };
      float* s0 = stage_base + wave * 2048;
      float* s1 = s0 + 1024;
      using M0 = std::integral_constant<int, 0>;
      using M1 = std::integral_constant<int, 1>;
      using M2 = std::integral_constant<int, 2>;
      stage(M0{}, s0);
      stage(M1{}, s1);
      drain(M0{}, s0);
      stage(M2{}, s0);
      drain(M1{}, s1);
      drain(M2{}, s0);
    } else {
    // ONE copy of the general epilogue code, looped over the three tiles (the accumulator is selected: 32 moves): unrolled three times,
    // with both paths and the run-time activation switch, the epilogue was 20 000 of the kernel's 25 000 instructions -- and a
    // one-round kernel runs every instruction out of a cold instruction cache
    // BatchNorm-backward epilogue (every data gradient of the trunk): the BatchNorm input y comes in as 4 x 16 bytes per lane and tile
    // (one tile ahead), goes through the wave's staging block the other way round -- written row-major, read back in accumulator
    // order -- and the gradient leaves as in the plain case; xbf_bnb_elem is the arithmetic of xbf_store_tile, element for element
    const int cloc = wn * 32 + (lane & 31), n = n0 + cloc;
    const bool nvalid = n < a.Cout;
    const float bias = (a.bias && nvalid) ? a.bias[n] : 0.f;
    const int quad = lane & 7, nq = n0 + wn * 32 + quad * 4;
    float b_mu = 0.f, b_rs = 0.f, b_sc = 1.f, b_sh = 0.f;
    if (bnb && nvalid) {
      if (a.bn_partial) {
        b_mu = a.bnb_mean[n];
        b_rs = a.bnb_rstd[n];
      }
      if (a.bnb_act && a.bnb_scale) {
        b_sc = a.bnb_scale[n];
        b_sh = a.bnb_shift[n];
      }
    }
    float4 yq[4];
    auto load_y = [&](const int m, float4 (&dst)[4]) __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int mm = m0 + 64 * m + wm * 32 + 8 * k + (lane >> 3);
        dst[k] = (mm < M && nq < a.Cout) ? *reinterpret_cast<const float4*>(a.bnb_y + (size_t)mm * a.Cout + nq) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    if (bnb && fast) load_y(0, yq);
#pragma clang loop unroll(disable)
    for (int m = 0; m < H3_TM; ++m) {
      if (m0 + 64 * m >= M) break;        // (uniform) a ragged last super-tile: tiles past the end have nothing to store
      floatx16 tile[1][1];
#pragma unroll
      for (int r = 0; r < 16; ++r) tile[0][0][r] = m == 0 ? acc[0][0][0][r] : m == 1 ? acc[1][0][0][r] : acc[2][0][0][r];
      float* red = red_base + ((ndone & 1) * H3_TM + m) * 256;
      if (!fast) {
        xbf_store_tile<1, 1>(a, tile, M, m0 + 64 * m, n0, wm, wn, lane, red);
        continue;
      }
      float* stg = stage_base + wave * 2048;
      float yv[16];
      float4 ynx[4];
      if (bnb) {
        if (m + 1 < H3_TM) load_y(m + 1, ynx);      // (rows past the end load nothing)
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<float4*>(stg + (8 * k + (lane >> 3)) * 32 + quad * 4) = yq[k];
#pragma unroll
        for (int r = 0; r < 16; ++r) yv[r] = stg[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)];
      }
      float sum = 0.f, sq = 0.f;
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int mm = m0 + 64 * m + wm * 32 + row;
        float raw = tile[0][0][r];
        if (mm < M && nvalid) {
          if (bnb) {
            const float dz = xbf_bnb_elem(raw, yv[r], b_sc, b_sh, b_mu, b_rs, a.bnb_act, sum, sq);
            if (a.bnb_store_dz) raw = dz;
          } else {
            sum += raw;
            sq = __builtin_fmaf(raw, raw, sq);
          }
        }
        v[r] = raw + bias;
      }
      if (a.out_act != TPGSR_ACT_NONE) {      // (uniform; the trunk's convolutions have none)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = apply_act(v[r], a.out_act);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = v[r];
      if (a.bn_partial) {
        sum += __shfl_xor(sum, 32);
        sq += __shfl_xor(sq, 32);
        if (lane < 32) {
          red[(wm * 2 + 0) * 64 + cloc] = sum;
          red[(wm * 2 + 1) * 64 + cloc] = sq;
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int row = 8 * k + (lane >> 3);
        const int mm = m0 + 64 * m + wm * 32 + row;
        const float4 q4 = *reinterpret_cast<const float4*>(stg + row * 32 + quad * 4);
        if (mm < M && nq < a.Cout) *reinterpret_cast<float4*>(a.out + (size_t)mm * a.out_ld + a.out_coff + nq) = q4;
      }
      if (bnb) {
#pragma unroll
        for (int k = 0; k < 4; ++k) yq[k] = ynx[k];
      }
    }
    }
    H3_STAMP(4 * (j - 1) + 3);
    if (a.bn_partial) {
      pend_mblk = mblk3 * H3_TM;
      pend_n0 = n0;
    }
    ++ndone;
  }
  __syncthreads();            // the final barrier
  flush_pending();
  }
fin_tail:
  // BatchNorm finalize by the LAST workgroup (tpgsr_conv_args.fin_mode): this workgroup's partial rows have left as write-through
  // stores; once they are acknowledged it draws a ticket, and whoever draws the last one reduces all rows (conv_xbf_common.h)
  if (a.fin_mode) {
    __shared__ int s_ticket;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_ticket = __hip_atomic_fetch_add(a.fin_counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket == (int)gridDim.x - 1) {
      xbf_fin_last(a, M, tid, 512, reinterpret_cast<double*>(hsm));
      if (tid == 0) __hip_atomic_store(a.fin_counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

/* upper bound of the halo length of any run of 64 * H3_TM consecutive output pixels (cf. halo_capacity in conv_xbf.hip) */
static int halo3_capacity(const tpgsr_conv_args* a) {
  const int P = 64 * H3_TM, Wp = a->OW + a->KW - 1, ohw = a->OH * a->OW;
  const int row_wraps = (a->OW % P == 0) ? 0 : (P - 1) / a->OW + 1;
  const int img_wraps = (ohw % P == 0) ? 0 : (P - 1) / ohw + 1;
  return P - 1 + row_wraps * (a->KW - 1) + img_wraps * (a->KH - 1) * Wp + (a->KH - 1) * Wp + a->KW;
}

extern "C" void tpgsr_conv_fin_fused_mark(void);      // conv_mfma.hip

extern "C" int tpgsr_halo3_trace(unsigned long long* buf) {   // buf: 8 * 8 * 256 uint64 of device memory, or nullptr to switch off
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_halo3_trace), &buf, sizeof(buf)) != hipSuccess) {
    tpgsr_set_error("tpgsr_halo3_trace: hipMemcpyToSymbol failed");
    return TPGSR_ERR_LAUNCH;
  }
  return 0;
}

static int g_h3_on = [] { const char* e = getenv("TPGSR_XBF_HALO3"); return (e && e[0] == '0') ? 0 : 1; }();
/* experiment / test switch: 0 sends every convolution back to the two-workgroup halo kernel */
extern "C" void tpgsr_halo3_set_enabled(int on) { g_h3_on = on ? 1 : 0; }

#define H3_LD_CASES(X) X(0) X(1) X(2) X(3) X(4) X(5) X(7) X(37)

// > 0 (the halo capacity) when the shape is this kernel's, else 0
static int halo3_takes(const tpgsr_conv_args* a, long long M, int ld) {
  const int T = a->terms;
  if (!g_h3_on || T < 1 || T > 2 || a->KH * a->KW < 3 || !((a->KH * a->KW) & 1) || a->wt_bf_cin != a->Cin || (a->Cin & 31) || a->stride_w > 1 || a->in_dil_w > 1 || a->in_b ||
      a->in_ps || ((ld & ~7) && ld != 37) || ld == 6 || a->OW + a->KW - 1 < 8)
    return 0;
  // the residual-add loader carries two quads per entry and has ONE register set (no load of the next block in flight), and a
  // pixel-shuffled store goes out four bytes at a time: with both (the up-sampling convolution: 102 us here, 87 us there) the
  // two-workgroup kernel, whose second workgroup covers those waits, is faster
  if ((ld & 4) && a->out_ps) return 0;
  const int Lcap = halo3_capacity(a);
  const size_t lds = (size_t)2 * T * H3_PLANE + 2 * H3_TM * 1024 + 8 * 4096;      // halo buffers + statistics scratch + epilogue staging (161 792 B at T = 2)
  if (Lcap > 32 * H3_NE || lds > 163840) return 0;
  // one round of the chip (or several full ones): with fewer super-tiles than CUs the two-workgroup kernel spreads the work better
  const long long nst = (long long)cdiv(cdiv(M, 64), H3_TM) * cdiv(a->Cout, 64);
  static const long long min_st = [] { const char* e = getenv("TPGSR_XBF_HALO3_MIN"); return e ? atoll(e) : 192ll; }();
  if (nst < min_st) return 0;
  switch (ld) {
#define H3_OK(B) case B:
    H3_LD_CASES(H3_OK)
#undef H3_OK
    return Lcap;
    default: return 0;
  }
}

/* (launcher-internal, conv_xbf.hip's split-K plan) does this kernel take the launch? */
extern "C" int tpgsr_conv_halo3_would_take(const tpgsr_conv_args* a, long long M, int ld) { return halo3_takes(a, M, ld) > 0 ? 1 : 0; }

/* 1 when a launch with a scaled residual operand (tpgsr_conv_args.in2_scale) is this kernel's -- the only one whose loader has it */
extern "C" int tpgsr_conv_in2_scale_ok(const tpgsr_conv_args* a) {
  if (!a || !(a->terms > 0 && a->wt_bf && (a->Cin & 3) == 0 && (a->wt_coff & 31) == 0) || a->fin_mode || !a->in2 || !a->in_scale || a->in_act || a->in_b) return 0;
  return halo3_takes(a, (long long)a->N * a->OH * a->OW, 37) > 0 ? 1 : 0;
}

/* tpgsr_conv_args.bn_row_tiles: 3 when tpgsr_conv_fwd(a) lands here (the dispatch of conv_mfma.hip / conv_xbf.hip up to this kernel) */
extern "C" int tpgsr_conv_bn_row_tiles(const tpgsr_conv_args* a) {
  if (!a || !(a->terms > 0 && a->wt_bf && (a->Cin & 3) == 0 && (a->wt_coff & 31) == 0) || a->fin_mode) return 1;
  const int ld = (a->in_scale ? 1 : 0) | (a->in_act ? 2 : 0) | (a->in2 ? 4 : 0) | (a->in_ps ? 8 : 0) | (a->in_b ? 16 : 0);
  return halo3_takes(a, (long long)a->N * a->OH * a->OW, ld) > 0 ? H3_TM : 1;
}

// returns 1 when launched, 0 when the shape is not this kernel's, < 0 on error
extern "C" int tpgsr_conv_halo3_xbf_launch(const tpgsr_conv_args* a, long long M, int ld, hipStream_t st) {
  const int T = a->terms;
  const int Lcap = halo3_takes(a, M, ld);
  if (Lcap <= 0) return 0;
  TPGSR_CHECK_ARG(a->bn_row_tiles == 0 || a->bn_row_tiles == 1 || (a->bn_row_tiles == H3_TM && !a->fin_mode),
                  "tpgsr_conv_fwd(halo3): bn_row_tiles %d (0, 1 or %d without fin_mode)", a->bn_row_tiles, H3_TM);
  const size_t lds = (size_t)2 * T * H3_PLANE + 2 * H3_TM * 1024 + 8 * 4096;
  const long long nst = (long long)cdiv(cdiv(M, 64), H3_TM) * cdiv(a->Cout, 64);
  const void* fn = nullptr;
  const bool t9 = a->KH * a->KW == 9;
#define H3_CASE(B)                                                                                                                 \
  case B:                                                                                                                          \
    fn = T == 1 ? (t9 ? (const void*)conv_halo3_xbf_kernel<B, 1, 9> : (const void*)conv_halo3_xbf_kernel<B, 1, 0>)                 \
                : (t9 ? (const void*)conv_halo3_xbf_kernel<B, 2, 9> : (const void*)conv_halo3_xbf_kernel<B, 2, 0>);                \
    break;
  switch (ld) {
    H3_LD_CASES(H3_CASE)
    default: return 0;
  }
#undef H3_CASE
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) {
    tpgsr_set_error("tpgsr_conv_fwd(halo3): device query failed");
    return TPGSR_ERR_LAUNCH;
  }
  {   // opt-in to > 64 KB of dynamic LDS, per (kernel, device): raised to the largest size seen so far
    static std::mutex mu;
    static std::vector<std::pair<std::pair<const void*, int>, size_t>> done;
    std::lock_guard<std::mutex> lock(mu);
    size_t* cur = nullptr;
    for (auto& d : done)
      if (d.first.first == fn && d.first.second == dev) cur = &d.second;
    if (!cur || *cur < lds) {
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        tpgsr_set_error("tpgsr_conv_fwd(halo3): LDS opt-in (%zu bytes) failed", lds);
        return TPGSR_ERR_LAUNCH;
      }
      if (cur) *cur = lds; else done.push_back({{fn, dev}, lds});
    }
  }
  dim3 grid((unsigned)(nst < cus ? nst : cus));      // persistent: one workgroup per CU
  int Mi = (int)M, Lc = Lcap;
  tpgsr_conv_args args = *a;
  void* params[] = {&args, &Mi, &Lc};
  if (hipLaunchKernel(fn, grid, dim3(512), params, lds, st) != hipSuccess) {
    tpgsr_set_error("tpgsr_conv_fwd(halo3): launch failed: %s", hipGetErrorString(hipGetLastError()));
    return TPGSR_ERR_LAUNCH;
  }
  if (a->fin_mode) tpgsr_conv_fin_fused_mark();      // this kernel's last workgroup finalizes the BatchNorm (conv_mfma.hip: no extra launch)
  return 1;
}
