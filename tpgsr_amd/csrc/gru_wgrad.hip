// ALL weight gradients of one GruBlock (model/tsrn.py:491-508: 1x1 conv -> BiGRU, hidden 32) in ONE launch.
//
// What the backward pass of a GruBlock needs after back-propagation through time (gru.hip) has produced, per pixel p,
//   dgi [p][192]  = (dr_pre, dz_pre, dn_pre) of both directions      (gradient of the input projections gi = loader(x) Wc^T + bc)
//   dghn[p][64]   = dn_pre * r of both directions                      (the one plane where the hidden-side gradient differs from dgi)
// are four contractions over the pixels:
//   dWc [k][n]    = sum_p loader(x)[p][k] dgi[p][n]                    k < Cin (64 | 96), n < 192     (composed operand Wc = W_ih W_1)
//   dbc [n]       = sum_p dgi[p][n]
//   dWhh[d][k][n'] = sum_p h_prev_d[p][k] dgh_d[p][n'],  dgh_d = (dgi r, dgi z, dghn) of direction d,  k < 32, n' < 96
//   dbhh[d][n']   = sum_p dgh_d[p][n']
// Until round 4 these were three tile-loop weight-gradient launches per block (Cin -> 192: 22 / 33 us; 2 x 32 -> 96: 16 us each for
// 0.3 GFLOP -- the 20 hidden-side launches of a step sat at 2 % of their roof), each re-reading dgi / a full [P][192] dgh.
// Here a workgroup owns a range of pixels and walks it in chunks of 32: every operand of the chunk -- loader(x), h_prev of both
// directions (h shifted one step against the scan direction, zero at the sequence start), dgi, dghn -- is loaded ONCE (16-byte
// buffer loads, hardware zero fill), split into bf16 terms and stored pixel-major into one LDS image [32 pixels][416 channels];
// six waves, one per 32-column block of dgi, contract over the pixels with transposing fragment reads (ds_read_b64_tr_b16):
//   wave w = (direction d = w / 3, gate g = w % 3):   Cin / 32 tiles of dWc + 1 tile of dWhh[d], B operand = its dgi block
//   (g = 2: dghn for the hidden tile).  The launch is bound by HBM (82 MB in at batch 48: loader(x) + h + dgi + dghn once).
// Bias gradients: fp32 column sums kept by the staging threads, combined in a fixed order at the end.
// Output: per workgroup z one slab per weight (partC [Z][Cin][192], partH [2][Z][32][96], dbC [Z][192], dbH [2][Z][96]) in the layout
// tpgsr_wgrad_reduce(_program) sums -- few workgroups (Z = 128 by default), so 12.5 MB of slabs per block instead of 17.
#include "conv_xbf_common.h"
#include <mutex>
#include <stdlib.h>

#define GW_PITCH 832                    // bytes per pixel row of an LDS plane (416 bf16); 832 = 3 * 256 + 64: four consecutive rows sit in
#define GW_ROWS 32                      //   disjoint 64-byte bank windows (the transposing reads touch 4 rows x 64 bytes per half wave)
#define GW_PLANE (GW_ROWS * GW_PITCH)
#define GW_HP 96                        // channel offsets inside a row: loader(x) 0..95 | h_prev (dir 0, dir 1) 96..159 |
#define GW_GI 160                       //   dgi 160..351 |
#define GW_GN 352                       //   dghn 352..415

__device__ __forceinline__ bf16x8 gw_frag(const unsigned char* plane, int lane, int col0, int mb) {
  const int G = lane >> 4, q = lane & 15;
  const int mbase = mb * 16 + (G >> 1) * 8;
  const int c0 = col0 + (G & 1) * 16 + (q & 3) * 4;
  const unsigned char* p = plane + (mbase + (q >> 2)) * GW_PITCH + c0 * 2;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + 4 * GW_PITCH));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

template <int LD, int T, int KB>
__global__ __launch_bounds__(384) void gru_wgrad_kernel(tpgsr_gru_wgrad_args w, int P, int MB) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];     // [T planes][32 rows][GW_PITCH]
  const tpgsr_conv_args& a = w.c;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int z = blockIdx.x, Z = gridDim.x;
  const int mbeg = z * MB, mend = min(P, mbeg + MB);
  constexpr int QA = KB * 8;                         // quads of loader(x) per pixel

  // ---- staging roles (fixed per thread) ----
  const bool doA = tid < 16 * QA;                    // KB = 3: everybody; KB = 2: threads 0..255
  const int a_cq = doA ? tid % QA : 0, a_r0 = doA ? tid / QA : 0;            // rows a_r0, a_r0 + 16
  const bool doH = tid < 256;                        // h_prev: quad cq of rows r0, r0 + 16
  const int h_cq = tid & 15, h_r0 = (tid >> 4) & 15;
  const bool doN = tid >= 128;                       // dghn: threads 128..383, same map
  const int n_cq = (tid - 128) & 15, n_r0 = ((tid - 128) >> 4) & 15;
  const int g_cq = tid % 48, g_r0 = tid / 48;        // dgi: rows g_r0 + 8 i, i < 4

  const size_t in_floats = (size_t)a.N * a.H * a.W * a.in_ld;
  const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(a.in, in_floats);
  const __amdgpu_buffer_rsrc_t rs_in2 = (LD & 16) ? make_rsrc(a.in_b, (size_t)a.N * a.W * a.in_b_ld)
                                                  : make_rsrc(a.in2 ? a.in2 : a.in, (size_t)a.N * a.H * a.W * a.in2_ld);
  const __amdgpu_buffer_rsrc_t rs_h = make_rsrc(w.h, (size_t)P * 64);
  const __amdgpu_buffer_rsrc_t rs_gi = make_rsrc(w.dgi, (size_t)P * 192);
  const __amdgpu_buffer_rsrc_t rs_gn = make_rsrc(w.dghn, (size_t)P * 64);
  float4 qs = make_float4(1.f, 1.f, 1.f, 1.f), qt = make_float4(0.f, 0.f, 0.f, 0.f);
  if ((LD & 1) && doA) {
    qs = *reinterpret_cast<const float4*>(a.in_scale + a_cq * 4);
    qt = *reinterpret_cast<const float4*>(a.in_shift + a_cq * 4);
  }
  const KPos kp = {0, 0, a_cq * 4};
  const int hd = h_cq >> 3;                          // direction of this thread's h_prev quad
  const int hstep = (w.axis == 0 ? 1 : a.W) * (hd == 0 ? -1 : 1);            // pixel offset of the previous state in scan order

  ARaw ra[2];
  float4 rh[2], rn[2], rg[4];
  float4 dbq = make_float4(0.f, 0.f, 0.f, 0.f), dbn = make_float4(0.f, 0.f, 0.f, 0.f);
  auto load_chunk = [&](const int mc) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (doA) {
        const PixelPos p = decode_pixel(a, mc + a_r0 + 16 * i, mend);
        ra[i] = load_a_raw<LD>(a, rs_in, rs_in2, p, kp);
      }
      if (doH) {
        const int m = mc + h_r0 + 16 * i;
        const PixelPos p = decode_pixel(a, m, mend);
        const int pos = w.axis == 0 ? p.ow : p.oh, last = (w.axis == 0 ? a.W : a.H) - 1;
        const bool ok = p.valid && (hd == 0 ? pos > 0 : pos < last);
        rh[i] = buf_load4(rs_h, ok ? ((unsigned)(m + hstep) * 64u + (unsigned)(h_cq * 4)) * 4u : OOB_OFF);
      }
      if (doN) {
        const int m = mc + n_r0 + 16 * i;
        rn[i] = buf_load4(rs_gn, m < mend ? ((unsigned)m * 64u + (unsigned)(n_cq * 4)) * 4u : OOB_OFF);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = mc + g_r0 + 8 * i;
      rg[i] = buf_load4(rs_gi, m < mend ? ((unsigned)m * 192u + (unsigned)(g_cq * 4)) * 4u : OOB_OFF);
    }
  };
  auto put = [&](const float4& v, const int row, const int col) __attribute__((always_inline)) {
    uint2 hsp[T];
    split4<T>(v, hsp);
#pragma unroll
    for (int t = 0; t < T; ++t) *reinterpret_cast<uint2*>(gsm + t * GW_PLANE + row * GW_PITCH + col * 2) = hsp[t];
  };
  auto store_chunk = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (doA) put(finish_a<LD>(a, ra[i], qs, qt), a_r0 + 16 * i, a_cq * 4);
      if (doH) put(rh[i], h_r0 + 16 * i, GW_HP + h_cq * 4);
      if (doN) {
        put(rn[i], n_r0 + 16 * i, GW_GN + n_cq * 4);
        dbn.x += rn[i].x; dbn.y += rn[i].y; dbn.z += rn[i].z; dbn.w += rn[i].w;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      put(rg[i], g_r0 + 8 * i, GW_GI + g_cq * 4);
      dbq.x += rg[i].x; dbq.y += rg[i].y; dbq.z += rg[i].z; dbq.w += rg[i].w;
    }
  };

  // ---- consumers: wave = 32-column block of dgi = (direction, gate) ----
  const int d = wave / 3, g = wave - 3 * d;
  floatx16 acc[KB], acch;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    acch[r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) acc[kb][r] = 0.f;
  }

  if (mbeg < mend) {
    load_chunk(mbeg);
    store_chunk();
  }
  __syncthreads();
  for (int mc = mbeg; mc < mend; mc += GW_ROWS) {
    const bool more = mc + GW_ROWS < mend;
    if (more) load_chunk(mc + GW_ROWS);              // in flight while the matrix pipe works on this chunk
#pragma unroll
    for (int mb = 0; mb < GW_ROWS / 16; ++mb) {
      bf16x8 by[T], byh[T], av[T];
#pragma unroll
      for (int t = 0; t < T; ++t) by[t] = gw_frag(gsm + t * GW_PLANE, lane, GW_GI + 32 * wave, mb);
      if (g == 2) {                                  // (wave-uniform) the n gate's hidden-side gradient is dn_pre * r
#pragma unroll
        for (int t = 0; t < T; ++t) byh[t] = gw_frag(gsm + t * GW_PLANE, lane, GW_GN + 32 * d, mb);
      } else {
#pragma unroll
        for (int t = 0; t < T; ++t) byh[t] = by[t];
      }
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
        for (int t = 0; t < T; ++t) av[t] = gw_frag(gsm + t * GW_PLANE, lane, 32 * kb, mb);
        acc[kb] = mfma_terms<T>(av, by, acc[kb]);
      }
#pragma unroll
      for (int t = 0; t < T; ++t) av[t] = gw_frag(gsm + t * GW_PLANE, lane, GW_HP + 32 * d, mb);
      acch = mfma_terms<T>(av, byh, acch);
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                 // everybody is done reading this chunk's image
    if (more) {
      store_chunk();
      __syncthreads();
    }
  }

  // ---- this workgroup's slabs ----
  const int col = lane & 31;
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    float* dst = w.partC + ((size_t)z * a.Cin + 32 * kb) * 192 + 32 * wave + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[(size_t)((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 192] = acc[kb][r];
  }
  {
    float* dst = w.partH + ((size_t)d * Z + z) * 32 * 96 + 32 * g + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[(size_t)((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 96] = acch[r];
  }
  // bias gradients: the staging threads' column sums, added in row-lane order (the image is dead: behind the loop's last barrier)
  float* red = reinterpret_cast<float*>(gsm);        // [8][192]
  float* redn = red + 8 * 192;                       // [16][64]
  *reinterpret_cast<float4*>(red + g_r0 * 192 + g_cq * 4) = dbq;
  if (doN) *reinterpret_cast<float4*>(redn + n_r0 * 64 + n_cq * 4) = dbn;
  __syncthreads();
  if (tid < 192) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) s += red[r * 192 + tid];
    w.dbC[(size_t)z * 192 + tid] = s;
    const int dd = tid / 96, c = tid - 96 * dd;
    if (c < 64) w.dbH[((size_t)dd * Z + z) * 96 + c] = s;          // the r and z columns of dgh are dgi's
  } else if (tid < 256) {
    const int c = tid - 192;
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += redn[r * 64 + c];
    w.dbH[((size_t)(c >> 5) * Z + z) * 96 + 64 + (c & 31)] = s;
  }
}

static int g_gw_z = [] { const char* e = getenv("TPGSR_GRU_WGRAD_Z"); const int v = e ? atoi(e) : 128; return v > 0 ? v : 128; }();

/* number of pixel splits (= workgroups = slabs per weight) tpgsr_gru_wgrad uses for P pixels: the caller sizes the slabs with it */
extern "C" int tpgsr_gru_wgrad_splits(long long P) {
  long long z = g_gw_z;
  const long long maxz = (P + 63) / 64;              // at least two chunks per workgroup
  if (z > maxz) z = maxz;
  if (z < 1) z = 1;
  const long long mb = ((P + z - 1) / z + GW_ROWS - 1) / GW_ROWS * GW_ROWS;
  return (int)((P + mb - 1) / mb);
}

extern "C" int tpgsr_gru_wgrad(const tpgsr_gru_wgrad_args* w, void* stream) {
  TPGSR_CHECK_ARG(w != nullptr, "tpgsr_gru_wgrad: null args");
  const tpgsr_conv_args* a = &w->c;
  TPGSR_CHECK_ARG(a->in && w->dgi && w->dghn && w->h && w->partC && w->partH && w->dbC && w->dbH, "tpgsr_gru_wgrad: null pointer");
  TPGSR_CHECK_ARG(a->N > 0 && a->H > 0 && a->W > 0 && (a->Cin == 64 || a->Cin == 96) && a->Cout == 192 && a->KH == 1 && a->KW == 1 &&
                      a->pad_h == 0 && a->pad_w == 0 && a->OH == a->H && a->OW == a->W && (w->axis == 0 || w->axis == 1),
                  "tpgsr_gru_wgrad: geometry (Cin %d must be 64 | 96, Cout %d must be 192, 1x1, axis %d)", a->Cin, a->Cout, w->axis);
  TPGSR_CHECK_ARG(a->terms >= 1 && a->terms <= 3, "tpgsr_gru_wgrad: runs on the split-bf16 matrix-core path only (terms %d)", a->terms);
  TPGSR_CHECK_ARG(!a->in_ps && !a->in_act && a->in_dil_w <= 1 && a->stride_w <= 1 && (a->in_ld & 3) == 0 && (a->in_coff & 3) == 0 &&
                      ((uintptr_t)a->in & 15) == 0, "tpgsr_gru_wgrad: unsupported loader (pixel shuffle / activation / dilation / unaligned rows)");
  TPGSR_CHECK_ARG((a->in_scale == nullptr) == (a->in_shift == nullptr), "tpgsr_gru_wgrad: in_scale / in_shift must come together");
  if (a->in_b)
    TPGSR_CHECK_ARG(a->cin_a > 0 && a->cin_a < a->Cin && (a->cin_a & 3) == 0 && a->in_b_ld >= a->Cin - a->cin_a && (a->in_b_ld & 3) == 0 && !a->in2,
                    "tpgsr_gru_wgrad: bad concat description");
  TPGSR_CHECK_ARG(a->in_ld >= (a->in_b ? a->cin_a : a->Cin) + a->in_coff, "tpgsr_gru_wgrad: in_ld too small");
  if (a->in2) TPGSR_CHECK_ARG(a->in2_ld >= a->Cin && (a->in2_ld & 3) == 0, "tpgsr_gru_wgrad: bad in2_ld");
  const long long P = (long long)a->N * a->H * a->W;
  TPGSR_CHECK_ARG(P * 192 * 4 <= 0x7fffffffll && P * a->in_ld * 4 <= 0x7fffffffll, "tpgsr_gru_wgrad: operand exceeds the 2 GiB buffer-addressing window (%lld pixels)", P);
  const int Z = w->zsplits;
  TPGSR_CHECK_ARG(Z > 0, "tpgsr_gru_wgrad: zsplits must be the caller's tpgsr_gru_wgrad_splits(P)");
  const int MB = (int)(((P + Z - 1) / Z + GW_ROWS - 1) / GW_ROWS * GW_ROWS);
  TPGSR_CHECK_ARG((long long)MB * Z >= P, "tpgsr_gru_wgrad: %d splits do not cover %lld pixels", Z, P);
  const int ld = (a->in_scale ? 1 : 0) | (a->in2 ? 4 : 0) | (a->in_b ? 16 : 0);
  const int T = a->terms, KB = a->Cin / 32;
  const void* fn = nullptr;
#define GW_PICK(L, K_)                                                                                        \
  fn = T == 1 ? (const void*)gru_wgrad_kernel<L, 1, K_> : T == 2 ? (const void*)gru_wgrad_kernel<L, 2, K_> \
                                                                   : (const void*)gru_wgrad_kernel<L, 3, K_>;
  if (ld == 0 && KB == 2) { GW_PICK(0, 2) }
  else if (ld == 1 && KB == 2) { GW_PICK(1, 2) }
  else if (ld == 4 && KB == 2) { GW_PICK(4, 2) }
  else if (ld == 17 && KB == 3) { GW_PICK(17, 3) }
  else if (ld == 0 && KB == 3) { GW_PICK(0, 3) }
#undef GW_PICK
  TPGSR_CHECK_ARG(fn != nullptr, "tpgsr_gru_wgrad: no kernel for loader %d with %d input channels", ld, a->Cin);
  const size_t lds = (size_t)T * GW_PLANE;
  if (lds > 64 * 1024) {   // opt in to > 64 KB of dynamic LDS, once per (kernel, device)
    static std::mutex mu;
    static const void* done[64][8];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
      tpgsr_set_error("tpgsr_gru_wgrad: hipGetDevice failed");
      return TPGSR_ERR_LAUNCH;
    }
    std::lock_guard<std::mutex> lock(mu);
    bool seen = false;
    int free_slot = -1;
    for (int i = 0; i < 8; ++i) {
      if (done[dev][i] == fn) seen = true;
      if (!done[dev][i] && free_slot < 0) free_slot = i;
    }
    if (!seen) {
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        tpgsr_set_error("tpgsr_gru_wgrad: LDS opt-in (%zu bytes) failed", lds);
        return TPGSR_ERR_LAUNCH;
      }
      if (free_slot >= 0) done[dev][free_slot] = fn;
    }
  }
  tpgsr_gru_wgrad_args args = *w;
  int Pi = (int)P, MBi = MB;
  void* params[] = {&args, &Pi, &MBi};
  if (hipLaunchKernel(fn, dim3(Z), dim3(384), params, lds, (hipStream_t)stream) != hipSuccess) {
    tpgsr_set_error("tpgsr_gru_wgrad: launch failed: %s", hipGetErrorString(hipGetLastError()));
    return TPGSR_ERR_LAUNCH;
  }
  return 0;
}
