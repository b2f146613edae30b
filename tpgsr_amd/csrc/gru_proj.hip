// GruBlock forward in ONE launch (round 5): the composed input projection gi = loader(x) Wc^T + bc
// (model/tsrn.py:495-496: the 1x1 convolution and nn.GRU's W_ih back to back, tpgsr_amd.engine.GruLayer) is computed by the wave that owns
// the sequence, on the matrix cores, straight into LDS -- and the bidirectional scan of gru.hip runs from there.  Until round 4 the
// projection was a launch of its own (the row-panel kernel, conv_panel.hip) that wrote gi [P][192] to HBM (37.7 MB at bs 48) for the
// scan to read back through a prefetch ring: 10 launches and 0.75 GB per forward pass that need not exist.
//
// One workgroup (four waves) = 64 time steps: ONE sequence of T = 64 (the 16 x 64 map along W: the four waves share the projection, wave 0
// scans -- one wavefront per sequence, as in gru.hip) or FOUR sequences of T = 16 (along H: the waves share the projection of all four --
// one fetch of the weight fragments instead of four -- and each scans one).
//   phase 1  the [64 x Cin] input panel as MFMA fragments: a 16x16x32 fragment is 8 consecutive channels of one pixel per lane (32
//            contiguous bytes), so no transposition is needed.  Wave w fetches row tile w ONLY, applies the fused prologue of the
//            convolution loaders (BatchNorm affine, residual add, concatenated text strip: conv_loader.h's LD bits 1 / 4 / 16), splits
//            the values into bf16 terms (conv_xbf_common.h: exact three-term or two-term split-operand arithmetic) and passes the
//            fragments on through LDS -- the staging area is the gi block itself, not yet written -- so the panel crosses the L1 once per
//            workgroup, not once per wave (round 5's form: ~10 of the 45 us of a launch, 100 MB through the L1s for 25 MB of input);
//   phase 2  gi^T tiles = Wc^T-tile x panel^T: the weight fragment is the A operand, the panel the B operand, so a lane ends up with FOUR
//            CONSECUTIVE gate columns of one time step -> one ds_write_b128 per 16 x 16 tile into gi [64][196] (row pitch 196 floats: the
//            sixteen time steps of a tile land in distinct bank quads).  Weight fragments come from the planes tpgsr_split_bf_program
//            wrote for the 32x32x16 kernels ([term][n/32][k/16][lane][8]); a 16x16x32 fragment is a different 16-byte gather of the same
//            bytes.  288 MFMAs per workgroup in two-term arithmetic (~2 us), the weights stay L2-resident.
//   phase 3  the scan of bigru_fwd_kernel, value for value (same gate functions, same packed-FMA order), with the step's three input
//            projections read from LDS one step ahead instead of from HBM eight steps ahead.
// LDS: 64 x 196 x 4 B + 512 B per scanning wave = 49.5 / 51 KB per workgroup (three per CU = the 768 workgroups of either scan in one
// round).  Forward only: back-propagation through time keeps its own kernels.
//
// REPEATABILITY (round 6; profiles/r06_gru_proj_root_cause.md).  Round 5 built this staged form and took it back: with three workgroups
// per CU, 2-5 of 3072 sequences came out different from launch to launch.  Found: nothing was wrong with the staging, the barriers or LDS
// (gi right before and after every scan, fragments identical in all four waves, W_hh intact in the registers afterwards).  One packed
// multiply-add of the scan -- v_pk_fma_f32 ... op_sel:[0,1,0], the form that takes its LOW half from the ODD register of a pair, here h.y
// of a broadcast ds_read_b128 that had just returned -- read that register as ZERO in lanes 48-63 of one time step (the high half of the
// SAME instruction read it correctly), whenever three scanning waves shared a SIMD.  gru_common.h's gru_dup_odd() takes the odd
// component through a v_mov first; no packed instruction of the scans carries that op_sel any more (tests/test_round6_cpu.py checks the
// ISA), and tests/test_gru_soak_gpu.py launches the kernels 1000 times next to a co-running load: same bits.
#include "conv_xbf_common.h"
#include "gru_common.h"
#include <mutex>
#include <vector>

#define GP_RS 196     // floats per gi row in LDS

// LAB BUILDS ONLY (-DTPGSR_LAB, tools/lab/gp_probe.py): parts of the kernel switched off to time the rest -- bit 0 = no scan (wave 0 leaves
// after the barrier), 1 = the panel is not loaded (zeros), 2 = no MFMAs / gi stores of phase 2, 3 = W_hh not loaded (constants), 4 = the scan
// stores nothing to global memory, 5 = the scan's gate math replaced by two multiplies.  Results are garbage with any bit set; in a release
// build the switch is the constant 0 and every test of it folds away (the ISA is the same with and without these lines).
#ifdef TPGSR_LAB
__device__ int g_gp_dbg = 0;
extern "C" int tpgsr_gp_debug(int bits) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_gp_dbg), &bits, sizeof(bits)) == hipSuccess ? 0 : TPGSR_ERR_LAUNCH;
}
#define GP_DBG() __builtin_amdgcn_readfirstlane(g_gp_dbg)
#else
#define GP_DBG() 0
#endif

__device__ __forceinline__ floatx4 gp_mfma(const bf16x8 a, const bf16x8 b, const floatx4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// acc += sum over the term pairs kept by the policy, smallest magnitudes first (as mfma_terms of conv_xbf_common.h)
template <int TT>
__device__ __forceinline__ floatx4 gp_mfma_terms(const bf16x8 (&w)[TT], const bf16x8 (&x)[TT], floatx4 acc) {
  if (TT == 3) {
    acc = gp_mfma(w[0], x[2], acc);
    acc = gp_mfma(w[2], x[0], acc);
    acc = gp_mfma(w[1], x[1], acc);
    acc = gp_mfma(w[0], x[1], acc);
    acc = gp_mfma(w[1], x[0], acc);
  }
  if (TT == 2) {
    acc = gp_mfma(w[0], x[1], acc);
    acc = gp_mfma(w[1], x[0], acc);
  }
  return gp_mfma(w[0], x[0], acc);
}

// LD: 1 = per-channel affine (BatchNorm) on the image channels, 4 = residual add (in2), 16 = channels >= cin_a from the [N][W][.] strip
template <int LD, int TT, int NRT, int NKS, bool TRAIN>
__global__ __launch_bounds__(256) void bigru_proj_fwd_kernel(const tpgsr_bigru_proj_args p) {
  // T = 64: one sequence per workgroup (four row tiles of 16 steps, wave 0 scans).  T = 16: FOUR sequences per workgroup -- one row tile and
  // one scanning wave each: the weight fragments of phase 2 (49-74 KB per workgroup from L2, 221 MB per launch when every 16-step sequence
  // fetched them for itself: 7.6 of that launch's 43 us, tools/lab/gp_probe.py) serve four sequences, and no wave idles through the scan
  constexpr int T = 16 * NRT, SPW = NRT == 1 ? 4 : 1, RT = NRT * SPW;
  extern __shared__ __attribute__((aligned(16))) float gsm[];      // gi [64][GP_RS], then hs [SPW][2][64]
  float* const gi = gsm;
  float* const hs = gsm + 16 * RT * GP_RS;
  const tpgsr_conv_args& a = p.c;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const SeqGeom g = seq_geom(SPW * blockIdx.x + (SPW > 1 ? wave : 0), a.N, a.H, a.W, p.axis);      // (SPW > 1: the sequence this wave scans)
  if (!g.active) return;        // (workgroup-uniform: the launcher sends whole workgroups only)
  const int l16 = lane & 15, kq = lane >> 4;
  const int dbg = GP_DBG();      // (0 in a release build: every test below folds away)

  // ---- phase 1: the panel as split MFMA fragments: xf[rt][ks][term] = 8 channels 32 ks + 8 kq .. of row 16 rt + l16 (time step 16 rt + l16 of
  // the one sequence, or step l16 of sequence rt of the four).  Wave w fetches, finishes (prologue) and splits row tile w and stages it ----
  bf16x8 xf[RT][NKS][TT];
  {
    u32x4* const stage = reinterpret_cast<u32x4*>(gsm);      // [rt][ks][term][lane] 16-byte pieces: RT NKS TT KB <= 36 KB of the 49 KB gi block
    const int hw = a.H * a.W;
    float4 lo[NKS], hi[NKS], lo2[NKS], hi2[NKS];
    // (SPW > 1: g is this wave's own sequence = row tile `wave`)
    const int pix = SPW == 1 ? g.base + (16 * wave + l16) * g.stride : g.base + l16 * g.stride;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int c = 32 * ks + 8 * kq;
      const float* src;
      if ((LD & 16) && c >= a.cin_a) {        // the text strip: one row per (image, column), shared by all H rows
        const int n = pix / hw, w = pix % a.W;
        src = a.in_b + (size_t)(n * a.W + w) * a.in_b_ld + (c - a.cin_a);
      } else {
        src = a.in + (size_t)pix * a.in_ld + a.in_coff + c;
      }
      if (dbg & 2) {
        lo[ks] = hi[ks] = lo2[ks] = hi2[ks] = make_float4(0.f, 0.f, 0.f, 0.f);
        continue;
      }
      lo[ks] = *reinterpret_cast<const float4*>(src);
      hi[ks] = *reinterpret_cast<const float4*>(src + 4);
      if (LD & 4) {
        const float* s2 = a.in2 + (size_t)pix * a.in2_ld + c;
        lo2[ks] = *reinterpret_cast<const float4*>(s2);
        hi2[ks] = *reinterpret_cast<const float4*>(s2 + 4);
      }
    }
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int c = 32 * ks + 8 * kq;
      const bool img = !(LD & 16) || c < a.cin_a;
      float4 s0 = make_float4(1.f, 1.f, 1.f, 1.f), s1 = s0, t0 = make_float4(0.f, 0.f, 0.f, 0.f), t1 = t0;
      if ((LD & 1) && img) {
        s0 = *reinterpret_cast<const float4*>(a.in_scale + c);
        s1 = *reinterpret_cast<const float4*>(a.in_scale + c + 4);
        t0 = *reinterpret_cast<const float4*>(a.in_shift + c);
        t1 = *reinterpret_cast<const float4*>(a.in_shift + c + 4);
      }
      float4 u = lo[ks], v = hi[ks];
      if (LD & 1) {      // (identity on the strip's channels: 1 * x + 0 is exact)
        u.x = u.x * s0.x + t0.x; u.y = u.y * s0.y + t0.y; u.z = u.z * s0.z + t0.z; u.w = u.w * s0.w + t0.w;
        v.x = v.x * s1.x + t1.x; v.y = v.y * s1.y + t1.y; v.z = v.z * s1.z + t1.z; v.w = v.w * s1.w + t1.w;
      }
      if (LD & 4) {
        u.x += lo2[ks].x; u.y += lo2[ks].y; u.z += lo2[ks].z; u.w += lo2[ks].w;
        v.x += hi2[ks].x; v.y += hi2[ks].y; v.z += hi2[ks].z; v.w += hi2[ks].w;
      }
      uint2 hu[TT], hv[TT];
      split4<TT>(u, hu);
      split4<TT>(v, hv);
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        u32x4 q;
        q.x = hu[t].x; q.y = hu[t].y; q.z = hv[t].x; q.w = hv[t].y;
        stage[((wave * NKS + ks) * TT + t) * 64 + lane] = q;
      }
    }
    __syncthreads();      // the four row tiles' fragments are staged
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int t = 0; t < TT; ++t) xf[rt][ks][t] = __builtin_bit_cast(bf16x8, stage[((rt * NKS + ks) * TT + t) * 64 + lane]);
    __syncthreads();      // everybody holds the whole panel: phase 2 may overwrite the staging area with gi
  }

  // ---- phase 2: gi [row][col] = bc[col] + sum_k panel[row][k] Wc[k][col], twelve 16-column tiles ----
  {
    constexpr int KB16 = 2 * NKS;                       // k-blocks of 16 in the split planes (kp = 32 NKS)
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(reinterpret_cast<const float*>(a.wt_bf), (size_t)TT * 6 * 32 * (32 * NKS) / 2);
    constexpr unsigned plane_w = 6u * KB16 * 1024u;
    // this lane's 16-byte piece of a (16-column tile ct, k-step ks) weight fragment inside the 32x32x16 fragment planes: column
    // ct 16 + l16 -> block (ct >> 1), lane slot (ct & 1) 16 + l16 (+ 32 for the upper 8 of a 16-k block); k = 32 ks + 8 kq -> k-block
    // 2 ks + (kq >> 1), upper half when kq is odd
    const unsigned wlane = ((unsigned)l16 + 32u * (kq & 1)) * 16u + (unsigned)(kq >> 1) * 1024u;
    // the four waves share the projection: wave w computes column tiles w, w + 4, w + 8 for all 64 rows (alone, one wave spent ~8 us here
    // before its scan could start, with the other three SIMDs of the CU idle)
    for (int ct = wave; ct < 12 && !(dbg & 4); ct += 4) {
      bf16x8 wf[NKS][TT];
      const unsigned wbase = ((unsigned)(ct >> 1) * KB16) * 1024u + (unsigned)(ct & 1) * 256u + wlane;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int t = 0; t < TT; ++t)
          wf[ks][t] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)(wbase + t * plane_w + (unsigned)ks * 2048u), 0, 0));
      const float4 b4 = *reinterpret_cast<const float4*>(a.bias + ct * 16 + 4 * kq);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) acc = gp_mfma_terms<TT>(wf[ks], xf[rt][ks], acc);
        // lane: row 16 rt + l16, gate columns ct 16 + 4 kq .. + 3
        float4 o;
        o.x = acc[0] + b4.x; o.y = acc[1] + b4.y; o.z = acc[2] + b4.z; o.w = acc[3] + b4.w;
        *reinterpret_cast<float4*>(gi + (16 * rt + l16) * GP_RS + ct * 16 + 4 * kq) = o;
      }
    }
  }

  __syncthreads();      // all of gi is in LDS
  if (wave >= SPW || (dbg & 1)) return;      // a scan is one wave's work (LDS stays allocated until the last one is through)
  float* const hsw = hs + (SPW > 1 ? wave * 128 : 0);                      // this wave's exchange slot [2][64]
  const float* const giw = gi + (SPW > 1 ? wave * T * GP_RS : 0);          // ... and its rows of gi

  // ---- phase 3: the scan (bigru_fwd_kernel of gru.hip, the inputs out of LDS) ----
  const int d = lane >> 5, j = lane & 31;
  f2 wrz[GRU_H], wn2[GRU_H / 2];
  {
    const float* pr = p.w_hh + ((size_t)(d * 96 + 0 * 32 + j)) * GRU_H;
    const float* pz = p.w_hh + ((size_t)(d * 96 + 1 * 32 + j)) * GRU_H;
    const float* pn = p.w_hh + ((size_t)(d * 96 + 2 * 32 + j)) * GRU_H;
    if (dbg & 8) {
#pragma unroll
      for (int k = 0; k < GRU_H; ++k) wrz[k] = mk2(0.01f * k, 0.02f);
#pragma unroll
      for (int k = 0; k < GRU_H / 2; ++k) wn2[k] = mk2(0.03f, 0.01f * k);
    } else {
#pragma unroll
      for (int k = 0; k < GRU_H; ++k) wrz[k] = mk2(pr[k], pz[k]);
#pragma unroll
      for (int k = 0; k < GRU_H / 2; ++k) wn2[k] = mk2(pn[2 * k], pn[2 * k + 1]);
    }
  }
  const float br = p.b_hh[d * 96 + j], bz = p.b_hh[d * 96 + 32 + j], bn = p.b_hh[d * 96 + 64 + j];
  float h = 0.f;
  hsw[lane] = 0.f;
  __builtin_amdgcn_wave_barrier();      // LDS operations of one wave execute in order; this only pins the compiler's order
  const int dpix = d == 0 ? g.stride : -g.stride;
  int pix = g.base + (d == 0 ? 0 : (T - 1) * g.stride);
  const int drow = d == 0 ? GP_RS : -GP_RS;
  const float* gp = giw + (d == 0 ? 0 : (T - 1) * GP_RS) + d * 96 + j;      // this lane's r-gate input of the current step
  float cr = gp[0], cz = gp[32], cn = gp[64];
#pragma unroll 2
  for (int step = 0; step < T; ++step) {
    const float ir = cr, iz = cz, in_ = cn;
    if (step + 1 < T) {                   // next step's inputs: issued now, needed one step later
      gp += drow;
      cr = gp[0]; cz = gp[32]; cn = gp[64];
    }
    f2 a0 = mk2(0.f, 0.f), a1 = a0, a2 = a0, a3 = a0, n0 = a0, n1 = a0;
    const float4* hp = reinterpret_cast<const float4*>(&hsw[(step & 1) * 64 + d * 32]);
#pragma unroll
    for (int k = 0; k < GRU_H / 4; ++k) {
      const float4 hv = hp[k];
      a0 = pk_fma(wrz[4 * k], mk2(hv.x, hv.x), a0);
      a1 = pk_fma(wrz[4 * k + 1], gru_dup_odd(hv.y), a1);      // (NOT mk2(hv.y, hv.y): gru_common.h)
      a2 = pk_fma(wrz[4 * k + 2], mk2(hv.z, hv.z), a2);
      a3 = pk_fma(wrz[4 * k + 3], gru_dup_odd(hv.w), a3);
      n0 = pk_fma(wn2[2 * k], mk2(hv.x, hv.y), n0);
      n1 = pk_fma(wn2[2 * k + 1], mk2(hv.z, hv.w), n1);
    }
    const f2 rz = (a0 + a1) + (a2 + a3), nn = n0 + n1;
    const float an = bn + (nn.x + nn.y);
    f2 sg;
    float n;
    if (dbg & 32) {
      sg = mk2(ir + (br + rz.x), iz + (bz + rz.y)) * mk2(0.25f, 0.25f);
      n = __builtin_fmaf(sg.x, an, in_) * 0.5f;
    } else {
      sg = gru_sigmoid2(mk2(ir + (br + rz.x), iz + (bz + rz.y)));      // both gates in lock step (gru_common.h)
      n = gru_tanh(__builtin_fmaf(sg.x, an, in_));
    }
    const float r = sg.x, z = sg.y;
    h = __builtin_fmaf(z, h, (1.f - z) * n);      // (explicit: the same contraction in every kernel that runs this step)
    hsw[((step + 1) & 1) * 64 + lane] = h;
    if (!(dbg & 16)) p.h_out[pix * 64 + d * 32 + j] = h;
    if (TRAIN && !(dbg & 16)) {
      float* q = p.gates + pix * 256 + d * 128 + j;
      q[0] = r; q[32] = z; q[64] = n; q[96] = an;
    }
    pix += dpix;
    __builtin_amdgcn_wave_barrier();
  }
}

static int g_gp_on = [] { const char* e = getenv("TPGSR_GRU_PROJ_FUSE"); return (e && e[0] == '0') ? 0 : 1; }();
/* experiment / test switch: 0 makes tpgsr_bigru_proj_supported() say no (the engines then record projection + scan as two launches) */
extern "C" void tpgsr_bigru_proj_set_enabled(int on) { g_gp_on = on ? 1 : 0; }

static int gp_loader_bits(const tpgsr_conv_args* a) {
  return (a->in_scale ? 1 : 0) | (a->in_act ? 2 : 0) | (a->in2 ? 4 : 0) | (a->in_ps ? 8 : 0) | (a->in_b ? 16 : 0);
}

/* 1 when tpgsr_bigru_proj_fwd takes this GruBlock: split-operand arithmetic (terms 1..3 with the pre-split weight planes), Cin 64 or 96,
 * 192 gate columns, scan length 16 (a multiple of four sequences) or 64, a loader it has (plain, BatchNorm affine, residual add, affine + text strip) */
extern "C" int tpgsr_bigru_proj_supported(const tpgsr_bigru_proj_args* p) {
  if (!p || !g_gp_on) return 0;
  const tpgsr_conv_args* a = &p->c;
  const int T = p->axis == 0 ? a->W : a->H, ld = gp_loader_bits(a);
  if (a->terms < 1 || a->terms > 3 || !a->wt_bf || a->wt_bf_cin != 0 || a->Cout != 192 || (a->Cin != 64 && a->Cin != 96) || a->kp != a->Cin) return 0;
  if (a->KH * a->KW != 1 || a->wt_ld || a->wt_coff || a->in_coff || a->stride_w > 1 || a->in_dil_w > 1) return 0;
  if (T != 16 && T != 64) return 0;
  if (T == 16 && ((p->axis == 0 ? a->N * a->H : a->N * a->W) & 3)) return 0;      // (16-step sequences go four to a workgroup)
  if (!(ld == 0 || ld == 1 || ld == 4 || ld == 17)) return 0;
  if ((ld & 16) && (a->cin_a != 64 || a->Cin != 96)) return 0;
  return 1;
}

extern "C" int tpgsr_bigru_proj_fwd(const tpgsr_bigru_proj_args* p, void* stream) {
  TPGSR_CHECK_ARG(p && p->c.in && p->c.bias && p->w_hh && p->b_hh && p->h_out, "tpgsr_bigru_proj_fwd: null pointer");
  TPGSR_CHECK_ARG(p->axis == 0 || p->axis == 1, "tpgsr_bigru_proj_fwd: bad axis");
  TPGSR_CHECK_ARG(tpgsr_bigru_proj_supported(p), "tpgsr_bigru_proj_fwd: this GruBlock is not the fused kernel's (ask tpgsr_bigru_proj_supported first: "
                  "Cin %d, terms %d, scan length %d, loader %d)", p->c.Cin, p->c.terms, p->axis == 0 ? p->c.W : p->c.H, gp_loader_bits(&p->c));
  const tpgsr_conv_args* a = &p->c;
  TPGSR_CHECK_ARG((long long)a->N * a->H * a->W * 256 < (1ll << 31), "tpgsr_bigru_proj_fwd: map too large for the kernel's 32-bit indices");
  TPGSR_CHECK_ARG((a->in_ld & 3) == 0 && (((uintptr_t)a->in | (uintptr_t)a->bias) & 15) == 0 && (!a->in2 || ((a->in2_ld & 3) == 0 && ((uintptr_t)a->in2 & 15) == 0)) &&
                  (!a->in_b || ((a->in_b_ld & 3) == 0 && ((uintptr_t)a->in_b & 15) == 0)) && (!a->in_scale || (((uintptr_t)a->in_scale | (uintptr_t)a->in_shift) & 15) == 0),
                  "tpgsr_bigru_proj_fwd: operands must be 16-byte aligned with row pitches that are multiples of 4 floats");
  const int T = p->axis == 0 ? a->W : a->H, nseq = p->axis == 0 ? a->N * a->H : a->N * a->W;
  const int ld = gp_loader_bits(a), nks = a->Cin / 32, tt = a->terms;
  const bool train = p->gates != nullptr;
  const void* fn = nullptr;
#define GP_PICK3(LDV, TTV, NRTV, NKSV) fn = train ? (const void*)bigru_proj_fwd_kernel<LDV, TTV, NRTV, NKSV, true> : (const void*)bigru_proj_fwd_kernel<LDV, TTV, NRTV, NKSV, false>;
#define GP_PICK2(LDV, TTV)                                       \
  if (T == 64 && nks == 2) { GP_PICK3(LDV, TTV, 4, 2) }           \
  else if (T == 16 && nks == 2) { GP_PICK3(LDV, TTV, 1, 2) }      \
  else if (T == 64 && nks == 3) { GP_PICK3(LDV, TTV, 4, 3) }      \
  else { GP_PICK3(LDV, TTV, 1, 3) }
#define GP_PICK1(LDV)                      \
  if (tt == 1) { GP_PICK2(LDV, 1) }        \
  else if (tt == 2) { GP_PICK2(LDV, 2) }   \
  else { GP_PICK2(LDV, 3) }
  switch (ld) {
    case 0: GP_PICK1(0) break;
    case 1: GP_PICK1(1) break;
    case 4: GP_PICK1(4) break;
    default: GP_PICK1(17) break;
  }
#undef GP_PICK1
#undef GP_PICK2
#undef GP_PICK3
  const int spw = T == 16 ? 4 : 1;
  const size_t lds = ((size_t)64 * GP_RS + spw * 128) * sizeof(float);
  tpgsr_bigru_proj_args args = *p;
  void* params[] = {&args};
  if (hipLaunchKernel(fn, dim3((unsigned)(nseq / spw)), dim3(256), params, lds, (hipStream_t)stream) != hipSuccess) {      // (four waves: the kernel's row-tile split)
    tpgsr_set_error("tpgsr_bigru_proj_fwd: launch failed: %s", hipGetErrorString(hipGetLastError()));
    return TPGSR_ERR_LAUNCH;
  }
  TPGSR_LAUNCH_CHECK("tpgsr_bigru_proj_fwd");
}
