// LAB: the STAGED form of the one-launch GruBlock forward (tools/lab/gru_proj_staged.patch of round 5: four 16-step sequences per workgroup,
// the input panel staged through LDS once per workgroup) with instrumentation, to find out why it was not repeatable at full batch
// (profiles/r05j_gru_scan_anatomy.md section 6).  Stand-alone library: hipcc -shared of this file + csrc/error.cpp.
//   mode bit 0  every wave leaves an XOR checksum of its panel fragments (xf) per lane            -> dump_x [wg][4][64] u32
//        bit 1  the whole gi block right after the barrier that follows phase 2                     -> dump_g1 [wg][64][192] f32
//        bit 2  each scanning wave's gi rows again after its scan                                   -> dump_g2 [wg][64][192] f32
//        bit 3  the staging area as wave 0 sees it right after the first barrier                    -> dump_s [wg][stage words] u32
//        bit 4  where and when the workgroup ran: HW_ID, XCC_ID, LDS_ALLOC, wall clock at start / end   -> dump_i [wg][4 waves][8] u32
// VAR (compile time): 0 = the patch as it was; 1 = every barrier written out as s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier, twice;
//   2 = the staging area does NOT alias gi (it sits behind the exchange slots: more LDS); 3 = gi stored as four ds_write_b32;
//   4 = eight values per (step, unit) into a [P][512] gates buffer: r z n an AND ir iz in rz.x (what went into the gates);
//   5 = no gate stores; 6 = s_waitcnt vmcnt(0) lgkmcnt(0) at the top of every step; 7 = the step's prefetch has landed before its gate math;
//   8 = the two sigmoids one after the other (not packed); 9 = idle issue slots behind every transcendental of the sigmoids;
//   11 = the recurrent weights dumped from the registers after the scan (dump_g2 [wg][4][96][64])
//   12 = the step's eight exchange reads (h of the previous step, broadcast ds_read_b128) all landed + sixteen idle issue slots before the first
//        multiply-add; 13 = the same without the idle slots
//   14 = as 0, but h.y of every exchange read goes through a v_mov first (no op_sel read of the odd register of a just-returned pair);
//   16 = all eight exchange reads in flight, each group consumed behind a hand-written partial wait (lgkmcnt 7 .. 0)
//   10 = a barrier after the W_hh loads (no VMEM return during the first steps of anybody's scan)
#include "../../tpgsr_amd/csrc/conv_xbf_common.h"
#include "../../tpgsr_amd/csrc/gru_common.h"

#define GP_RS 196
#define LAB_GETREG(id) __builtin_amdgcn_s_getreg((id) | (31 << 11))

struct gp_lab_args {
  tpgsr_bigru_proj_args p;
  unsigned* dump_x;
  float* dump_g1;
  float* dump_g2;
  unsigned* dump_s;
  unsigned* dump_i;
  int mode;
  int extra;
};

__device__ __forceinline__ floatx4 gp_mfma(const bf16x8 a, const bf16x8 b, const floatx4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
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

// VAR 9: the two sigmoids with idle issue slots behind every transcendental instruction (is a result or a source picked up too early / late?)
__device__ __forceinline__ f2 lab_sigmoid2_nops(f2 x) {
  const f2 nx = mk2(fminf(-x.x, 80.f), fminf(-x.y, 80.f));
  const f2 l2e = mk2(1.44269504088896341f, 1.44269504088896341f);
  const f2 t = nx * l2e;
  f2 lo = pk_fma(nx, l2e, -t);
  lo = pk_fma(nx, mk2(1.925963033500011e-08f, 1.925963033500011e-08f), lo);
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
  float ex = __builtin_amdgcn_exp2f(t.x);
  asm volatile("s_nop 7\n\ts_nop 7" : "+v"(ex));
  float ey = __builtin_amdgcn_exp2f(t.y);
  asm volatile("s_nop 7\n\ts_nop 7" : "+v"(ey));
  const f2 e = mk2(ex, ey);
  const f2 d = mk2(1.f, 1.f) + pk_fma(e, lo * mk2(0.6931471805599453f, 0.6931471805599453f), e);
  float rx = __builtin_amdgcn_rcpf(d.x);
  asm volatile("s_nop 7\n\ts_nop 7" : "+v"(rx));
  float ry = __builtin_amdgcn_rcpf(d.y);
  asm volatile("s_nop 7\n\ts_nop 7" : "+v"(ry));
  const f2 r = mk2(rx, ry);
  return pk_fma(pk_fma(-d, r, mk2(1.f, 1.f)), r, r);
}

template <int VAR>
__device__ __forceinline__ void lab_sync() {
  if (VAR == 1) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n\ts_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  } else {
    __syncthreads();
  }
}

template <int LD, int TT, int NRT, int NKS, int VAR>
__global__ __launch_bounds__(256) void gp_staged_kernel(const gp_lab_args la) {
  const tpgsr_bigru_proj_args& p = la.p;
  constexpr int T = 16 * NRT, SPW = NRT == 1 ? 4 : 1, RT = NRT * SPW;
  constexpr int STAGE_PIECES = RT * NKS * TT * 64;      // 16-byte pieces
  extern __shared__ __attribute__((aligned(16))) float gsm[];
  float* const gi = gsm;
  float* const hs = gsm + 16 * RT * GP_RS;      // [SPW][2][64]
  const tpgsr_conv_args& a = p.c;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const SeqGeom g = seq_geom(SPW * blockIdx.x + (SPW > 1 && wave < SPW ? wave : 0), a.N, a.H, a.W, p.axis);
  if (!g.active) return;
  const int l16 = lane & 15, kq = lane >> 4;
  const int mode = la.mode;
  const unsigned long long t_start = wall_clock64();

  bf16x8 xf[RT][NKS][TT];
  {
    u32x4* const stage = reinterpret_cast<u32x4*>(VAR == 2 ? hs + SPW * 128 : gsm);
    const int hw = a.H * a.W;
    float4 lo[NKS], hi[NKS], lo2[NKS], hi2[NKS];
    const int pix = SPW == 1 ? g.base + (16 * wave + l16) * g.stride : g.base + l16 * g.stride;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int c = 32 * ks + 8 * kq;
      const float* src;
      if ((LD & 16) && c >= a.cin_a) {
        const int n = pix / hw, w = pix % a.W;
        src = a.in_b + (size_t)(n * a.W + w) * a.in_b_ld + (c - a.cin_a);
      } else {
        src = a.in + (size_t)pix * a.in_ld + a.in_coff + c;
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
      if (LD & 1) {
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
    lab_sync<VAR>();
    if ((mode & 8) && wave == 0) {
      for (int i = lane; i < STAGE_PIECES; i += 64) {
        const u32x4 q = stage[i];
        unsigned* d = la.dump_s + ((size_t)blockIdx.x * STAGE_PIECES + i) * 4;
        d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w;
      }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int t = 0; t < TT; ++t) xf[rt][ks][t] = __builtin_bit_cast(bf16x8, stage[((rt * NKS + ks) * TT + t) * 64 + lane]);
    lab_sync<VAR>();
  }
  if (mode & 1) {
    unsigned cs = 0;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          const u32x4 q = __builtin_bit_cast(u32x4, xf[rt][ks][t]);
          cs = cs * 31u + (q.x ^ (q.y * 3u) ^ (q.z * 5u) ^ (q.w * 7u));
        }
    la.dump_x[((size_t)blockIdx.x * 4 + wave) * 64 + lane] = cs;
  }

  {
    constexpr int KB16 = 2 * NKS;
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(reinterpret_cast<const float*>(a.wt_bf), (size_t)TT * 6 * 32 * (32 * NKS) / 2);
    constexpr unsigned plane_w = 6u * KB16 * 1024u;
    const unsigned wlane = ((unsigned)l16 + 32u * (kq & 1)) * 16u + (unsigned)(kq >> 1) * 1024u;
    for (int ct = wave; ct < 12; ct += 4) {
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
        float4 o;
        o.x = acc[0] + b4.x; o.y = acc[1] + b4.y; o.z = acc[2] + b4.z; o.w = acc[3] + b4.w;
        float* dst = gi + (16 * rt + l16) * GP_RS + ct * 16 + 4 * kq;
        if (VAR == 3) {
          dst[0] = o.x; dst[1] = o.y; dst[2] = o.z; dst[3] = o.w;
        } else {
          *reinterpret_cast<float4*>(dst) = o;
        }
      }
    }
  }

  lab_sync<VAR>();
  if (mode & 2) {
    for (int i = threadIdx.x; i < 64 * 192; i += 256) {
      const int r = i / 192, cc = i - r * 192;
      la.dump_g1[(size_t)blockIdx.x * 64 * 192 + i] = gi[r * GP_RS + cc];
    }
  }
  if ((mode & 16) && lane == 0 && wave >= SPW) {
    unsigned* o = la.dump_i + ((size_t)blockIdx.x * 4 + wave) * 8;
    const unsigned long long t1 = wall_clock64();
    o[0] = LAB_GETREG(4); o[1] = LAB_GETREG(20); o[2] = LAB_GETREG(6); o[3] = 0;
    o[4] = (unsigned)t_start; o[5] = (unsigned)(t_start >> 32); o[6] = (unsigned)t1; o[7] = (unsigned)(t1 >> 32);
  }
  if (wave >= SPW) return;
  float* const hsw = hs + (SPW > 1 ? wave * 128 : 0);
  const float* const giw = gi + (SPW > 1 ? wave * T * GP_RS : 0);

  const int d = lane >> 5, j = lane & 31;
  f2 wrz[GRU_H], wn2[GRU_H / 2];
  {
    const float* pr = p.w_hh + ((size_t)(d * 96 + 0 * 32 + j)) * GRU_H;
    const float* pz = p.w_hh + ((size_t)(d * 96 + 1 * 32 + j)) * GRU_H;
    const float* pn = p.w_hh + ((size_t)(d * 96 + 2 * 32 + j)) * GRU_H;
#pragma unroll
    for (int k = 0; k < GRU_H; ++k) wrz[k] = mk2(pr[k], pz[k]);
#pragma unroll
    for (int k = 0; k < GRU_H / 2; ++k) wn2[k] = mk2(pn[2 * k], pn[2 * k + 1]);
  }
  const float br = p.b_hh[d * 96 + j], bz = p.b_hh[d * 96 + 32 + j], bn = p.b_hh[d * 96 + 64 + j];
  if (VAR == 10) {      // every wave of the workgroup has its W_hh in registers before any scan starts (no VMEM return during the first steps)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  float h = 0.f;
  hsw[lane] = 0.f;
  __builtin_amdgcn_wave_barrier();
  const int dpix = d == 0 ? g.stride : -g.stride;
  int pix = g.base + (d == 0 ? 0 : (T - 1) * g.stride);
  const int drow = d == 0 ? GP_RS : -GP_RS;
  const float* gp = giw + (d == 0 ? 0 : (T - 1) * GP_RS) + d * 96 + j;
  float cr = gp[0], cz = gp[32], cn = gp[64];
#pragma unroll 2
  for (int step = 0; step < T; ++step) {
    if (VAR == 6) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // nothing in flight at the top of a step
    const float ir = cr, iz = cz, in_ = cn;
    if (step + 1 < T) {
      gp += drow;
      cr = gp[0]; cz = gp[32]; cn = gp[64];
    }
    if (VAR == 7) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cr), "+v"(cz), "+v"(cn) : : "memory");      // the prefetch has landed before the gate math starts
    f2 a0 = mk2(0.f, 0.f), a1 = a0, a2 = a0, a3 = a0, n0 = a0, n1 = a0;
    const float4* hp = reinterpret_cast<const float4*>(&hsw[(step & 1) * 64 + d * 32]);
    float4 hall[GRU_H / 4];
    if (VAR == 16) {      // all eight reads in flight, consumed behind hand-written PARTIAL waits (lgkmcnt 7, 6, ... 0): maximal overlap of returns and reads
#pragma unroll
      for (int k = 0; k < GRU_H / 4; ++k) hall[k] = hp[k];
    }
    if (VAR == 12 || VAR == 13) {      // all eight exchange reads first, everything landed (12: and sixteen idle issue slots) before the first multiply-add
#pragma unroll
      for (int k = 0; k < GRU_H / 4; ++k) hall[k] = hp[k];
      if (VAR == 12) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < GRU_H / 4; ++k) asm volatile("" : "+v"(hall[k].x), "+v"(hall[k].y), "+v"(hall[k].z), "+v"(hall[k].w));
    }
#pragma unroll
    for (int k = 0; k < GRU_H / 4; ++k) {
      float4 hv = (VAR == 12 || VAR == 13 || VAR == 16) ? hall[k] : hp[k];
      if (VAR == 16) {
        switch (k) {
          case 0: asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(hv.x), "+v"(hv.y), "+v"(hv.z), "+v"(hv.w)); break;
          case 1: asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(hv.x), "+v"(hv.y), "+v"(hv.z), "+v"(hv.w)); break;
          case 2: asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(hv.x), "+v"(hv.y), "+v"(hv.z), "+v"(hv.w)); break;
          case 3: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(hv.x), "+v"(hv.y), "+v"(hv.z), "+v"(hv.w)); break;
          case 4: asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(hv.x), "+v"(hv.y), "+v"(hv.z), "+v"(hv.w)); break;
          case 5: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(hv.x), "+v"(hv.y), "+v"(hv.z), "+v"(hv.w)); break;
          case 6: asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(hv.x), "+v"(hv.y), "+v"(hv.z), "+v"(hv.w)); break;
          default: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(hv.x), "+v"(hv.y), "+v"(hv.z), "+v"(hv.w)); break;
        }
      }
      if (VAR == 14) {      // the y component through a register of its own: no packed instruction takes its LOW half from the ODD register of the returned pair
        float y2;
        asm volatile("v_mov_b32 %0, %1" : "=v"(y2) : "v"(hv.y));
        hv.y = y2;
      }
      a0 = pk_fma(wrz[4 * k], mk2(hv.x, hv.x), a0);
      a1 = pk_fma(wrz[4 * k + 1], mk2(hv.y, hv.y), a1);
      a2 = pk_fma(wrz[4 * k + 2], mk2(hv.z, hv.z), a2);
      a3 = pk_fma(wrz[4 * k + 3], mk2(hv.w, hv.w), a3);
      n0 = pk_fma(wn2[2 * k], mk2(hv.x, hv.y), n0);
      n1 = pk_fma(wn2[2 * k + 1], mk2(hv.z, hv.w), n1);
    }
    const f2 rz = (a0 + a1) + (a2 + a3), nn = n0 + n1;
    const float an = bn + (nn.x + nn.y);
    f2 sg;
    if (VAR == 9) sg = lab_sigmoid2_nops(mk2(ir + (br + rz.x), iz + (bz + rz.y)));
    else if (VAR == 8) sg = mk2(gru_rcp(1.f + gru_exp(fminf(-(ir + (br + rz.x)), 80.f))), gru_rcp(1.f + gru_exp(fminf(-(iz + (bz + rz.y)), 80.f))));
    else sg = gru_sigmoid2(mk2(ir + (br + rz.x), iz + (bz + rz.y)));
    const float n = gru_tanh(__builtin_fmaf(sg.x, an, in_));
    const float r = sg.x, z = sg.y;
    h = __builtin_fmaf(z, h, (1.f - z) * n);
    hsw[((step + 1) & 1) * 64 + lane] = h;
    p.h_out[pix * 64 + d * 32 + j] = h;
    if (VAR == 4) {             // eight values per (step, unit): the gates AND what went into them (gates buffer [P][512])
      float* q = p.gates + pix * 512 + d * 256 + j;
      q[0] = r; q[32] = z; q[64] = n; q[96] = an; q[128] = ir; q[160] = iz; q[192] = in_; q[224] = rz.x;
    } else if (VAR != 5 && p.gates) {
      float* q = p.gates + pix * 256 + d * 128 + j;
      q[0] = r; q[32] = z; q[64] = n; q[96] = an;
    }
    pix += dpix;
    __builtin_amdgcn_wave_barrier();
  }
  if (VAR == 11) {      // the recurrent weights as the registers hold them AFTER the scan -> dump_g2 [wg][4 waves][96][64 lanes]
    float* o = la.dump_g2 + ((size_t)blockIdx.x * 4 + wave) * 96 * 64 + lane;
#pragma unroll
    for (int k = 0; k < GRU_H; ++k) { o[(2 * k) * 64] = wrz[k].x; o[(2 * k + 1) * 64] = wrz[k].y; }
#pragma unroll
    for (int k = 0; k < GRU_H / 2; ++k) { o[(64 + 2 * k) * 64] = wn2[k].x; o[(65 + 2 * k) * 64] = wn2[k].y; }
  }
  if (mode & 4) {
    for (int i = lane; i < T * 192; i += 64) {
      const int r = i / 192, cc = i - r * 192;
      la.dump_g2[(size_t)blockIdx.x * 64 * 192 + (size_t)(SPW > 1 ? wave * T : 0) * 192 + i] = giw[r * GP_RS + cc];
    }
  }
  if ((mode & 16) && lane == 0) {
    unsigned* o = la.dump_i + ((size_t)blockIdx.x * 4 + wave) * 8;
    const unsigned long long t1 = wall_clock64();
    o[0] = LAB_GETREG(4); o[1] = LAB_GETREG(20); o[2] = LAB_GETREG(6); o[3] = 1;
    o[4] = (unsigned)t_start; o[5] = (unsigned)(t_start >> 32); o[6] = (unsigned)t1; o[7] = (unsigned)(t1 >> 32);
  }
}

// variant = VAR; loader 1 (affine, Cin 64) or 17 (affine + strip, Cin 96); terms 2 or 3; T 16 or 64
extern "C" int gp_lab_launch(const gp_lab_args* la, int variant, void* stream) {
  const tpgsr_conv_args* a = &la->p.c;
  const int T = la->p.axis == 0 ? a->W : a->H, nseq = la->p.axis == 0 ? a->N * a->H : a->N * a->W;
  const int ld = (a->in_scale ? 1 : 0) | (a->in2 ? 4 : 0) | (a->in_b ? 16 : 0), nks = a->Cin / 32, tt = a->terms;
  const void* fn = nullptr;
#define PICKV(LDV, TTV, NRTV, NKSV)                                                            \
  switch (variant) {                                                                           \
    case 0: fn = (const void*)gp_staged_kernel<LDV, TTV, NRTV, NKSV, 0>; break;                \
    case 1: fn = (const void*)gp_staged_kernel<LDV, TTV, NRTV, NKSV, 1>; break;                \
    case 2: fn = (const void*)gp_staged_kernel<LDV, TTV, NRTV, NKSV, 2>; break;                \
    case 3: fn = (const void*)gp_staged_kernel<LDV, TTV, NRTV, NKSV, 3>; break;                \
    case 4: fn = (const void*)gp_staged_kernel<LDV, TTV, NRTV, NKSV, 4>; break;                \
    case 5: fn = (const void*)gp_staged_kernel<LDV, TTV, NRTV, NKSV, 5>; break;                \
    case 6: fn = (const void*)gp_staged_kernel<LDV, TTV, NRTV, NKSV, 6>; break;                \
    case 7: fn = (const void*)gp_staged_kernel<LDV, TTV, NRTV, NKSV, 7>; break;                \
    case 8: fn = (const void*)gp_staged_kernel<LDV, TTV, NRTV, NKSV, 8>; break;                \
    case 9: fn = (const void*)gp_staged_kernel<LDV, TTV, NRTV, NKSV, 9>; break;                \
    case 11: fn = (const void*)gp_staged_kernel<LDV, TTV, NRTV, NKSV, 11>; break;              \
    case 12: fn = (const void*)gp_staged_kernel<LDV, TTV, NRTV, NKSV, 12>; break;              \
    case 13: fn = (const void*)gp_staged_kernel<LDV, TTV, NRTV, NKSV, 13>; break;              \
    case 14: fn = (const void*)gp_staged_kernel<LDV, TTV, NRTV, NKSV, 14>; break;              \
    case 16: fn = (const void*)gp_staged_kernel<LDV, TTV, NRTV, NKSV, 16>; break;              \
    default: fn = (const void*)gp_staged_kernel<LDV, TTV, NRTV, NKSV, 10>; break;              \
  }
  if (ld == 17 && tt == 2 && T == 16 && nks == 3) { PICKV(17, 2, 1, 3) }
  else if (ld == 17 && tt == 3 && T == 16 && nks == 3) { PICKV(17, 3, 1, 3) }
  else if (ld == 1 && tt == 2 && T == 16 && nks == 2) { PICKV(1, 2, 1, 2) }
  else if (ld == 4 && tt == 2 && T == 64 && nks == 2) { PICKV(4, 2, 4, 2) }
  else { tpgsr_set_error("gp_lab_launch: case not instantiated (ld %d terms %d T %d nks %d)", ld, tt, T, nks); return TPGSR_ERR_ARG; }
#undef PICKV
  const int spw = T == 16 ? 4 : 1;
  const int stage_bytes = (T == 16 ? 4 : 4) * nks * tt * 64 * 16;
  size_t lds = ((size_t)64 * GP_RS + spw * 128) * sizeof(float) + (variant == 2 ? stage_bytes : 0) + (size_t)la->extra;
  if (lds > 65536) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  gp_lab_args args = *la;
  void* params[] = {&args};
  if (hipLaunchKernel(fn, dim3((unsigned)(nseq / spw)), dim3(256), params, lds, (hipStream_t)stream) != hipSuccess) {
    tpgsr_set_error("gp_lab_launch: launch failed: %s", hipGetErrorString(hipGetLastError()));
    return TPGSR_ERR_LAUNCH;
  }
  TPGSR_LAUNCH_CHECK("gp_lab_launch");
}
