// BiLSTM (model/crnn/crnn.py:5-26, nn.LSTM(nIn, 256, bidirectional=True)) forward and backward recurrences as ONE PERSISTENT LAUNCH each,
// instead of two launches per time step (tpgsr_lstm_rec_gemm + tpgsr_lstm_step_{fwd,bwd}: 306 launches and the largest host-bound
// gaps of a C3 training step).  Hh == 256, N <= 64.
//
// Grid: 2 directions x 32 workgroups of 256 threads.  Workgroup (d, u) owns hidden units 8u .. 8u+7 of direction d = the 32 gate columns
// {q 256 + 8u + j : q = i,f,g,o; j < 8}; its slice of W_hh stays in LDS for all T steps, pre-split into three bf16 terms in MFMA
// fragment order (fp32-equivalent arithmetic as in conv_xbf.hip: six bf16 MFMAs per product block, fp32 accumulation).
//
// Inter-workgroup exchange, once per time step and direction, WITHOUT any fence: payload by 16-byte write-through (sc1) stores,
// `s_waitcnt vmcnt(0)`, workgroup barrier, ONE relaxed agent-scope arrival on a monotonic counter; consumers poll the counter with
// relaxed agent-scope (sc1) loads from one lane, pass a workgroup barrier and read the payload with sc1 loads (which bypass the
// vector L1).  This is the "{sc1 stores and sc1 loads on both sides}" form of MI355X_MICROARCH.md (Workgroup dispatch, XCD placement &
// inter-workgroup visibility): correct for any placement of the workgroups on XCDs, and -- unlike the release / acquire pair of the
// first version of this kernel (`buffer_wbl2` writes back every dirty line of the XCD's L2 while the other streams of the training
// step keep it dirty; measured slower than the launches it replaced) -- its cost does not depend on what else runs on the chip.
//
//   forward  (gather): h_t[n][256] is needed by everybody: every workgroup publishes its 8 units as bf16 terms in A-FRAGMENT order
//            (8 consecutive k of a fragment row = exactly one workgroup's units: one 16-byte store per sequence and term) into a
//            parity-double-buffered buffer hx and reads all of it back (96 KB) as MFMA operands: gh[n][32] = h[n][256] W^T[256][32].
//   backward (reduce-scatter): dh[n][j] = sum over ALL 1024 gate columns of dG[n][col] W_hh[col][j]: a workgroup contracts over ITS 32
//            gate columns for all 256 units, P_u[n][256] = dG_u[n][32] W_hh[cols_u][256], publishes P_u sorted by consumer, and every
//            consumer adds the 32 partial sums of its 8 units in producer order (deterministic).  Same 48 KB in and out per step as
//            the forward pass, where an all-gather of dG would move 192 KB per workgroup and step.
//
// Both kernels need their 64 workgroups to become resident together at some point (they are: 64 workgroups on 256 CUs; other
// kernels running next to them only delay that); a poll that sees nothing for ~1 s sets sync[2], gives up instead of hanging the GPU
// and POISONS its outputs with NaN from then on (see LS_SPINS below): a timed-out hand-off ends in a NaN loss, never in a silently
// wrong one.
#include "common.h"
#include "gru_common.h"   // the recurrences' gate functions: compensated v_exp_f32 + v_rcp_f32 with a Newton step (<= 4.5 ulp, a third of libm's instructions)
#include <mutex>

#define LS_NW 32
typedef __bf16 ls_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int ls_u32x4 __attribute__((ext_vector_type(4)));
#define LS_SC1 16   // cache-policy bit of the buffer intrinsics: sc1 (system-coherent level 1: write-through stores, L1-bypassing loads)

__device__ __forceinline__ void ls_split3(float x, unsigned short (&h)[3]) {
  __bf16 a = (__bf16)x;
  float r = x - (float)a;
  __bf16 b = (__bf16)r;
  __bf16 c = (__bf16)(r - (float)b);
  h[0] = __builtin_bit_cast(unsigned short, a);
  h[1] = __builtin_bit_cast(unsigned short, b);
  h[2] = __builtin_bit_cast(unsigned short, c);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t ls_rsrc(const void* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes > 0x7fffffffull ? 0x7fffffff : (int)bytes, 0x00020000);
}

__device__ __forceinline__ floatx16 ls_mfma6(const ls_bf16x8 (&a)[3], const ls_bf16x8 (&b)[3], floatx16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);   // smallest magnitudes first
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
}

// A hand-off that never completes must not pass silently: a workgroup that gives up sets sync[2], every workgroup
// that sees the flag stops waiting as well, and everything such a workgroup produces from then on is NaN -- the loss of the step goes
// NaN, which no training loop overlooks.  Budgets: ~1 s per wait (the 64 workgroups need to be co-resident at some point: other
// kernels only delay that), a handful of polls once the flag is up.
#define LS_SPINS (1 << 20)
#define LS_TRIES (1 << 18)

// one lane: wait until `target` workgroups have arrived on *cnt (relaxed agent-scope polls; a sleeping poller costs the memory
// system next to nothing).  Returns true when it gave up.
__device__ __forceinline__ bool ls_wait(unsigned* cnt, unsigned target, unsigned* flag) {
  int spins = 0;
  while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(1);
    ++spins;
    if ((spins & 255) == 0 && __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return true;
    if (spins > LS_SPINS) {
      __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return true;
    }
  }
  return false;
}

// ------------------------------------------------------------------------------------------------------
// forward.  G [N][T][2][4Hh]: input projections (+ b_ih) in, ACTIVATED gates out; whhT [2][Hh][4Hh] = W_hh^T; bhh [2][4Hh] or null;
// Cst [N][T][2][Hh]; out [N][T][2Hh];
//   hx   [2 parity][2 dir][3 terms][2 row blocks][16 k-blocks][64 lanes][8] bf16, zeroed ONCE by the caller (rows >= N stay 0)
//   sync [4] u32: arrival counters of the two directions, timeout flag, spare; zeroed by the launcher on the stream
// LDS (one object): W fragments [3][16 k-blocks][64][8] bf16 48 KB | partial sums [2 k-halves][64 rows][32 cols] f32 16 KB |
//   cell state [64][8] f32 2 KB | h terms [64][3][8] bf16 3 KB | b_hh [32] f32
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lstm_seq_fwd_kernel(float* __restrict__ G, const float* __restrict__ whhT,
                                                           const float* __restrict__ bhh, float* __restrict__ Cst,
                                                           float* __restrict__ out, unsigned short* __restrict__ hx,
                                                           unsigned* __restrict__ sync, int N, int T) {
  constexpr int Hh = 256, G4 = 1024;
  __shared__ __attribute__((aligned(16))) unsigned char lsm[48 * 1024 + 16 * 1024 + 2 * 1024 + 3 * 1024 + 128];
  unsigned short* wfr = reinterpret_cast<unsigned short*>(lsm);
  float* red = reinterpret_cast<float*>(lsm + 48 * 1024);
  float* cst = reinterpret_cast<float*>(lsm + 64 * 1024);
  unsigned short* hst = reinterpret_cast<unsigned short*>(lsm + 66 * 1024);
  float* bsm = reinterpret_cast<float*>(lsm + 69 * 1024);
  __shared__ int tmo_s;       // a hand-off of this workgroup timed out: everything it produces from then on is NaN
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = blockIdx.x / LS_NW, u = blockIdx.x % LS_NW, u0 = u * 8;
  const float* W = whhT + (size_t)d * Hh * G4;
  if (tid == 0) tmo_s = 0;
  // W fragments: B operand of k-block kb: lane l holds column c = l & 31 (gate c >> 3, unit u0 + (c & 7)), k = 16 kb + 8 (l >> 5) + j
  for (int idx = tid; idx < 16 * 64; idx += 256) {
    const int kb = idx >> 6, l = idx & 63, c = l & 31;
    const int col = (c >> 3) * Hh + u0 + (c & 7);
    unsigned short hv[8][3];
#pragma unroll
    for (int j = 0; j < 8; ++j) ls_split3(W[(size_t)(kb * 16 + (l >> 5) * 8 + j) * G4 + col], hv[j]);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      unsigned short* dst = wfr + ((size_t)(t * 16 + kb) * 64 + l) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) dst[j] = hv[j][t];
    }
  }
  for (int i = tid; i < 64 * 8; i += 256) cst[i] = 0.f;
  if (tid < 32) bsm[tid] = bhh ? bhh[(size_t)d * G4 + (tid >> 3) * Hh + u0 + (tid & 7)] : 0.f;
  __syncthreads();
  const int rb = wave & 1, kh = wave >> 1;
  const size_t dir_elems = (size_t)3 * 2 * 16 * 512, par_elems = 2 * dir_elems;       // bf16 elements
  const __amdgpu_buffer_rsrc_t rs_hx = ls_rsrc(hx, 2 * par_elems * 2);
  const int ul = tid & 7;
  for (int s = 0; s < T; ++s) {
    const int t = d == 0 ? s : T - 1 - s;
    // this step's input projections: independent of the exchange, so they are on their way while the poll below waits
    float pre[2][4];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int n = (tid >> 3) + 32 * it;
      if (n < N) {
        const float* g = G + (((size_t)n * T + t) * 2 + d) * G4 + u0 + ul;
#pragma unroll
        for (int q = 0; q < 4; ++q) pre[it][q] = g[q * Hh];
      }
    }
    if (s > 0) {
      if (tid == 0 && !tmo_s && ls_wait(sync + d, (unsigned)s * LS_NW, sync + 2)) tmo_s = 1;
      __syncthreads();
      // gh partial of this wave: row block rb, k-blocks 8 kh .. 8 kh + 7 of the h everybody published in step s - 1
      const unsigned hp = (unsigned)((((s - 1) & 1) * par_elems + d * dir_elems) * 2);
      ls_u32x4 av[8][3];
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int tt = 0; tt < 3; ++tt)
          av[i][tt] = __builtin_amdgcn_raw_buffer_load_b128(rs_hx, (int)(hp + ((((unsigned)(tt * 2 + rb) * 16 + kh * 8 + i) * 64 + lane) * 16)), 0, LS_SC1);
      floatx16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        ls_bf16x8 a[3], b[3];
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
          a[tt] = __builtin_bit_cast(ls_bf16x8, av[i][tt]);
          b[tt] = *reinterpret_cast<const ls_bf16x8*>(wfr + ((size_t)(tt * 16 + kh * 8 + i) * 64 + lane) * 8);
        }
        acc = ls_mfma6(a, b, acc);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        red[(kh * 64 + row) * 32 + (lane & 31)] = acc[r];
      }
      __syncthreads();
    }
    // gate math: item = (sequence n, local unit ul)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int n = (tid >> 3) + 32 * it;
      if (n < N) {
        const int item = n * 8 + ul, unit = u0 + ul;
        float p[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          p[q] = pre[it][q] + bsm[q * 8 + ul];
          if (s > 0) p[q] += red[n * 32 + q * 8 + ul] + red[(64 + n) * 32 + q * 8 + ul];
        }
        const float ig = gru_sigmoid1(p[0]), fg = gru_sigmoid1(p[1]), gg = gru_tanh(p[2]), og = gru_sigmoid1(p[3]);
        const float c = fg * cst[item] + ig * gg;
        const float h = og * gru_tanh(c);
        cst[item] = c;
        unsigned short hv[3];
        ls_split3(h, hv);
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) hst[(n * 3 + tt) * 8 + ul] = hv[tt];
        float* g = G + (((size_t)n * T + t) * 2 + d) * G4 + unit;
        g[0] = ig;
        g[Hh] = fg;
        g[2 * Hh] = gg;
        g[3 * Hh] = og;
        Cst[(((size_t)n * T + t) * 2 + d) * Hh + unit] = c;
        out[((size_t)n * T + t) * 2 * Hh + d * Hh + unit] = tmo_s ? __builtin_nanf("") : h;   // (only the STORE: the arithmetic above stays as it was)
      }
    }
    if (s + 1 < T) {
      __syncthreads();
      // publish: one 16-byte write-through store per (sequence, term) -- k-block u >> 1, fragment lanes (u & 1) 32 + (n & 31)
      if (tid < N * 3) {
        const int n = tid / 3, tt = tid - n * 3;
        const ls_u32x4 v = *reinterpret_cast<const ls_u32x4*>(hst + (n * 3 + tt) * 8);
        const unsigned off = (unsigned)(((s & 1) * par_elems + d * dir_elems) * 2) +
                             ((((unsigned)(tt * 2 + (n >> 5)) * 16 + (u >> 1)) * 64 + (u & 1) * 32 + (n & 31)) * 16);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs_hx, (int)off, 0, LS_SC1);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(sync + d, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

extern "C" int tpgsr_lstm_seq_fwd(float* G, const float* whhT, const float* bhh, float* Cst, float* out, void* hx, unsigned* sync, int N,
                                  int T, int Hh, void* stream) {
  TPGSR_CHECK_ARG(G && whhT && Cst && out && hx && sync && N > 0 && N <= 64 && T > 0 && Hh == 256,
                  "tpgsr_lstm_seq_fwd: needs Hh == 256, 1 <= N <= 64, T >= 1 and non-null buffers (got Hh %d, N %d, T %d)", Hh, N, T);
  if (hipMemsetAsync(sync, 0, 4 * sizeof(unsigned), (hipStream_t)stream) != hipSuccess) {
    tpgsr_set_error("tpgsr_lstm_seq_fwd: hipMemsetAsync failed");
    return TPGSR_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(lstm_seq_fwd_kernel, dim3(2 * LS_NW), dim3(256), 0, (hipStream_t)stream, G, whhT, bhh, Cst, out, (unsigned short*)hx,
                     sync, N, T);
  TPGSR_LAUNCH_CHECK("tpgsr_lstm_seq_fwd");
}

// ------------------------------------------------------------------------------------------------------
// forward, DATA-TAGGED hand-off ("granules", MI355X_MICROARCH.md price list: handoff-1to1 / allgather rows): no counter, no flag, no wait
// for the stores at all.  Every published value is one naturally aligned 8-byte granule {h1, h2, h3 (the three bf16 terms), tag}
// written by ONE sc1 store (two granules per 16-byte store; each 8-byte half lands untorn); tag = (launch epoch, step).  A consumer
// loads the granules it needs with sc1 loads and simply re-loads until every tag is this step's: the data is its own flag, so
// the per-step chain is  store -> (memory) -> load  instead of  store -> wait for the write-through ack -> barrier -> atomic ->
// poll -> barrier -> load.  Stale tags cannot match: the epoch (sync[4 + d], bumped by workgroup (d, 0) when it finishes -- which
// implies every workgroup of direction d has read it) changes with every launch, the step with every use of a parity buffer.
//   hg   [2 parity][2 dir][2 row blocks][16 k-blocks][4 quads][64 lanes][2] granules of 8 B (512 KB), zeroed ONCE by the caller:
//        fragment lane l of k-block kb needs the 8 granules k = 16 kb + 8 (l >> 5) + j of row (l & 31); quad q holds j = 2q, 2q + 1,
//        so one load instruction of a wave is 1 KB contiguous
//   sync [8] u32: [2] timeout flag, [4 + d] launch epoch of direction d; zeroed ONCE by the caller (not per launch)
// Needs T <= 31 (5 tag bits for the step).  LDS: W fragments 48 KB | partial sums, double-buffered by step parity, 2 x 16 KB |
// cell state 2 KB | b_hh.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lstm_seq_fwdg_kernel(float* __restrict__ G, const float* __restrict__ whhT,
                                                            const float* __restrict__ bhh, float* __restrict__ Cst,
                                                            float* __restrict__ out, unsigned* __restrict__ hg,
                                                            unsigned* __restrict__ sync, int N, int T) {
  constexpr int Hh = 256, G4 = 1024;
  __shared__ __attribute__((aligned(16))) unsigned char lsm[48 * 1024 + 32 * 1024 + 2 * 1024 + 128];
  unsigned short* wfr = reinterpret_cast<unsigned short*>(lsm);
  float* red0 = reinterpret_cast<float*>(lsm + 48 * 1024);
  float* cst = reinterpret_cast<float*>(lsm + 80 * 1024);
  float* bsm = reinterpret_cast<float*>(lsm + 82 * 1024);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = blockIdx.x / LS_NW, u = blockIdx.x % LS_NW, u0 = u * 8;
  const float* W = whhT + (size_t)d * Hh * G4;
  for (int idx = tid; idx < 16 * 64; idx += 256) {
    const int kb = idx >> 6, l = idx & 63, c = l & 31;
    const int col = (c >> 3) * Hh + u0 + (c & 7);
    unsigned short hv[8][3];
#pragma unroll
    for (int j = 0; j < 8; ++j) ls_split3(W[(size_t)(kb * 16 + (l >> 5) * 8 + j) * G4 + col], hv[j]);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      unsigned short* dst = wfr + ((size_t)(t * 16 + kb) * 64 + l) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) dst[j] = hv[j][t];
    }
  }
  for (int i = tid; i < 64 * 8; i += 256) cst[i] = 0.f;
  if (tid < 32) bsm[tid] = bhh ? bhh[(size_t)d * G4 + (tid >> 3) * Hh + u0 + (tid & 7)] : 0.f;
  const unsigned ep = sync[4 + d] & 0x7ffu;         // written by the previous launch on this stream (kernel boundary: visible)
  __syncthreads();
  const int rb = wave & 1, kh = wave >> 1;
  const unsigned dir_bytes = 2u * 16 * 4 * 64 * 16, par_bytes = 2 * dir_bytes;
  const __amdgpu_buffer_rsrc_t rs = ls_rsrc(hg, 2 * (size_t)par_bytes);
  const int ul = tid & 7;
  const bool row_live = rb * 32 + (lane & 31) < N;   // rows past the batch are never published: their tags are not checked
  bool tmo = false;                                  // (wave-uniform) a hand-off of this wave timed out
  for (int s = 0; s < T; ++s) {
    const int t = d == 0 ? s : T - 1 - s;
    float pre[2][4];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int n = (tid >> 3) + 32 * it;
      if (n < N) {
        const float* g = G + (((size_t)n * T + t) * 2 + d) * G4 + u0 + ul;
#pragma unroll
        for (int q = 0; q < 4; ++q) pre[it][q] = g[q * Hh];
      }
    }
    float* red = red0 + (s & 1) * 4096;
    if (s > 0) {
      const unsigned want = ((ep << 5) | (unsigned)s) << 16;                       // tag of step s - 1's data = (epoch, (s - 1) + 1)
      // (rows past the batch are never published and stay zero: their lanes ask for an out-of-range offset -- hardware zeros, no memory
      //  traffic: a quarter of the 131 KB a workgroup ingests per step at N = 48)
      const unsigned base = row_live ? (unsigned)(((s - 1) & 1) * par_bytes + d * dir_bytes) + ((unsigned)(rb * 16 + kh * 8) * 4 * 64 + lane) * 16u
                                     : 0x7ff00000u;
      ls_u32x4 raw[8][4];
      int tries = 0;
      while (true) {
        // (the compiler treats buffer loads as ordinary reads: without this barrier it hoists all 32 of them OUT of the retry loop
        //  and the loop only sleeps -- seen in the ISA of the first version, which timed out on every hand-off)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            raw[i][q] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(base + (unsigned)((i * 4 + q) * 64) * 16u), 0, LS_SC1);
        unsigned bad = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) bad |= (raw[i][q].y ^ want) | (raw[i][q].w ^ want);
        const bool ok = !row_live || (bad >> 16) == 0;
        if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
        __builtin_amdgcn_s_sleep(2);
        ++tries;
        // ~1 s without progress (or somebody else gave up): flag it, stop waiting for good, and poison what this wave computes
        if (tmo || tries > LS_TRIES || ((tries & 255) == 0 && __hip_atomic_load(sync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
          if (lane == 0) __hip_atomic_store(sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          tmo = true;
          break;
        }
      }
      // (round 6, measured: products issued chunk by chunk INSIDE the attempt, behind the compiler's partial vmcnt waits, so that the MFMA
      //  pipe time hides under the arrival of the 96 KB -- 125.8 us per launch against 120.0: a thrown-away attempt gets longer by its
      //  products, and how soon the NEXT attempt starts is what the step waits for.  An attempt that re-loads only the chunks it found
      //  stale: 119.7, no difference -- attempts fail whole, not in part.  tools/lab/lstm_seq_time.py)
      floatx16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        ls_u32x4 av[3];
#pragma unroll
        for (int q = 0; q < 4; ++q) {       // granule pair (x, y), (z, w): x / z = h1 | h2 << 16, y / w = h3 | tag << 16
          av[0][q] = __builtin_amdgcn_perm(raw[i][q].z, raw[i][q].x, 0x05040100u);
          av[1][q] = __builtin_amdgcn_perm(raw[i][q].z, raw[i][q].x, 0x07060302u);
          av[2][q] = __builtin_amdgcn_perm(raw[i][q].w, raw[i][q].y, 0x05040100u);
        }
        ls_bf16x8 a[3], b[3];
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
          a[tt] = __builtin_bit_cast(ls_bf16x8, av[tt]);
          b[tt] = *reinterpret_cast<const ls_bf16x8*>(wfr + ((size_t)(tt * 16 + kh * 8 + i) * 64 + lane) * 8);
        }
        acc = ls_mfma6(a, b, acc);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        red[(kh * 64 + row) * 32 + (lane & 31)] = acc[r];
      }
      __syncthreads();      // the only barrier of a step (red is double-buffered by step parity)
    }
    const unsigned tagw = ((ep << 5) | (unsigned)(s + 1)) << 16;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int n = (tid >> 3) + 32 * it;
      unsigned glo = 0, ghi = 0;
      float gv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (n < N) {
        const int item = n * 8 + ul;
        float p[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          p[q] = pre[it][q] + bsm[q * 8 + ul];
          if (s > 0) p[q] += red[n * 32 + q * 8 + ul] + red[(64 + n) * 32 + q * 8 + ul];
        }
        const float ig = gru_sigmoid1(p[0]), fg = gru_sigmoid1(p[1]), gg = gru_tanh(p[2]), og = gru_sigmoid1(p[3]);
        const float c = fg * cst[item] + ig * gg;
        const float h = og * gru_tanh(c);
        cst[item] = c;
        unsigned short hv[3];
        ls_split3(h, hv);
        glo = (unsigned)hv[0] | ((unsigned)hv[1] << 16);
        ghi = (unsigned)hv[2] | tagw;
        gv[0] = ig; gv[1] = fg; gv[2] = gg; gv[3] = og; gv[4] = c; gv[5] = h;
      }
      if (s + 1 < T) {      // publish FIRST: the even unit of a pair stores both granules (16 bytes, write-through), nothing waits for it --
        // the 31 other workgroups of the direction wait for exactly this store; the step's six bookkeeping stores (activated gates, cell
        // state, output) queue up behind it instead of in front of it (round 6)
        const unsigned nlo = __shfl_xor(glo, 1), nhi = __shfl_xor(ghi, 1);
        if (n < N && !(ul & 1)) {
          ls_u32x4 v;
          v.x = glo; v.y = ghi; v.z = nlo; v.w = nhi;
          const unsigned off = (unsigned)((s & 1) * par_bytes + d * dir_bytes) +
                               ((((unsigned)((n >> 5) * 16 + (u >> 1)) * 4 + (ul >> 1)) * 64) + (u & 1) * 32 + (n & 31)) * 16u;
          __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)off, 0, LS_SC1);
        }
      }
      if (n < N) {
        const int unit = u0 + ul;
        float* g = G + (((size_t)n * T + t) * 2 + d) * G4 + unit;
        g[0] = gv[0];
        g[Hh] = gv[1];
        g[2 * Hh] = gv[2];
        g[3 * Hh] = gv[3];
        Cst[(((size_t)n * T + t) * 2 + d) * Hh + unit] = gv[4];
        out[((size_t)n * T + t) * 2 * Hh + d * Hh + unit] = tmo ? __builtin_nanf("") : gv[5];
      }
    }
  }
  if (u == 0 && tid == 0) sync[4 + d] = ep + 1;     // this workgroup is done => everybody of direction d has read the epoch long ago
}

extern "C" long long tpgsr_lstm_seq_hg_bytes(void) { return 2ll * 2 * 2 * 16 * 4 * 64 * 16; }

extern "C" int tpgsr_lstm_seq_fwdg(float* G, const float* whhT, const float* bhh, float* Cst, float* out, void* hg, unsigned* sync, int N,
                                   int T, int Hh, void* stream) {
  TPGSR_CHECK_ARG(G && whhT && Cst && out && hg && sync && N > 0 && N <= 64 && T > 0 && T <= 31 && Hh == 256,
                  "tpgsr_lstm_seq_fwdg: needs Hh == 256, 1 <= N <= 64, 1 <= T <= 31 and non-null buffers (got Hh %d, N %d, T %d)", Hh, N, T);
  hipLaunchKernelGGL(lstm_seq_fwdg_kernel, dim3(2 * LS_NW), dim3(256), 0, (hipStream_t)stream, G, whhT, bhh, Cst, out, (unsigned*)hg, sync, N, T);
  TPGSR_LAUNCH_CHECK("tpgsr_lstm_seq_fwdg");
}

// ------------------------------------------------------------------------------------------------------
// backward (BPTT).  Step s' of direction d processes t = T-1-s' (d = 0) / t = s' (d = 1), the reverse of the forward order.
//   G    [N][T][2][4Hh]  activated gates in, gate (pre-activation) gradients dG out -- what the weight-gradient and input-projection
//                        data-gradient launches after the recurrence read
//   Cst  [N][T][2][Hh], dout [N][T][2Hh] = dL/dh from the embedding's data gradient
//   w0, w1 [4Hh][Hh]     W_hh of the two directions as PyTorch stores them (rows = gate columns)
//   px   [2 parity][2 dir][32 consumers][32 producers][8 units][64 rows] f32 (4 MB); no initialisation needed
//   sync [4] u32 as above
// Dynamic LDS: W fragments [3][8 n-blocks][2 k-blocks][64][8] bf16 48 KB | P staging [256 units][64 rows + 4] f32 68 KB (rows padded:
//   the 32 units of a wave's 16-byte stores land in distinct bank quads) | A fragments [3][2 row blocks][2 k-blocks][64][8] bf16 12 KB |
//   recurrent sums [2 groups][8][64] f32 4 KB | dc [64][8] f32 2 KB
// ------------------------------------------------------------------------------------------------------
#define LSB_PLD 68
#define LSB_LDS (48 * 1024 + 256 * LSB_PLD * 4 + 12 * 1024 + 4 * 1024 + 2 * 1024)
__global__ __launch_bounds__(256) void lstm_seq_bwd_kernel(float* __restrict__ G, const float* __restrict__ Cst,
                                                           const float* __restrict__ dout, const float* __restrict__ w0,
                                                           const float* __restrict__ w1, float* __restrict__ px,
                                                           unsigned* __restrict__ sync, int N, int T) {
  constexpr int Hh = 256, G4 = 1024;
  extern __shared__ __attribute__((aligned(16))) unsigned char lsb[];
  unsigned short* wfr = reinterpret_cast<unsigned short*>(lsb);
  float* pst = reinterpret_cast<float*>(lsb + 48 * 1024);
  unsigned short* afr = reinterpret_cast<unsigned short*>(lsb + 48 * 1024 + 256 * LSB_PLD * 4);
  float* rsum = reinterpret_cast<float*>(lsb + 48 * 1024 + 256 * LSB_PLD * 4 + 12 * 1024);
  float* dcc = reinterpret_cast<float*>(lsb + 48 * 1024 + 256 * LSB_PLD * 4 + 16 * 1024);
  __shared__ int tmo_s;       // a hand-off of this workgroup timed out: everything it produces from then on is NaN
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = blockIdx.x / LS_NW, u = blockIdx.x % LS_NW, u0 = u * 8;
  const float* W = d == 0 ? w0 : w1;
  if (tid == 0) tmo_s = 0;
  // B operand of (n-block nb, k-block kb): lane l holds unit nb 32 + (l & 31), k = 16 kb + 8 (l >> 5) + j <-> gate column (k >> 3) Hh + u0 + (k & 7)
  for (int idx = tid; idx < 8 * 2 * 64; idx += 256) {
    const int nb = idx >> 7, kb = (idx >> 6) & 1, l = idx & 63;
    const int unit = nb * 32 + (l & 31);
    unsigned short hv[8][3];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = kb * 16 + (l >> 5) * 8 + j;
      ls_split3(W[(size_t)((k >> 3) * Hh + u0 + (k & 7)) * Hh + unit], hv[j]);
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      unsigned short* dst = wfr + ((size_t)((t * 8 + nb) * 2 + kb) * 64 + l) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) dst[j] = hv[j][t];
    }
  }
  for (int i = tid; i < 3 * 2 * 2 * 64 * 8 / 2; i += 256) reinterpret_cast<unsigned*>(afr)[i] = 0u;   // rows >= N stay zero
  for (int i = tid; i < 64 * 8; i += 256) dcc[i] = 0.f;
  __syncthreads();
  const int rb = wave & 1, nbh = wave >> 1;
  const size_t dir_f = (size_t)32 * 32 * 8 * 64, par_f = 2 * dir_f;    // floats
  const __amdgpu_buffer_rsrc_t rs_px = ls_rsrc(px, 2 * par_f * 4);
  const int ul = tid & 7;
  for (int s = 0; s < T; ++s) {
    const int t = d == 0 ? T - 1 - s : s;
    const int tp = d == 0 ? t - 1 : t + 1;           // previous state in this direction's forward order
    const bool has_prev = d == 0 ? t > 0 : t < T - 1;
    // everything of this step that does not depend on the exchange
    float gt[2][4], cc[2], cp[2], dh[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int n = (tid >> 3) + 32 * it;
      if (n < N) {
        const float* g = G + (((size_t)n * T + t) * 2 + d) * G4 + u0 + ul;
#pragma unroll
        for (int q = 0; q < 4; ++q) gt[it][q] = g[q * Hh];
        cc[it] = Cst[(((size_t)n * T + t) * 2 + d) * Hh + u0 + ul];
        cp[it] = has_prev ? Cst[(((size_t)n * T + tp) * 2 + d) * Hh + u0 + ul] : 0.f;
        dh[it] = dout[((size_t)n * T + t) * 2 * Hh + d * Hh + u0 + ul];
      }
    }
    if (s > 0) {
      if (tid == 0 && !tmo_s && ls_wait(sync + d, (unsigned)s * LS_NW, sync + 2)) tmo_s = 1;
      __syncthreads();
      // gather: my region [32 producers][8 units][64 rows]; thread = (producer group g of 16, unit gu, row quad rq)
      const int g = tid >> 7, gu = (tid >> 4) & 7, rq = tid & 15;
      float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
      if (rq * 4 < N) {
        const unsigned base = (unsigned)((((s - 1) & 1) * par_f + d * dir_f + (size_t)u * 32 * 8 * 64) * 4);
        ls_u32x4 v[16];
#pragma unroll
        for (int p = 0; p < 16; ++p)
          v[p] = __builtin_amdgcn_raw_buffer_load_b128(rs_px, (int)(base + ((((unsigned)(g * 16 + p) * 8 + gu) * 64 + rq * 4) * 4)), 0, LS_SC1);
#pragma unroll
        for (int p = 0; p < 16; ++p) {     // producer order: deterministic
          sum.x += __uint_as_float(v[p].x);
          sum.y += __uint_as_float(v[p].y);
          sum.z += __uint_as_float(v[p].z);
          sum.w += __uint_as_float(v[p].w);
        }
      }
      *reinterpret_cast<float4*>(rsum + ((g * 8 + gu) * 64 + rq * 4)) = sum;
      __syncthreads();
    }
    // cell backward: item = (sequence n, local unit ul); the gate gradients go to G and, split, into the A fragments of the next step
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int n = (tid >> 3) + 32 * it;
      if (n < N) {
        const int item = n * 8 + ul;
        const float ig = gt[it][0], fg = gt[it][1], gg = gt[it][2], og = gt[it][3];
        float dhh = dh[it], dc = 0.f;
        if (s > 0) {
          dhh += rsum[ul * 64 + n] + rsum[(8 + ul) * 64 + n];
          dc = dcc[item];
        }
        const float tc = gru_tanh(cc[it]);
        const float dog = dhh * tc * og * (1.f - og);
        dc += dhh * og * (1.f - tc * tc);
        const float dig = dc * gg * ig * (1.f - ig);
        const float dfg = dc * cp[it] * fg * (1.f - fg);
        const float dgg = dc * ig * (1.f - gg * gg);
        dcc[item] = dc * fg;
        float* g = G + (((size_t)n * T + t) * 2 + d) * G4 + u0 + ul;
        const float po = tmo_s ? __builtin_nanf("") : 0.f;      // a timed-out hand-off: NaN gate gradients (the stores only)
        g[0] = tmo_s ? po : dig;
        g[Hh] = tmo_s ? po : dfg;
        g[2 * Hh] = tmo_s ? po : dgg;
        g[3 * Hh] = tmo_s ? po : dog;
        const float dq[4] = {dig, dfg, dgg, dog};
#pragma unroll
        for (int q = 0; q < 4; ++q) {       // k = 8 q + ul: k-block q >> 1, fragment lane (q & 1) 32 + (n & 31), element ul
          unsigned short hv[3];
          ls_split3(dq[q], hv);
#pragma unroll
          for (int tt = 0; tt < 3; ++tt)
            afr[((size_t)((tt * 2 + (n >> 5)) * 2 + (q >> 1)) * 64 + (q & 1) * 32 + (n & 31)) * 8 + ul] = hv[tt];
        }
      }
    }
    if (s + 1 < T) {
      __syncthreads();
      // P_u[n][256] = dG_u[n][32] W_hh[cols_u][256]: wave (row block rb, n-blocks 4 nbh .. 4 nbh + 3)
      ls_bf16x8 a[2][3];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) a[kb][tt] = *reinterpret_cast<const ls_bf16x8*>(afr + ((size_t)((tt * 2 + rb) * 2 + kb) * 64 + lane) * 8);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int nb = nbh * 4 + i;
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          ls_bf16x8 b[3];
#pragma unroll
          for (int tt = 0; tt < 3; ++tt) b[tt] = *reinterpret_cast<const ls_bf16x8*>(wfr + ((size_t)((tt * 8 + nb) * 2 + kb) * 64 + lane) * 8);
          acc = ls_mfma6(a[kb], b, acc);
        }
        // staging image [unit][row]: a lane holds 4 consecutive rows of one unit per register quad
        const int unit = nb * 32 + (lane & 31);
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int row = rb * 32 + 8 * rg + 4 * (lane >> 5);
          *reinterpret_cast<float4*>(pst + unit * LSB_PLD + row) = make_float4(acc[4 * rg], acc[4 * rg + 1], acc[4 * rg + 2], acc[4 * rg + 3]);
        }
      }
      __syncthreads();
      // publish, coalesced: piece o = (unit, row quad); consumer unit >> 3 gets [producer u][unit & 7][row] contiguous
      const unsigned wbase = (unsigned)(((s & 1) * par_f + d * dir_f) * 4);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int o = tid + 256 * i, unit = o >> 4, rq = o & 15;
        if (rq * 4 < N) {
          const ls_u32x4 v = *reinterpret_cast<const ls_u32x4*>(pst + unit * LSB_PLD + rq * 4);
          const unsigned off = wbase + (((((unsigned)(unit >> 3) * 32 + u) * 8 + (unit & 7)) * 64 + rq * 4) * 4);
          __builtin_amdgcn_raw_buffer_store_b128(v, rs_px, (int)off, 0, LS_SC1);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(sync + d, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// backward with the DATA-TAGGED hand-off (as lstm_seq_fwdg_kernel): every partial sum travels as an 8-byte granule {fp32 value, tag =
// (launch epoch, step)}; a consumer re-loads its 96 KB until every tag is this step's.  Twice the bytes of the counter form, but no
// wait for the write-through acknowledgement, no arrival counter and no poll on the per-step chain.
//   pg   [2 parity][2 dir][32 consumers][32 producers][8 units][64 rows] granules (16 MB), zeroed ONCE by the caller;
//   sync [8] u32 as for the forward kernel (epoch in sync[4 + d]), zeroed ONCE by the caller.   T < 256.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lstm_seq_bwdg_kernel(float* __restrict__ G, const float* __restrict__ Cst,
                                                           const float* __restrict__ dout, const float* __restrict__ w0,
                                                           const float* __restrict__ w1, unsigned* __restrict__ pg,
                                                           unsigned* __restrict__ sync, int N, int T) {
  constexpr int Hh = 256, G4 = 1024;
  extern __shared__ __attribute__((aligned(16))) unsigned char lsb[];
  unsigned short* wfr = reinterpret_cast<unsigned short*>(lsb);
  float* pst = reinterpret_cast<float*>(lsb + 48 * 1024);
  unsigned short* afr = reinterpret_cast<unsigned short*>(lsb + 48 * 1024 + 256 * LSB_PLD * 4);
  float* rsum = reinterpret_cast<float*>(lsb + 48 * 1024 + 256 * LSB_PLD * 4 + 12 * 1024);
  float* dcc = reinterpret_cast<float*>(lsb + 48 * 1024 + 256 * LSB_PLD * 4 + 16 * 1024);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = blockIdx.x / LS_NW, u = blockIdx.x % LS_NW, u0 = u * 8;
  const float* W = d == 0 ? w0 : w1;
  // B operand of (n-block nb, k-block kb): lane l holds unit nb 32 + (l & 31), k = 16 kb + 8 (l >> 5) + j <-> gate column (k >> 3) Hh + u0 + (k & 7)
  for (int idx = tid; idx < 8 * 2 * 64; idx += 256) {
    const int nb = idx >> 7, kb = (idx >> 6) & 1, l = idx & 63;
    const int unit = nb * 32 + (l & 31);
    unsigned short hv[8][3];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = kb * 16 + (l >> 5) * 8 + j;
      ls_split3(W[(size_t)((k >> 3) * Hh + u0 + (k & 7)) * Hh + unit], hv[j]);
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      unsigned short* dst = wfr + ((size_t)((t * 8 + nb) * 2 + kb) * 64 + l) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) dst[j] = hv[j][t];
    }
  }
  for (int i = tid; i < 3 * 2 * 2 * 64 * 8 / 2; i += 256) reinterpret_cast<unsigned*>(afr)[i] = 0u;   // rows >= N stay zero
  for (int i = tid; i < 64 * 8; i += 256) dcc[i] = 0.f;
  __syncthreads();
  const int rb = wave & 1, nbh = wave >> 1;
  const size_t dir_f = (size_t)32 * 32 * 8 * 64, par_f = 2 * dir_f;    // granules of 8 bytes {partial sum, tag}
  const __amdgpu_buffer_rsrc_t rs_px = ls_rsrc(pg, 2 * par_f * 8);
  const unsigned ep = sync[4 + d] & 0xffffffu;
  const int ul = tid & 7;
  bool tmo = false;                                  // (wave-uniform) a hand-off of this wave timed out
  for (int s = 0; s < T; ++s) {
    const int t = d == 0 ? T - 1 - s : s;
    const int tp = d == 0 ? t - 1 : t + 1;           // previous state in this direction's forward order
    const bool has_prev = d == 0 ? t > 0 : t < T - 1;
    // everything of this step that does not depend on the exchange
    float gt[2][4], cc[2], cp[2], dh[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int n = (tid >> 3) + 32 * it;
      if (n < N) {
        const float* g = G + (((size_t)n * T + t) * 2 + d) * G4 + u0 + ul;
#pragma unroll
        for (int q = 0; q < 4; ++q) gt[it][q] = g[q * Hh];
        cc[it] = Cst[(((size_t)n * T + t) * 2 + d) * Hh + u0 + ul];
        cp[it] = has_prev ? Cst[(((size_t)n * T + tp) * 2 + d) * Hh + u0 + ul] : 0.f;
        dh[it] = dout[((size_t)n * T + t) * 2 * Hh + d * Hh + u0 + ul];
      }
    }
    if (s > 0) {
      // gather: my region [32 producers][8 units][64 rows] of granules; thread = (producer group g of 16, unit gu, row quad rq).
      // The data is its own flag: re-load until every granule carries (epoch, step s - 1)'s tag.
      const int g = tid >> 7, gu = (tid >> 4) & 7, rq = tid & 15;
      float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
      const bool live = rq * 4 < N;
      const unsigned want = (ep << 8) | (unsigned)s;
      const unsigned base = (unsigned)((((s - 1) & 1) * par_f + d * dir_f + (size_t)u * 32 * 8 * 64) * 8);
      ls_u32x4 v[16][2];
      int tries = 0;
      while (true) {
        asm volatile("" ::: "memory");      // keeps the loads inside the retry loop (see lstm_seq_fwdg_kernel)
        if (live) {
#pragma unroll
          for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
              v[p][hh] = __builtin_amdgcn_raw_buffer_load_b128(
                  rs_px, (int)(base + ((((unsigned)(g * 16 + p) * 8 + gu) * 64 + rq * 4 + hh * 2) * 8)), 0, LS_SC1);
        }
        unsigned bad = 0;
        if (live) {
#pragma unroll
          for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) bad |= (v[p][hh].y ^ want) | (v[p][hh].w ^ want);
        }
        if (__builtin_amdgcn_ballot_w64(bad != 0) == 0) break;
        __builtin_amdgcn_s_sleep(2);
        ++tries;
        if (tmo || tries > LS_TRIES || ((tries & 255) == 0 && __hip_atomic_load(sync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
          if (lane == 0) __hip_atomic_store(sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          tmo = true;
          break;
        }
      }
      if (live) {
#pragma unroll
        for (int p = 0; p < 16; ++p) {     // producer order: deterministic
          sum.x += __uint_as_float(v[p][0].x);
          sum.y += __uint_as_float(v[p][0].z);
          sum.z += __uint_as_float(v[p][1].x);
          sum.w += __uint_as_float(v[p][1].z);
        }
      }
      *reinterpret_cast<float4*>(rsum + ((g * 8 + gu) * 64 + rq * 4)) = sum;
      __syncthreads();
    }
    // cell backward: item = (sequence n, local unit ul); the gate gradients go to G and, split, into the A fragments of the next step
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int n = (tid >> 3) + 32 * it;
      if (n < N) {
        const int item = n * 8 + ul;
        const float ig = gt[it][0], fg = gt[it][1], gg = gt[it][2], og = gt[it][3];
        float dhh = dh[it], dc = 0.f;
        if (s > 0) {
          dhh += rsum[ul * 64 + n] + rsum[(8 + ul) * 64 + n];
          dc = dcc[item];
        }
        const float tc = gru_tanh(cc[it]);
        const float dog = dhh * tc * og * (1.f - og);
        dc += dhh * og * (1.f - tc * tc);
        const float dig = dc * gg * ig * (1.f - ig);
        const float dfg = dc * cp[it] * fg * (1.f - fg);
        const float dgg = dc * ig * (1.f - gg * gg);
        dcc[item] = dc * fg;
        float* g = G + (((size_t)n * T + t) * 2 + d) * G4 + u0 + ul;
        const float po = tmo ? __builtin_nanf("") : 0.f;
        g[0] = tmo ? po : dig;
        g[Hh] = tmo ? po : dfg;
        g[2 * Hh] = tmo ? po : dgg;
        g[3 * Hh] = tmo ? po : dog;
        const float dq[4] = {dig, dfg, dgg, dog};
#pragma unroll
        for (int q = 0; q < 4; ++q) {       // k = 8 q + ul: k-block q >> 1, fragment lane (q & 1) 32 + (n & 31), element ul
          unsigned short hv[3];
          ls_split3(dq[q], hv);
#pragma unroll
          for (int tt = 0; tt < 3; ++tt)
            afr[((size_t)((tt * 2 + (n >> 5)) * 2 + (q >> 1)) * 64 + (q & 1) * 32 + (n & 31)) * 8 + ul] = hv[tt];
        }
      }
    }
    if (s + 1 < T) {
      __syncthreads();
      // P_u[n][256] = dG_u[n][32] W_hh[cols_u][256]: wave (row block rb, n-blocks 4 nbh .. 4 nbh + 3)
      ls_bf16x8 a[2][3];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) a[kb][tt] = *reinterpret_cast<const ls_bf16x8*>(afr + ((size_t)((tt * 2 + rb) * 2 + kb) * 64 + lane) * 8);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int nb = nbh * 4 + i;
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          ls_bf16x8 b[3];
#pragma unroll
          for (int tt = 0; tt < 3; ++tt) b[tt] = *reinterpret_cast<const ls_bf16x8*>(wfr + ((size_t)((tt * 8 + nb) * 2 + kb) * 64 + lane) * 8);
          acc = ls_mfma6(a[kb], b, acc);
        }
        // staging image [unit][row]: a lane holds 4 consecutive rows of one unit per register quad
        const int unit = nb * 32 + (lane & 31);
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int row = rb * 32 + 8 * rg + 4 * (lane >> 5);
          *reinterpret_cast<float4*>(pst + unit * LSB_PLD + row) = make_float4(acc[4 * rg], acc[4 * rg + 1], acc[4 * rg + 2], acc[4 * rg + 3]);
        }
      }
      __syncthreads();
      // publish, coalesced: piece o = (unit, row quad) = four tagged granules = two 16-byte write-through stores; nothing waits for them
      const unsigned wbase = (unsigned)(((s & 1) * par_f + d * dir_f) * 8);
      const unsigned tagv = (ep << 8) | (unsigned)(s + 1);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int o = tid + 256 * i, unit = o >> 4, rq = o & 15;
        if (rq * 4 < N) {
          const ls_u32x4 v = *reinterpret_cast<const ls_u32x4*>(pst + unit * LSB_PLD + rq * 4);
          const unsigned off = wbase + (((((unsigned)(unit >> 3) * 32 + u) * 8 + (unit & 7)) * 64 + rq * 4) * 8);
          ls_u32x4 lo, hi;
          lo.x = v.x; lo.y = tagv; lo.z = v.y; lo.w = tagv;
          hi.x = v.z; hi.y = tagv; hi.z = v.w; hi.w = tagv;
          __builtin_amdgcn_raw_buffer_store_b128(lo, rs_px, (int)off, 0, LS_SC1);
          __builtin_amdgcn_raw_buffer_store_b128(hi, rs_px, (int)(off + 16), 0, LS_SC1);
        }
      }
    }
  }
  if (u == 0 && tid == 0) sync[4 + d] = ep + 1;     // see lstm_seq_fwdg_kernel
}


extern "C" long long tpgsr_lstm_seq_pg_bytes(void) { return 2ll * 2 * 32 * 32 * 8 * 64 * 8; }

extern "C" int tpgsr_lstm_seq_bwdg(float* G, const float* Cst, const float* dout, const float* w0, const float* w1, void* pg,
                                   unsigned* sync, int N, int T, int Hh, void* stream) {
  TPGSR_CHECK_ARG(G && Cst && dout && w0 && w1 && pg && sync && N > 0 && N <= 64 && T > 0 && T < 256 && Hh == 256,
                  "tpgsr_lstm_seq_bwdg: needs Hh == 256, 1 <= N <= 64, 1 <= T <= 255 and non-null buffers (got Hh %d, N %d, T %d)", Hh, N, T);
  {   // opt in to > 64 KB of dynamic LDS, once per device
    static std::mutex mu;
    static unsigned long long done = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
      tpgsr_set_error("tpgsr_lstm_seq_bwdg: hipGetDevice failed");
      return TPGSR_ERR_LAUNCH;
    }
    std::lock_guard<std::mutex> lock(mu);
    if (!(done >> dev & 1ull)) {
      if (hipFuncSetAttribute((const void*)lstm_seq_bwdg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LSB_LDS) != hipSuccess) {
        tpgsr_set_error("tpgsr_lstm_seq_bwdg: LDS opt-in (%d bytes) failed", LSB_LDS);
        return TPGSR_ERR_LAUNCH;
      }
      done |= 1ull << dev;
    }
  }
  hipLaunchKernelGGL(lstm_seq_bwdg_kernel, dim3(2 * LS_NW), dim3(256), LSB_LDS, (hipStream_t)stream, G, Cst, dout, w0, w1, (unsigned*)pg, sync, N,
                     T);
  TPGSR_LAUNCH_CHECK("tpgsr_lstm_seq_bwdg");
}

extern "C" long long tpgsr_lstm_seq_px_bytes(void) { return 2ll * 2 * 32 * 32 * 8 * 64 * 4; }

extern "C" int tpgsr_lstm_seq_bwd(float* G, const float* Cst, const float* dout, const float* w0, const float* w1, void* px,
                                  unsigned* sync, int N, int T, int Hh, void* stream) {
  TPGSR_CHECK_ARG(G && Cst && dout && w0 && w1 && px && sync && N > 0 && N <= 64 && T > 0 && Hh == 256,
                  "tpgsr_lstm_seq_bwd: needs Hh == 256, 1 <= N <= 64, T >= 1 and non-null buffers (got Hh %d, N %d, T %d)", Hh, N, T);
  {   // opt in to > 64 KB of dynamic LDS, once per device
    static std::mutex mu;
    static unsigned long long done = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
      tpgsr_set_error("tpgsr_lstm_seq_bwd: hipGetDevice failed");
      return TPGSR_ERR_LAUNCH;
    }
    std::lock_guard<std::mutex> lock(mu);
    if (!(done >> dev & 1ull)) {
      if (hipFuncSetAttribute((const void*)lstm_seq_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LSB_LDS) != hipSuccess) {
        tpgsr_set_error("tpgsr_lstm_seq_bwd: LDS opt-in (%d bytes) failed", LSB_LDS);
        return TPGSR_ERR_LAUNCH;
      }
      done |= 1ull << dev;
    }
  }
  if (hipMemsetAsync(sync, 0, 4 * sizeof(unsigned), (hipStream_t)stream) != hipSuccess) {
    tpgsr_set_error("tpgsr_lstm_seq_bwd: hipMemsetAsync failed");
    return TPGSR_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(lstm_seq_bwd_kernel, dim3(2 * LS_NW), dim3(256), LSB_LDS, (hipStream_t)stream, G, Cst, dout, w0, w1, (float*)px, sync, N,
                     T);
  TPGSR_LAUNCH_CHECK("tpgsr_lstm_seq_bwd");
}

// ------------------------------------------------------------------------------------------------------
// Co-residency probe (round 5).  Both persistent kernels hand data between their 2 x 32 workgroups every time
// step: they only make progress when all 64 are RESIDENT TOGETHER.  On an idle MI355X they are (64 workgroups on 256 CUs); on a GPU that
// is shared with another process, partitioned, or smaller, they may never be -- and a hand-off that times out poisons the step with NaN.
// The engines therefore ask ONCE per device, when they record their plans: 64 workgroups of 256 threads with the backward kernel's LDS
// footprint (the larger of the two: one workgroup per CU) arrive on a counter and wait until everybody has, or ~50 ms have passed.
// Returns 1 (co-resident: record the persistent launches), 0 (not: the engines record the per-step launches instead -- slower, never
// wrong) or < 0 on error.  Synchronises `stream`; never part of a recorded plan.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lstm_seq_probe_kernel(unsigned* __restrict__ words, long long budget_ticks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char probe_lds[];
  if (threadIdx.x == 0) {
    probe_lds[0] = 1;      // (touch the allocation)
    __hip_atomic_fetch_add(words, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long t0 = wall_clock64();
    bool all = false;
    while (!(all = __hip_atomic_load(words, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gridDim.x) && wall_clock64() - t0 < budget_ticks)
      __builtin_amdgcn_s_sleep(8);
    if (!all) __hip_atomic_store(words + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

extern "C" int tpgsr_lstm_seq_probe(unsigned* words /* 2 u32 of device memory */, void* stream) {
  TPGSR_CHECK_ARG(words, "tpgsr_lstm_seq_probe: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (hipFuncSetAttribute((const void*)lstm_seq_probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LSB_LDS) != hipSuccess ||
      hipMemsetAsync(words, 0, 2 * sizeof(unsigned), st) != hipSuccess) {
    tpgsr_set_error("tpgsr_lstm_seq_probe: set-up failed: %s", hipGetErrorString(hipGetLastError()));
    return TPGSR_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(lstm_seq_probe_kernel, dim3(2 * LS_NW), dim3(256), LSB_LDS, st, words, 5000000ll /* 50 ms of the 100 MHz clock */);
  unsigned host[2] = {0, 1};
  if (hipGetLastError() != hipSuccess || hipMemcpyAsync(host, words, sizeof(host), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess) {
    tpgsr_set_error("tpgsr_lstm_seq_probe: launch / read-back failed: %s", hipGetErrorString(hipGetLastError()));
    return TPGSR_ERR_LAUNCH;
  }
  return (host[0] == 2u * LS_NW && host[1] == 0u) ? 1 : 0;
}
