// Native launch-plan executor: a recorded list of C-ABI kernel launches (+ fork/join edges between the caller's stream
// and a side stream) replayed by ONE call, so a training step costs one host round trip instead of ~450 interpreter-level
// calls.  The plan stores, per launch, the entry point and its scalar/pointer arguments; pointers to argument structs
// (tpgsr_conv_args / tpgsr_wgrad_args) are copied into plan-owned storage.  Patchable pointer slots (per-step input /
// output tensors) are rewritten with tpgsr_plan_set_arg.  Capturable into a hipGraph like the launches themselves.
#include <hip/hip_runtime.h>
#include <string.h>
#include <memory>
#include <string>
#include <tuple>
#include <type_traits>
#include <unordered_map>
#include <utility>
#include <vector>
#include "../../include/tpgsr_hip.h"

void tpgsr_set_error(const char* fmt, ...);

namespace {

typedef int (*thunk_fn)(const tpgsr_plan_arg* a, void* stream);

template <typename T>
inline T arg_as(const tpgsr_plan_arg& a) {
  if constexpr (std::is_pointer<T>::value) return reinterpret_cast<T>(const_cast<void*>(a.p));
  else if constexpr (std::is_floating_point<T>::value) return static_cast<T>(a.f);
  else return static_cast<T>(a.i);
}

// every launch entry point is int f(P0, ..., Pn-1, void* stream): unpack P_i from a[i], pass the stream last
template <typename Tuple, typename F, size_t... I>
inline int call_with(F fn, const tpgsr_plan_arg* a, void* st, std::index_sequence<I...>) {
  return fn(arg_as<typename std::tuple_element<I, Tuple>::type>(a[I])..., st);
}

template <typename F, F fn>
struct Thunk;
template <typename... P, int (*fn)(P...)>
struct Thunk<int (*)(P...), fn> {
  typedef std::tuple<P...> Tuple;
  static constexpr int nargs = (int)sizeof...(P) - 1;
  static_assert(std::is_same<typename std::tuple_element<sizeof...(P) - 1, Tuple>::type, void*>::value,
                "launch entry points end with the stream");
  static int call(const tpgsr_plan_arg* a, void* st) {
    return call_with<Tuple>(fn, a, st, std::make_index_sequence<sizeof...(P) - 1>{});
  }
};

struct Entry {
  thunk_fn fn;
  int nargs;
  int struct_bytes;   // > 0: argument 0 points to a struct of this size (copied into the plan)
};

#define TPGSR_REG(sym) {#sym, {&Thunk<decltype(&sym), &sym>::call, Thunk<decltype(&sym), &sym>::nargs, 0}}
#define TPGSR_REG_S(sym, T) {#sym, {&Thunk<decltype(&sym), &sym>::call, Thunk<decltype(&sym), &sym>::nargs, (int)sizeof(T)}}

const std::unordered_map<std::string, Entry>& registry() {
  static const std::unordered_map<std::string, Entry> r = {
      TPGSR_REG_S(tpgsr_conv_fwd, tpgsr_conv_args), TPGSR_REG_S(tpgsr_conv_wgrad, tpgsr_wgrad_args),
      TPGSR_REG(tpgsr_wgrad_reduce), TPGSR_REG(tpgsr_wgrad_reduce_program), TPGSR_REG(tpgsr_conv_wgrad_batch), TPGSR_REG(tpgsr_compose_bwd_program), TPGSR_REG(tpgsr_pack_conv_weight), TPGSR_REG(tpgsr_pack_tail_weight),
      TPGSR_REG(tpgsr_pack_program), TPGSR_REG(tpgsr_mfma_probe), TPGSR_REG(tpgsr_copy), TPGSR_REG(tpgsr_zero),
      TPGSR_REG(tpgsr_bn_finalize), TPGSR_REG(tpgsr_bn_stats), TPGSR_REG(tpgsr_bn_bwd_reduce),
      TPGSR_REG(tpgsr_bn_bwd_finalize), TPGSR_REG(tpgsr_bn_bwd_apply), TPGSR_REG(tpgsr_affine_act),
      TPGSR_REG_S(tpgsr_affine_act_bnd, tpgsr_bn_derive), TPGSR_REG_S(tpgsr_affine_act_pool_bnd, tpgsr_bn_derive), TPGSR_REG_S(tpgsr_bn_bwd_apply_bnd, tpgsr_bn_derive),
      TPGSR_REG(tpgsr_affine_act_pool), TPGSR_REG(tpgsr_affine_act_pool_bwd), TPGSR_REG(tpgsr_prelu_fwd),
      TPGSR_REG(tpgsr_prelu_bwd), TPGSR_REG(tpgsr_add), TPGSR_REG(tpgsr_act_bwd), TPGSR_REG(tpgsr_nchw_to_nhwc),
      TPGSR_REG(tpgsr_nhwc_to_nchw), TPGSR_REG(tpgsr_reduce_partials), TPGSR_REG(tpgsr_bigru_fwd),
      TPGSR_REG(tpgsr_bigru_bwd), TPGSR_REG(tpgsr_bigru_bwd2), TPGSR_REG_S(tpgsr_bigru_proj_fwd, tpgsr_bigru_proj_args), TPGSR_REG_S(tpgsr_gru_wgrad, tpgsr_gru_wgrad_args), TPGSR_REG(tpgsr_tps_grid_fwd), TPGSR_REG(tpgsr_tps_grid_bwd),
      TPGSR_REG(tpgsr_grid_sample_fwd), TPGSR_REG(tpgsr_grid_sample_bwd), TPGSR_REG(tpgsr_strip_resample_fwd),
      TPGSR_REG(tpgsr_strip_resample_bwd), TPGSR_REG(tpgsr_hsum), TPGSR_REG(tpgsr_bicubic_gray_fwd),
      TPGSR_REG(tpgsr_bicubic_gray_bwd), TPGSR_REG(tpgsr_pool2d_fwd), TPGSR_REG(tpgsr_pool2d_bwd),
      TPGSR_REG(tpgsr_lstm_rec_gemm), TPGSR_REG(tpgsr_lstm_step_fwd), TPGSR_REG(tpgsr_lstm_seq_fwd), TPGSR_REG(tpgsr_lstm_seq_fwdg), TPGSR_REG(tpgsr_lstm_seq_bwd), TPGSR_REG(tpgsr_lstm_seq_bwdg), TPGSR_REG(tpgsr_lstm_wfrag), TPGSR_REG(tpgsr_lstm_stepx_fwd), TPGSR_REG(tpgsr_lstm_step_bwd), TPGSR_REG(tpgsr_softmax_prior_fwd),
      TPGSR_REG(tpgsr_semantic_loss_finalize), TPGSR_REG(tpgsr_softmax_prior_bwd), TPGSR_REG(tpgsr_tail_shiftsum_tanh), TPGSR_REG(tpgsr_shiftsum_nhwc),
      TPGSR_REG(tpgsr_tail_bwd), TPGSR_REG(tpgsr_image_loss_fwd), TPGSR_REG(tpgsr_image_loss_finalize),
      TPGSR_REG(tpgsr_image_loss_bwd), TPGSR_REG(tpgsr_sumsq_partial), TPGSR_REG(tpgsr_clip_coef),
      TPGSR_REG(tpgsr_adam_step), TPGSR_REG(tpgsr_step_inc), TPGSR_REG(tpgsr_scale_),
      TPGSR_REG(tpgsr_im2col3x3_c1), TPGSR_REG(tpgsr_col2im3x3_c1), TPGSR_REG(tpgsr_pad_channels),
      TPGSR_REG(tpgsr_semantic_loss_fwd), TPGSR_REG(tpgsr_semantic_loss_bwd), TPGSR_REG(tpgsr_split_bf_program),
      TPGSR_REG(tpgsr_copy_strided), TPGSR_REG(tpgsr_resize_nearest_fwd), TPGSR_REG(tpgsr_resize_nearest_bwd), TPGSR_REG(tpgsr_resize_bilinear_fwd),
      TPGSR_REG(tpgsr_resize_bilinear_bwd), TPGSR_REG(tpgsr_dilate2d), TPGSR_REG(tpgsr_subsample2d), TPGSR_REG(tpgsr_hreduce), TPGSR_REG(tpgsr_hbroadcast),
  };
  return r;
}

enum { OP_LAUNCH = 0, OP_FORK = 1, OP_JOIN = 2, OP_EDGE = 3 };   // OP_EDGE: stream `src` -> stream `dst` (sid = src, nargs = dst)
constexpr int MAX_ARGS = 24;

struct Op {
  int kind, sid, nargs;
  thunk_fn fn;
  const char* name;       // registry key (static storage)
  tpgsr_plan_arg args[MAX_ARGS];
  int blob;               // index into Plan::blobs of the copied argument struct, or -1
  hipEvent_t ev;
};

struct Plan {
  std::vector<Op> ops;
  std::vector<std::unique_ptr<char[]>> blobs;
  std::vector<hipEvent_t> stamps;   // stamp mode: one timing event per op (recorded behind it on its stream)
  std::vector<void*> stamp_stream;  // ... and the stream the op ran on in the last stamped run
  bool stamped = false;             // the last run recorded them
  ~Plan() {
    for (auto& o : ops)
      if (o.ev) (void)hipEventDestroy(o.ev);
    for (auto e : stamps)
      if (e) (void)hipEventDestroy(e);
  }
};

// ---- execution modes of tpgsr_plan_run3 (process-wide; tests and diagnostics, see tpgsr_plan_set_mode in the header) ----
struct Mode {
  int serial = 0;            // every launch on the caller's stream in recording order, stream edges dropped
  int fuzz_max_us = 0;       // > 0: a spin kernel of random length (0 .. fuzz_max_us) around every stream edge
  int noise_blocks = 0;      // > 0: ... and a co-running busy kernel of that many workgroups on a stream of its own
  unsigned long long rng = 0x9E3779B97F4A7C15ull;
  int stamp = 0;
  hipStream_t noise_stream = nullptr;
  hipEvent_t epoch = nullptr;
} g_mode;

inline unsigned long long rng_next() {   // xorshift64*
  unsigned long long x = g_mode.rng;
  x ^= x >> 12;
  x ^= x << 25;
  x ^= x >> 27;
  g_mode.rng = x;
  return x * 0x2545F4914F6CDD1Dull;
}

// busy-wait for `ticks` of the 100 MHz wall clock; `work` != 0: half of the waves keep the vector ALU busy meanwhile instead of sleeping
__global__ void spin_kernel(long long ticks, int work, float* sink) {
  const long long t0 = wall_clock64();
  float a = (float)threadIdx.x, b = 1.0001f;
  const bool busy = work && ((threadIdx.x >> 6) & 1);
  while (wall_clock64() - t0 < ticks) {
    if (busy) {
#pragma unroll
      for (int i = 0; i < 64; ++i) a = __builtin_fmaf(a, b, 0.5f);
    } else {
      __builtin_amdgcn_s_sleep(8);
    }
  }
  if (sink && a == 12345.678f) *sink = a;   // (keeps the arithmetic alive)
}

inline void spin_on(hipStream_t st, int blocks, int threads, float us, int work) {
  hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(threads), 0, st, (long long)(us * 100.f), work, (float*)nullptr);
}

// fuzz: delay stream `st` by a random time with probability 1/2
inline void fuzz_delay(hipStream_t st) {
  const unsigned long long r = rng_next();
  if (g_mode.fuzz_max_us > 0 && (r & 1)) spin_on(st, 1, 64, (float)((r >> 8) % (unsigned)(g_mode.fuzz_max_us + 1)), 0);
}
inline void fuzz_noise() {
  if (g_mode.noise_blocks <= 0) return;
  if (!g_mode.noise_stream && hipStreamCreateWithFlags(&g_mode.noise_stream, hipStreamNonBlocking) != hipSuccess) return;
  const unsigned long long r = rng_next();
  if (r & 3) spin_on(g_mode.noise_stream, g_mode.noise_blocks, 256, (float)((r >> 8) % (unsigned)(g_mode.fuzz_max_us > 0 ? g_mode.fuzz_max_us + 1 : 21)), 1);
}

}  // namespace

extern "C" void* tpgsr_plan_create(void) { return new Plan(); }

extern "C" void tpgsr_plan_destroy(void* plan) { delete static_cast<Plan*>(plan); }

extern "C" int tpgsr_plan_size(const void* plan) { return plan ? (int)static_cast<const Plan*>(plan)->ops.size() : -1; }

extern "C" int tpgsr_plan_add_launch(void* plan, const char* symbol, const tpgsr_plan_arg* args, int nargs, int side) {
  if (!plan || !symbol || !args) {
    tpgsr_set_error("tpgsr_plan_add_launch: null argument");
    return -1;
  }
  auto it = registry().find(symbol);
  if (it == registry().end()) {
    tpgsr_set_error("tpgsr_plan_add_launch: '%s' is not a launch entry point", symbol);
    return -1;
  }
  const Entry& e = it->second;
  if (nargs != e.nargs || nargs > MAX_ARGS) {
    tpgsr_set_error("tpgsr_plan_add_launch: %s takes %d arguments before the stream, got %d", symbol, e.nargs, nargs);
    return -1;
  }
  Plan* p = static_cast<Plan*>(plan);
  Op o;
  memset(&o, 0, sizeof(o));
  o.kind = OP_LAUNCH;
  o.sid = side < 0 ? 0 : (side > 2 ? 2 : side);    // 0: the caller's stream, 1: side (weight gradients), 2: leaf stream
  o.nargs = nargs;
  o.fn = e.fn;
  o.name = it->first.c_str();
  o.blob = -1;
  memcpy(o.args, args, sizeof(tpgsr_plan_arg) * nargs);
  if (e.struct_bytes > 0) {
    if (!args[0].p) {
      tpgsr_set_error("tpgsr_plan_add_launch: %s needs its argument struct", symbol);
      return -1;
    }
    std::unique_ptr<char[]> b(new char[e.struct_bytes]);
    memcpy(b.get(), args[0].p, e.struct_bytes);
    o.args[0].p = b.get();
    o.blob = (int)p->blobs.size();
    p->blobs.push_back(std::move(b));
  }
  p->ops.push_back(o);
  return (int)p->ops.size() - 1;
}

static int add_edge(void* plan, int kind) {
  if (!plan) {
    tpgsr_set_error("tpgsr_plan_add_fork/join: null plan");
    return -1;
  }
  Plan* p = static_cast<Plan*>(plan);
  Op o;
  memset(&o, 0, sizeof(o));
  o.kind = kind;
  o.blob = -1;
  p->ops.push_back(o);   // the event is created on first replay (recording needs no device)
  return (int)p->ops.size() - 1;
}
extern "C" int tpgsr_plan_add_fork(void* plan) { return add_edge(plan, OP_FORK); }
extern "C" int tpgsr_plan_add_join(void* plan) { return add_edge(plan, OP_JOIN); }
extern "C" int tpgsr_plan_add_edge(void* plan, int src, int dst) {
  if (src < 0 || src > 2 || dst < 0 || dst > 2 || src == dst) {
    tpgsr_set_error("tpgsr_plan_add_edge: stream ids must be two different values of 0, 1, 2 (got %d -> %d)", src, dst);
    return -1;
  }
  const int i = add_edge(plan, OP_EDGE);
  if (i >= 0) {
    Plan* p = static_cast<Plan*>(plan);
    p->ops[i].sid = src;
    p->ops[i].nargs = dst;
  }
  return i;
}

extern "C" int tpgsr_plan_set_arg(void* plan, int op, int arg, const tpgsr_plan_arg* value) {
  Plan* p = static_cast<Plan*>(plan);
  if (!p || op < 0 || op >= (int)p->ops.size() || p->ops[op].kind != OP_LAUNCH || arg < 0 || arg >= p->ops[op].nargs ||
      (arg == 0 && p->ops[op].blob >= 0) || !value) {
    tpgsr_set_error("tpgsr_plan_set_arg: bad slot (op %d, arg %d)", op, arg);
    return -1;
  }
  p->ops[op].args[arg] = *value;
  return 0;
}

extern "C" int tpgsr_plan_run3(void* plan, void* main_stream, void* side_stream, void* leaf_stream) {
  Plan* p = static_cast<Plan*>(plan);
  if (!p) {
    tpgsr_set_error("tpgsr_plan_run: null plan");
    return -1;
  }
  hipStream_t s[3] = {(hipStream_t)main_stream, (hipStream_t)side_stream, (hipStream_t)leaf_stream};
  const bool serial = g_mode.serial != 0;
  if (serial) s[1] = s[2] = s[0];
  const bool fuzz = !serial && (g_mode.fuzz_max_us > 0 || g_mode.noise_blocks > 0);
  const int n = (int)p->ops.size();
  const bool stamp = g_mode.stamp != 0;
  if (stamp && (int)p->stamps.size() != n) {
    p->stamps.resize(n, nullptr);
    p->stamp_stream.resize(n, nullptr);
    for (int i = 0; i < n; ++i)
      if (p->ops[i].kind == OP_LAUNCH && hipEventCreate(&p->stamps[i]) != hipSuccess) {
        tpgsr_set_error("tpgsr_plan_run: hipEventCreate (stamp mode) failed");
        return -2;
      }
  }
  p->stamped = stamp;
  for (int i = 0; i < n; ++i) {
    Op& o = p->ops[i];
    if (o.kind == OP_LAUNCH) {
      if (!serial && o.sid && (!s[o.sid] || s[o.sid] == s[0])) {
        tpgsr_set_error("tpgsr_plan_run: the plan has launches on stream %d but no distinct stream was given for it", o.sid);
        return -1;
      }
      int rc = o.fn(o.args, s[o.sid]);
      if (rc) return rc;   // the entry point has set the message
      if (stamp) {
        p->stamp_stream[i] = (void*)s[o.sid];
        if (hipEventRecord(p->stamps[i], s[o.sid]) != hipSuccess) {
          tpgsr_set_error("tpgsr_plan_run: hipEventRecord (stamp mode) failed");
          return -2;
        }
      }
    } else {
      if (serial) continue;   // one stream: recording order is execution order
      const int src = o.kind == OP_FORK ? 0 : o.kind == OP_JOIN ? 1 : o.sid;
      const int dst = o.kind == OP_FORK ? 1 : o.kind == OP_JOIN ? 0 : o.nargs;
      if ((src && !s[src]) || (dst && !s[dst]) || s[src] == s[dst]) {      // (the caller's stream may be the null stream)
        tpgsr_set_error("tpgsr_plan_run: the plan orders stream %d after stream %d but no distinct streams were given for them", dst, src);
        return -1;
      }
      if (!o.ev && hipEventCreateWithFlags(&o.ev, hipEventDisableTiming) != hipSuccess) {
        tpgsr_set_error("tpgsr_plan_run: hipEventCreateWithFlags failed");
        return -2;
      }
      if (fuzz) {             // a missing edge only shows when the streams drift: push a random one of them back, on either side of the edge
        const int k = (int)(rng_next() % 3);
        fuzz_delay(s[k] ? s[k] : s[0]);
        fuzz_delay(s[src]);
        fuzz_noise();
      }
      if (hipEventRecord(o.ev, s[src]) != hipSuccess || hipStreamWaitEvent(s[dst], o.ev, 0) != hipSuccess) {
        tpgsr_set_error("tpgsr_plan_run: stream fork/join failed: %s", hipGetErrorString(hipGetLastError()));
        return -2;
      }
      if (fuzz) fuzz_delay(s[dst]);
    }
  }
  return 0;
}

extern "C" int tpgsr_plan_run(void* plan, void* main_stream, void* side_stream) {
  return tpgsr_plan_run3(plan, main_stream, side_stream, nullptr);
}

// ---- side stream with a compute-unit mask -----------------------------------------------------------------------------
// The weight-gradient stream may be confined to a subset of the CUs so that its MFMA-heavy workgroups do not sit on the
// SIMDs the latency-bound critical path (BiGRU BPTT, BN reductions, data-gradient convs) runs on.
extern "C" void* tpgsr_stream_create(const unsigned int* cu_mask, int n_words) {
  hipStream_t st = nullptr;
  hipError_t e = (cu_mask && n_words > 0) ? hipExtStreamCreateWithCUMask(&st, (uint32_t)n_words, cu_mask)
                                          : hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  if (e != hipSuccess) {
    tpgsr_set_error("tpgsr_stream_create: %s", hipGetErrorString(e));
    return nullptr;
  }
  return st;
}

extern "C" int tpgsr_stream_destroy(void* stream) {
  if (stream && hipStreamDestroy((hipStream_t)stream) != hipSuccess) {
    tpgsr_set_error("tpgsr_stream_destroy failed");
    return -2;
  }
  return 0;
}

// ---- execution modes / diagnostics ------------------------------------------------------------------------------------
extern "C" void tpgsr_plan_set_mode(int serial, int fuzz_max_us, unsigned long long seed, int noise_blocks) {
  g_mode.serial = serial ? 1 : 0;
  g_mode.fuzz_max_us = fuzz_max_us > 0 ? fuzz_max_us : 0;
  g_mode.noise_blocks = noise_blocks > 0 ? noise_blocks : 0;
  g_mode.rng = seed ? seed * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull : 0x9E3779B97F4A7C15ull;
  if (!g_mode.rng) g_mode.rng = 1;
}

extern "C" int tpgsr_plan_get_mode(void) {
  return (g_mode.serial ? 1 : 0) | (g_mode.fuzz_max_us > 0 ? 2 : 0) | (g_mode.noise_blocks > 0 ? 4 : 0) | (g_mode.stamp ? 8 : 0);
}

/* the fuzz perturbation for stream edges made OUTSIDE a plan (a train step's own wait_stream calls): delays `stream` like an edge of a plan would */
extern "C" int tpgsr_plan_fuzz_point(void* stream) {
  if (g_mode.serial || (g_mode.fuzz_max_us <= 0 && g_mode.noise_blocks <= 0)) return 0;
  fuzz_delay((hipStream_t)stream);
  fuzz_noise();
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int tpgsr_spin(int blocks, int threads, float us, int work, void* stream) {
  if (blocks < 1 || threads < 64 || threads > 1024 || (threads & 63) || us < 0.f || us > 1e6f) {
    tpgsr_set_error("tpgsr_spin: bad arguments (blocks %d, threads %d, us %g)", blocks, threads, (double)us);
    return -1;
  }
  spin_on((hipStream_t)stream, blocks, threads, us, work);
  if (hipGetLastError() != hipSuccess) {
    tpgsr_set_error("tpgsr_spin: launch failed");
    return -2;
  }
  return 0;
}

/* stamp mode: every launch of a plan is followed by a timing event on its stream; tpgsr_plan_stamp_epoch records the common origin */
extern "C" int tpgsr_plan_set_stamp(int on) {
  g_mode.stamp = on ? 1 : 0;
  return 0;
}

extern "C" int tpgsr_plan_stamp_epoch(void* stream) {
  if (!g_mode.epoch && hipEventCreate(&g_mode.epoch) != hipSuccess) {
    tpgsr_set_error("tpgsr_plan_stamp_epoch: hipEventCreate failed");
    return -2;
  }
  if (hipEventRecord(g_mode.epoch, (hipStream_t)stream) != hipSuccess) {
    tpgsr_set_error("tpgsr_plan_stamp_epoch: hipEventRecord failed");
    return -2;
  }
  return 0;
}

/* after a device synchronisation: ms_out[i] = time from the epoch to the end of op i of the plan's last run (-1 for stream edges);
 * stream_out[i] = the HIP stream it ran on (as an integer); returns the number of ops written (<= cap), or a negative error */
extern "C" int tpgsr_plan_read_stamps(void* plan, float* ms_out, long long* stream_out, int cap) {
  Plan* p = static_cast<Plan*>(plan);
  if (!p || !ms_out || !p->stamped || !g_mode.epoch) {
    tpgsr_set_error("tpgsr_plan_read_stamps: the plan's last run was not stamped (tpgsr_plan_set_stamp(1), tpgsr_plan_stamp_epoch first)");
    return -1;
  }
  const int n = (int)p->ops.size() < cap ? (int)p->ops.size() : cap;
  for (int i = 0; i < n; ++i) {
    ms_out[i] = -1.f;
    if (stream_out) stream_out[i] = (long long)(intptr_t)p->stamp_stream[i];
    if (p->ops[i].kind != OP_LAUNCH) continue;
    if (hipEventElapsedTime(&ms_out[i], g_mode.epoch, p->stamps[i]) != hipSuccess) {
      tpgsr_set_error("tpgsr_plan_read_stamps: hipEventElapsedTime failed at op %d (synchronise the device first)", i);
      return -2;
    }
  }
  return n;
}
