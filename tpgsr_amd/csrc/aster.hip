// ASTER evaluation recognizer, greedy decode (reference model/recognizer/*, interfaces/base.py:844-864):
//   * parse_aster_data: bicubic resize of the RGB planes + [0,1] -> [-1,1], NCHW in, NHWC out (interfaces/base.py:852-858)
//   * one decoder step of AttentionRecognitionHead.sample (attention_recognition_head.py:47-67): attention weights + context
//     (AttentionUnit :196-218, DecoderUnit :258-260), embedding lookup + concat (:262-264), GRU cell gate math (nn.GRU, :264),
//     softmax arg-max + score (:60-61)
// The encoder (ResNet_ASTER + 2-layer BiLSTM), the STN head and the TPS rectification reuse the conv / BN / pool / LSTM / TPS kernels.
#include "common.h"

__device__ __forceinline__ void aster_cubic(float t, float (&w)[4]) {     // A = -0.75, as F.interpolate(mode='bicubic')
  const float A = -0.75f;
  float x = t + 1.f;
  w[0] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
  x = t;
  w[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  x = 1.f - t;
  w[2] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  x = 2.f - t;
  w[3] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
}
__device__ __forceinline__ void aster_src(int o, int in_size, int out_size, int& i0, float (&w)[4]) {
  const float scale = (float)in_size / (float)out_size;
  const float x = ((float)o + 0.5f) * scale - 0.5f;
  const float fx = floorf(x);
  i0 = (int)fx;
  aster_cubic(x - fx, w);
}
__device__ __forceinline__ int aster_clamp(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

// out[n][oh][ow][c] = scale * bicubic(in[n][c])(oh, ow) + shift,  c < C (the first C of Ctot planes), align_corners = False
__global__ __launch_bounds__(256) void bicubic_resize_kernel(const float* __restrict__ in, int N, int Ctot, int C, int H, int W, int OH,
                                                             int OW, float scale, float shift, float* __restrict__ out) {
  const long long total = (long long)N * OH * OW * C;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  long long r = i / C;
  const int ow = (int)(r % OW);
  r /= OW;
  const int oh = (int)(r % OH), n = (int)(r / OH);
  int y0, x0;
  float wy[4], wx[4];
  aster_src(oh, H, OH, y0, wy);
  aster_src(ow, W, OW, x0, wx);
  const float* p = in + ((size_t)n * Ctot + c) * H * W;
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int yy = aster_clamp(y0 - 1 + a, H - 1);
    float rowv = 0.f;
#pragma unroll
    for (int b = 0; b < 4; ++b) rowv += p[(size_t)yy * W + aster_clamp(x0 - 1 + b, W - 1)] * wx[b];
    acc += rowv * wy[a];
  }
  out[i] = acc * scale + shift;
}

extern "C" int tpgsr_bicubic_resize(const float* in_nchw, int N, int Ctot, int C, int H, int W, int OH, int OW, float scale, float shift,
                                    float* out_nhwc, void* stream) {
  TPGSR_CHECK_ARG(in_nchw && out_nhwc && N > 0 && C > 0 && Ctot >= C && H > 0 && W > 0 && OH > 0 && OW > 0, "tpgsr_bicubic_resize: bad arguments");
  const long long total = (long long)N * OH * OW * C;
  hipLaunchKernelGGL(bicubic_resize_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, in_nchw, N, Ctot, C, H, W, OH, OW,
                     scale, shift, out_nhwc);
  TPGSR_LAUNCH_CHECK("tpgsr_bicubic_resize");
}

// ------------------------------------------------------------------------------------------------------
// attention of one decoder step, one workgroup per sequence:
//   v[t] = wv . tanh(sproj[n] + xproj[n][t]) + bv;  alpha = softmax_t(v);  context[n] = sum_t alpha[t] x[n][t]
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void aster_attention_kernel(const float* __restrict__ xproj, const float* __restrict__ sproj,
                                                              const float* __restrict__ wv, const float* __restrict__ bv,
                                                              const float* __restrict__ x, int T, int A, int D, float* __restrict__ alpha,
                                                              float* __restrict__ context) {
  extern __shared__ float sm[];      // [T] scores
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xp = xproj + (size_t)n * T * A;
  const float* sp = sproj + (size_t)n * A;
  for (int t = wave; t < T; t += 4) {           // one wave per time step, lanes over the attention dimension
    float acc = 0.f;
    for (int a = lane; a < A; a += 64) acc += wv[a] * tanh_f(sp[a] + xp[(size_t)t * A + a]);
    acc = wave_sum(acc);
    if (lane == 0) sm[t] = acc + bv[0];
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int t = 0; t < T; ++t) mx = fmaxf(mx, sm[t]);
  float den = 0.f;
  for (int t = 0; t < T; ++t) den += expf(sm[t] - mx);
  const float inv = 1.f / den;
  if (tid < T) alpha[(size_t)n * T + tid] = expf(sm[tid] - mx) * inv;
  const float* xn = x + (size_t)n * T * D;
  for (int d = tid; d < D; d += 256) {
    float acc = 0.f;
    for (int t = 0; t < T; ++t) acc += expf(sm[t] - mx) * inv * xn[(size_t)t * D + d];
    context[(size_t)n * D + d] = acc;
  }
}

extern "C" int tpgsr_aster_attention(const float* xproj, const float* sproj, const float* wv, const float* bv, const float* x, int N, int T,
                                     int A, int D, float* alpha, float* context, void* stream) {
  TPGSR_CHECK_ARG(xproj && sproj && wv && bv && x && alpha && context && N > 0 && T > 0 && T <= 256 && A > 0 && D > 0,
                  "tpgsr_aster_attention: bad arguments (T must be <= 256)");
  hipLaunchKernelGGL(aster_attention_kernel, dim3(N), dim3(256), T * sizeof(float), (hipStream_t)stream, xproj, sproj, wv, bv, x, T, A, D,
                     alpha, context);
  TPGSR_LAUNCH_CHECK("tpgsr_aster_attention");
}

// out[n] = [ emb[ids[n]] (E floats) | ctx[n] (D floats) ]
__global__ __launch_bounds__(256) void embed_concat_kernel(const int* __restrict__ ids, const float* __restrict__ emb, int V, int E,
                                                           const float* __restrict__ ctx, int D, int N, float* __restrict__ out) {
  const long long total = (long long)N * (E + D);
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int n = (int)(i / (E + D)), j = (int)(i - (long long)n * (E + D));
  if (j < E) {
    int id = ids[n];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    out[i] = emb[(size_t)id * E + j];
  } else {
    out[i] = ctx[(size_t)n * D + (j - E)];
  }
}

extern "C" int tpgsr_embed_concat(const int* ids, const float* emb, int V, int E, const float* ctx, int D, int N, float* out, void* stream) {
  TPGSR_CHECK_ARG(ids && emb && ctx && out && V > 0 && E > 0 && D > 0 && N > 0, "tpgsr_embed_concat: bad arguments");
  hipLaunchKernelGGL(embed_concat_kernel, dim3(cdiv((long long)N * (E + D), 256)), dim3(256), 0, (hipStream_t)stream, ids, emb, V, E, ctx, D,
                     N, out);
  TPGSR_LAUNCH_CHECK("tpgsr_embed_concat");
}

// nn.GRU cell gate math (gate order r, z, n): h' = (1 - z) n + z h,  r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r gh_n)
__global__ __launch_bounds__(256) void gru_cell_kernel(const float* __restrict__ gi, const float* __restrict__ gh, const float* __restrict__ h,
                                                       int N, int Hd, float* __restrict__ hnew) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)N * Hd) return;
  const int n = (int)(i / Hd), j = (int)(i - (long long)n * Hd);
  const float* a = gi + (size_t)n * 3 * Hd;
  const float* b = gh + (size_t)n * 3 * Hd;
  const float r = sigmoid_f(a[j] + b[j]);
  const float z = sigmoid_f(a[Hd + j] + b[Hd + j]);
  const float nn_ = tanh_f(a[2 * Hd + j] + r * b[2 * Hd + j]);
  hnew[i] = (1.f - z) * nn_ + z * h[i];
}

extern "C" int tpgsr_gru_cell(const float* gi, const float* gh, const float* h, int N, int Hd, float* hnew, void* stream) {
  TPGSR_CHECK_ARG(gi && gh && h && hnew && N > 0 && Hd > 0, "tpgsr_gru_cell: bad arguments");
  hipLaunchKernelGGL(gru_cell_kernel, dim3(cdiv((long long)N * Hd, 256)), dim3(256), 0, (hipStream_t)stream, gi, gh, h, N, Hd, hnew);
  TPGSR_LAUNCH_CHECK("tpgsr_gru_cell");
}

// ids[n] = argmax_c logits[n][c] (first maximum, as torch.max), score[n] = softmax(logits[n])[ids[n]]; one wave per row;
// written at ids_out[n * ld + col] / score_out[n * ld + col] so a decode loop fills its (N, max_len) result in place
__global__ __launch_bounds__(64) void softmax_max_kernel(const float* __restrict__ logits, int C, int* __restrict__ ids, float* __restrict__ score,
                                                         int ld, int col, int* __restrict__ ids_next) {
  const int n = blockIdx.x, lane = threadIdx.x;
  const float* p = logits + (size_t)n * C;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane; c < C; c += 64) {
    const float v = p[c];
    if (v > best) {
      best = v;
      bi = c;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o);
    const int oi = __shfl_xor(bi, o);
    if (ov > best || (ov == best && oi < bi)) {
      best = ov;
      bi = oi;
    }
  }
  float den = 0.f;
  for (int c = lane; c < C; c += 64) den += expf(p[c] - best);
  den = wave_sum(den);
  if (lane == 0) {
    ids[(size_t)n * ld + col] = bi;
    score[(size_t)n * ld + col] = 1.f / den;
    if (ids_next) ids_next[n] = bi;
  }
}

extern "C" int tpgsr_softmax_max(const float* logits, int N, int C, int* ids, float* score, int ld, int col, int* ids_next, void* stream) {
  TPGSR_CHECK_ARG(logits && ids && score && N > 0 && C > 0 && ld > col && col >= 0, "tpgsr_softmax_max: bad arguments");
  hipLaunchKernelGGL(softmax_max_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, logits, C, ids, score, ld, col, ids_next);
  TPGSR_LAUNCH_CHECK("tpgsr_softmax_max");
}
