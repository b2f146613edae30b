/*
 * tpgsr_hip.h -- C ABI of libtpgsr_hip.so: the MI355X (gfx950 / CDNA4) kernels behind the TPGSR-TSRN
 * training / inference hot path.
 *
 * The reference (mjq11302010044/TPGSR) has NO native layer: its operator API is Python nn.Modules
 *  whose forward()s bottom out in stock ATen/cuDNN kernels.  Every entry point
 * below replaces one of those implicit kernel calls; the reference call site each one stands in for is
 * cited as file:line (relative to the reference tree).  The Python host side (tpgsr_amd/) mirrors the
 * reference's module names / state_dict layout and binds these symbols with ctypes.
 *
 * Conventions
 *   - all tensors are fp32, device pointers, owned by the caller (kernels never allocate);
 *   - activations are NHWC ("pixel-major"): element (n,h,w,c) of a logical [N][H][W][C] tensor lives at
 *     ((n*H+h)*W+w)*ld + coff + c   (ld = floats per pixel, coff = channel offset; default ld=C, coff=0);
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), no internal sync, safe to
 *     capture into a hipGraph;
 *   - return 0 on success, a negative code on a rejected argument / failed launch; the message is
 *     available from tpgsr_last_error().
 */
#ifndef TPGSR_HIP_H
#define TPGSR_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define TPGSR_ACT_NONE 0
#define TPGSR_ACT_RELU 1
#define TPGSR_ACT_MISH 2   /* x * tanh(softplus(x)), model/tsrn.py:480-488 */
#define TPGSR_ACT_TANH 3
#define TPGSR_ACT_PRELU 4

const char* tpgsr_last_error(void);
int tpgsr_version(void);
/* sizeof of the argument structs (0 conv_args, 1 wgrad_args, 2 pack_desc, 3 wgrad_reduce_desc, 4 compose_bwd_desc, 5 split_desc, 6 image_desc, 7 gru_wgrad_args, 8 wgrad_batch_item, 9 bn_derive, 10 bigru_proj_args): a binding can verify its mirror */
int tpgsr_sizeof(int which);

/* ------------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), stride 1.
 * Replaces every nn.Conv2d / nn.Linear / 1x1 conv on the path:
 *   model/tsrn.py:28 (9x9 4->64), :375,:379 (3x3 64->64), :467 (3x3 64->256 + PixelShuffle),
 *   :495 (GruBlock 1x1), :40/:159 (tail 9x9, run as a 9x1 conv with the 9 kw taps folded into N),
 *   the GRU/LSTM input projections (nn.GRU tsrn.py:496, nn.LSTM crnn/crnn.py:10),
 *   model/stn_head.py:15,48-52 and model/crnn/crnn.py:45-46,12.
 * out[m][n] = epilogue( sum_k A[m][k] * wt[k][n] ),  m = output pixel, k = (tap, ci), n = co.
 * A-operand prologue, fused into the tile loader (never materialised in HBM):
 *     a = act_in( in * in_scale[c] + in_shift[c] ) + in2        (zero padding applied afterwards)
 * which is how train-mode BatchNorm-apply + mish/ReLU and the residual adds of
 * model/tsrn.py:388-394,211 ride on the consumer conv.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const float* in;        /* logical [N][H][W][Cin] */
  const float* in2;       /* optional, same geometry as `in` (ld = in2_ld, coff = 0) */
  const float* in_scale;  /* optional [Cin] */
  const float* in_shift;  /* optional [Cin] */
  const float* wt;        /* packed [K = KH*KW*Cin][Cout], k = (kh*KW+kw)*Cin + ci */
  const float* bias;      /* optional [Cout] */
  float* out;             /* logical [N][OH][OW][Cout] */
  float* bn_partial;      /* optional [ceil(M/64)][2][Cout]: per-row-block sum / sum of squares of (out-bias) */
  int N, H, W, Cin;
  int in_ld, in_coff, in2_ld;
  int in_act;             /* TPGSR_ACT_NONE / RELU / MISH applied in the loader */
  int in_ps;              /* 1: `in` is stored pixel-shuffled: [N][2H][2W][Cin/4]  (un-PixelShuffle gather) */
  int Cout, KH, KW, pad_h, pad_w, OH, OW;
  int out_ld, out_coff;
  int out_act;            /* NONE / RELU / TANH */
  int out_ps;             /* 1: store pixel-shuffled (nn.PixelShuffle(2), model/tsrn.py:469): [N][2OH][2OW][Cout/4] */
  /* --- text-prior path (TSRN_TL, model/tsrn.py:81-108, :411-426) --- */
  const float* in_b;      /* optional second channel source: channels c >= cin_a come from this [N][W][Cin - cin_a]
                             strip, broadcast over H (torch.cat([residual, text_emb], 1) without materialising it) */
  int cin_a, in_b_ld;
  int in_dil_w;           /* >1: `in` is zero-dilated along W by this factor (logical W = (Wreal-1)*dil+1): the
                             stride-s ConvTranspose2d of InfoGen run as a stride-1 conv over the dilated strip */
  int wt_ld, wt_coff;     /* row stride / column offset of `wt` (0 = Cout / 0): lets a dgrad run on a column block */
  int stride_w;           /* output stride along W (0/1 = dense): iw = ow*stride_w + kw - pad_w (dgrad of a ConvTranspose2d) */
  /* --- bf16 matrix-core path with split fp32 operands (csrc/conv_xbf.hip) --- */
  int terms;              /* 0: fp32 MFMA (v_mfma_f32_32x32x2_f32).  3: fp32-equivalent on the bf16 matrix cores: every fp32
                             operand is the exact sum of three bf16 terms, six v_mfma_f32_32x32x16_bf16 per product block.
                             2: two-term split (a1 b1 + a1 b2 + a2 b1: three MFMAs, ~16 significand bits per operand, what is
                             dropped is <= 3 * 2^-18 |a b| per product).  1: plain bf16 operands, fp32 accumulate.  Needs the vector loader (Cin % 4 == 0) and, for
                             tpgsr_conv_fwd, wt_bf; otherwise the call silently stays on the fp32 kernel. */
  int kp;                 /* K rounded up to a multiple of 32 = row length of wt_bf */
  const void* wt_bf;      /* `wt` pre-split by tpgsr_split_bf_program: bf16 planes in MFMA fragment order
                             [3][ceil(rows / 32)][kp / 16][64 lanes][8], rows = wt_ld (or Cout), zero padded */
  int wt_bf_cin;          /* 0: wt_bf rows in the natural k = (tap, ci) order.  Cin (a multiple of 32): in channel-block order
                             k' = ((ci / 32) * KH*KW + tap) * 32 + ci % 32  (tpgsr_split_desc.cin), which is what lets KH x KW
                             convolutions run on the halo kernel */
  int reserved0;
  /* --- BatchNorm-backward statistics in the epilogue of the convolution that PRODUCES the incoming gradient (split-bf16 kernels only;
   *     the fp32 kernel refuses it).  With bnb_y set, `out` = da is stored as usual and bn_partial receives, per 64-pixel row block,
   *     [0][c] = sum dz, [1][c] = sum dz * (y - mean[c]) * rstd[c],  dz = da * act'(y * bnb_scale[c] + bnb_shift[c])
   *     -- the first pass of tpgsr_bn_bwd_reduce (autograd's batch_norm_backward reduction, model/tsrn.py:376,380) without its launch and
   *     its re-read of da.  y is dense [M][Cout]. --- */
  const float* bnb_y;     /* optional: the BatchNorm's input (the pre-normalisation map of the forward pass) */
  const float* bnb_mean;  /* [Cout] batch mean / reciprocal standard deviation saved by tpgsr_bn_finalize */
  const float* bnb_rstd;
  const float* bnb_scale; /* [Cout] folded scale / shift (needed when bnb_act != NONE) */
  const float* bnb_shift;
  int bnb_act;            /* activation that followed the BatchNorm in the forward pass: NONE / RELU / MISH */
  int bnb_store_dz;       /* 1: `out` receives dz instead of da.  With bn_partial NULL this is a plain activation backward on the
                             way out -- out = da * act'(y * scale + shift), scale / shift optional (1 / 0) -- which is how the mish
                             in front of the tail convolution (model/tsrn.py:39,159) is differentiated without a launch of its own */
  /* --- BatchNorm FINALIZE by the launch itself (fin_mode != 0; needs bn_partial).  What follows a statistics-producing convolution in
   *     the reference -- nn.BatchNorm2d's batch mean / variance (forward) or batch_norm_backward's coefficients (backward) -- is a
   *     reduction of bn_partial's ceil(M / 64) rows: a 3 us kernel behind a 5 us launch boundary, 36 times per C3 step.  With fin_mode
   *     set tpgsr_conv_fwd guarantees it has run when the call's work is done: kernels that can (the whole-CU halo kernel) let their
   *     LAST workgroup do it -- partial rows by write-through stores, one relaxed agent-scope ticket per workgroup, the last one reads
   *     the rows back with L1-bypassing loads and sums them in a FIXED order, so the result does not depend on which workgroup it was --
   *     for every other kernel the launcher appends tpgsr_bn_finalize / tpgsr_bn_bwd_finalize on the same stream.
   *     mode 1 (forward):  scale / shift / save_mean / save_rstd (+ running statistics) as tpgsr_bn_finalize, count = M
   *     mode 2 (backward): dgamma += sum dz xhat, dbeta += sum dz, coef [3][Cout] as tpgsr_bn_bwd_finalize (mean / rstd = bnb_mean / bnb_rstd) --- */
  int fin_mode;
  int fin_accumulate;        /* mode 2: 1 = add to dgamma / dbeta */
  long long fin_count;       /* elements per channel the statistics were taken over */
  int* fin_counter;          /* one zero-initialised int of device memory per BatchNorm layer; left at zero */
  const float* fin_gamma;    /* [Cout] */
  const float* fin_beta;     /* mode 1 */
  const float* fin_bias;     /* mode 1: the convolution's bias when it is NOT part of `out` already (mean shift), or NULL */
  float* fin_scale;          /* mode 1: [Cout] folded scale;    mode 2: coef [3][Cout] */
  float* fin_shift;          /* mode 1: [Cout] folded shift;    mode 2: dgamma [Cout] or NULL */
  float* fin_mean;           /* mode 1: save_mean [Cout] or NULL; mode 2: dbeta [Cout] or NULL */
  float* fin_rstd;           /* mode 1: save_rstd [Cout] or NULL */
  float* fin_rm;             /* mode 1: running_mean / running_var [Cout] or NULL */
  float* fin_rv;
  float fin_momentum;
  float fin_eps;
  /* --- granularity of bn_partial (round 5).  0 / 1: one row per 64-pixel block (every kernel).  3: one row per THREE consecutive blocks
   *     = per 192-pixel super-tile of the whole-CU halo kernel, [ceil(ceil(M / 64) / 3)][2][Cout] -- a third of the rows for whoever
   *     reduces them (csrc/bn_derive.h: every workgroup of the consuming launch).  Only the whole-CU kernel honours 3: ask
   *     tpgsr_conv_bn_row_tiles() first, a launch that lands on any other kernel with 3 set is refused.  Not with fin_mode. --- */
  int bn_row_tiles;
  int reserved1;
  /* --- scaled residual operand (round 6): a = in * in_scale[c] + in_shift[c] + in2 * in2_scale[c].  What it is for: the apply pass of
   *     batch_norm_backward, dy = c0 dz + c1 y + c2 (model/tsrn.py:376-380's BatchNorms in the backward pass), folded into the loader of
   *     the data-gradient convolution that consumes dy (in = dz, in2 = y, in_scale = c0, in_shift = c2, in2_scale = c1): the caller's
   *     stream no longer runs tpgsr_bn_bwd_apply in front of it.  Whole-CU halo kernel only: ask tpgsr_conv_in2_scale_ok() first. --- */
  const float* in2_scale;    /* optional [Cin], 16-byte aligned; needs in2 and in_scale / in_shift, no in_act / in_b / in_ps */
  /* --- split-K for the tile loop (round 6).  A few launches of the step have fewer 64 x 64 output tiles than the chip has CUs and a long
   *     contraction -- the BiLSTM projections' data gradients (2048 -> 512 / 256 over 1248 pixels), InfoGen's transposed convolutions
   *     (512 -> 128, 1 x 3 over a zero-dilated strip), the STN head's 3 x 3 convolutions on 96-pixel maps: 8 .. 160 workgroups walking
   *     48 .. 72 K chunks, every chunk a full round trip.  With sk_splits = S > 1 the launch runs S workgroups per tile, each over
   *     1 / S of the chunks, their raw accumulators go to sk_part [S][tiles][256 threads][16] (fp32, fragment order), and a second
   *     launch of `tiles` workgroups adds them in split order and runs the ordinary epilogue (bias, activation, pixel-shuffle store,
   *     BatchNorm statistics).  Deterministic; the sum is associated differently from the unsplit launch (last-bit differences).
   *     Ask tpgsr_conv_splitk_plan() for S and the scratch size; sk_part must not be shared by launches that may run concurrently. --- */
  float* sk_part;
  int sk_splits;             /* 0 / 1: off */
  int reserved2;
} tpgsr_conv_args;
/* split-K plan of tpgsr_conv_fwd(a) under the current switches: returns S (0 = do not split: the launch is another kernel's, or has
 * enough tiles / too short a contraction) and the bytes sk_part needs.  On by default; TPGSR_XBF_SPLITK=0 / tpgsr_splitk_set_enabled(0): always 0 */
int tpgsr_conv_splitk_plan(const tpgsr_conv_args* a, long long* bytes);
void tpgsr_splitk_set_enabled(int on);
/* 1 when tpgsr_conv_fwd(a) with a->in2_scale set will be taken (by the whole-CU halo kernel), else 0 */
int tpgsr_conv_in2_scale_ok(const tpgsr_conv_args* a);
/* 3 when tpgsr_conv_fwd(a) will run on the whole-CU halo kernel (which can leave one bn_partial row per 192 pixels), else 1 */
int tpgsr_conv_bn_row_tiles(const tpgsr_conv_args* a);

int tpgsr_conv_fwd(const tpgsr_conv_args* a, void* stream);

/* Weight-gradient GEMM:  part[z][k][n] = sum_{m in split z} A[m][k] * dy[m][n]   (A through the same
 * loader/prologue as tpgsr_conv_fwd), plus optional bias-gradient partials dbpart[z][n] = sum_m dy[m][n].
 * Replaces the cuDNN wgrad kernels autograd launches for every conv/linear above.
 * tpgsr_wgrad_splits() returns the number of splits Z the kernel will use for (M, K, Cout). */
typedef struct {
  tpgsr_conv_args c;      /* geometry + A-operand description (wt/bias/out/bn_partial/out_* ignored) */
  const float* dy;        /* logical [N][OH][OW][Cout] */
  int dy_ld, dy_coff;
  int dy_ps;              /* 1: dy stored pixel-shuffled [N][2OH][2OW][Cout/4] */
  float* part;            /* [Z][K][Cout] */
  float* dbpart;          /* optional [Z][Cout] */
  int zsplits;            /* 0: Z = tpgsr_wgrad_splits(M, K, Cout).  > 0: the caller sized part / dbpart for exactly this many pixel
                             splits (every kernel honours it); tpgsr_wgrad_halo_plan proposes the count the halo kernel wants */
  int reserved1;
  void* dy_bf;            /* optional scratch of tpgsr_wgrad_halo_plan's dy_bf_bytes: dy pre-split into bf16 fragment planes
                             [3][ceil(M/16)][ceil(Cout/32)][64 lanes][8].  With zsplits > 0, terms > 0 and a KH x KW > 1 x 1
                             stride-1 convolution over a multiple of 32 channels, the weight gradient runs on the halo kernel
                             (csrc/conv_xbf.hip); anything else stays on the tile loop. */
} tpgsr_wgrad_args;

int tpgsr_wgrad_splits(int M, int K, int Cout);
/* 1 + (*zsplits, *dy_bf_bytes) when the geometry (and a->terms > 0) suits the halo weight-gradient kernel, else 0.
 * Only N, H, W, Cin, Cout, KH, KW, pads, OH, OW, terms and the loader-shape fields of `a` are read. */
int tpgsr_wgrad_halo_plan(const tpgsr_conv_args* a, int* zsplits, long long* dy_bf_bytes);
/* (tpgsr_wgh_debug, the lab switch that turned parts of the halo weight-gradient kernel off, exists only in -DTPGSR_LAB builds of
 * csrc/conv_xbf.hip: tools/lab/wgh_probe.py) */
int tpgsr_conv_wgrad(const tpgsr_wgrad_args* a, void* stream);
/* Several independent weight-gradient GEMMs in ONE launch (the ten of the text-prior generator's two BiLSTM layers, model/crnn/crnn.py:5-26:
 * 25-65 us each alone -- start-up, not work).  items_dev: device-resident table; an item = the arguments of one tpgsr_conv_wgrad call plus
 * what its launcher derives from them, filled on the host by tpgsr_conv_wgrad_batch_prepare (returns the loader variant `ld`, >= 0, when
 * the call is a 1x1 geometry on the split-bf16 tile-loop kernel -- the only kind a batch takes; all items of a batch share `ld` and
 * `terms`); blk0 = prefix sum of nblk over the preceding items, total_blocks = the full sum.  Results are those of the single launches,
 * bit for bit (same workgroups, same slabs). */
typedef struct tpgsr_wgrad_batch_item {
  tpgsr_wgrad_args w;
  int M, K, MB, blk0, nblk, reserved0, reserved1, reserved2;
} tpgsr_wgrad_batch_item;
int tpgsr_conv_wgrad_batch_prepare(const tpgsr_wgrad_args* w, tpgsr_wgrad_batch_item* item);
int tpgsr_conv_wgrad_batch(const tpgsr_wgrad_batch_item* items_dev, int n, int total_blocks, int ld, int terms, void* stream);

/* Deterministic second stage: dw (+)= sum_z part[z], written in the PyTorch parameter layout
 *   layout 0: conv / linear weight [Cout][Cin][KH][KW]
 *   layout 1: ConvTranspose2d weight [Cin][Cout][KH][KW] run as its equivalent conv (flipped taps)
 *   layout 2: the folded tail conv (KH = KS, KW = 1, Cout = KS*Co) back to [Co][Cin][KS][KS]
 *   layout 3: InfoGen's ConvTranspose2d on an H=1 strip (1x3 conv, flipped taps) back to [Cin][Cout][3][3], row kh=1
 * and db (+)= sum_z dbpart[z].  accumulate != 0 adds to the existing gradient (autograd semantics). */
int tpgsr_wgrad_reduce(const float* part, const float* dbpart, int Z, int K, int Cin, int Cout, int KH, int KW,
                       int layout, float* dw, float* db, int accumulate, float gscale /* multiplies dw only */,
                       void* stream);

/* The same reduce for MANY layers in one launch: a device-resident table of descriptors (fields as the parameters of
 * tpgsr_wgrad_reduce; blk0 = first workgroup of the descriptor, consecutive descriptors own consecutive ranges of
 * tpgsr_wgrad_reduce_blocks2(...) workgroups; total_blocks = their sum).  Descriptors of one program must
 * target distinct dw / db. */
typedef struct tpgsr_wgrad_reduce_desc {
  const float* part;
  const float* dbpart;
  float* dw;
  float* db;
  int Z, K, Cin, Cout, KH, KW, layout, accumulate;
  float gscale;
  int blk0;
  int cin_ld;             /* slab row k = tap*cin_ld + ci (0 = Cin); rows with ci >= Cin or tap >= KH*KW (K may exceed
                             KH*KW*cin_ld) belong to a zero-padded operand and are skipped */
  int reserved;
} tpgsr_wgrad_reduce_desc;
/* workgroups ONE descriptor takes: layout-0 weights are reduced in 2-D tiles (all taps of 32 / taps input channels x 32 output
 * channels) and written transposed, so the count depends on the parameter's shape -- a program's blk0 offsets are built from this */
int tpgsr_wgrad_reduce_blocks2(int K, int Cin, int Cout, int KH, int KW, int layout, int cin_ld, int has_bias);
int tpgsr_wgrad_reduce_program(const tpgsr_wgrad_reduce_desc* descs_dev, int ndesc, int total_blocks, void* stream);

/* Pack a PyTorch conv weight [Cout][Cin][KH][KW] into the forward operand wt_f[(kh*KW+kw)*Cin+ci][co] and the
 * data-gradient operand wt_d[((KH-1-kh)*KW+(KW-1-kw))*Cout+co][ci] (dgrad == tpgsr_conv_fwd over dy with wt_d
 * and pad' = K-1-pad).  Either output may be NULL.  transposed != 0: source is a ConvTranspose2d weight
 * [Cin][Cout][KH][KW].  wscale multiplies the packed copies (STN fc2 sees 0.1*feat, model/stn_head.py:100). */
int tpgsr_pack_conv_weight(const float* w, int Cout, int Cin, int KH, int KW, int transposed, float wscale,
                           float* wt_f, float* wt_d, void* stream);
/* Tail conv 9x9 C->Co (Co = 3 or 4) folded to a 9x1 conv with N' = 9*Co columns (n' = kw*Co+co):
 * wt_f[(kh*C+ci)][kw*Co+co] = w[co][ci][kh][kw];  wt_d = its dgrad packing over the 9x1 conv. */
int tpgsr_pack_tail_weight(const float* w, int Co, int C, int KS, float* wt_f, float* wt_d, void* stream);

/* All operand packing of a model in ONE launch: a device-resident table of descriptors.
 *   kind 0: conv/linear weight [Cout][Cin][KH][KW] -> dst_f[k*f_ld + f_coff + co] and/or dst_d (dgrad operand;
 *           rows of d_ld floats when d_ld > 0, columns >= Cin untouched: allocate zero-filled)
 *   kind 1: tail conv [Co][C][KS][KS] folded (Cout = Co, KH = KW = KS)     kind 2: plain copy of numel floats
 *   kind 3: ConvTranspose2d weight [Cin][Cout][KH][KW] as its equivalent conv
 *   kind 4: ConvTranspose2d weight [Cin][Cout][3][3] on an H=1 strip -> 1x3 conv operand (kh=1 slice, taps flipped)
 *   kind 5: GruBlock (model/tsrn.py:491-508) 1x1 conv composed with one direction of the GRU input projection:
 *           Wc[g][ci] = sum_u src[g][u] * src2[u][ci]  (src = weight_ih [Cout=96][KH=64 hidden-of-conv], src2 = conv1
 *           weight [64][Cin]) -> dst_f[ci*f_ld + f_coff + g] and dst_d[(f_coff + g)*Cin + ci]; numel = Cout*Cin
 *   kind 6: its bias: dst_f[f_coff + g] = sum_u src[g][u] * src2[u] + src3[g]  (src2 = conv1 bias, src3 = bias_ih)
 *   kind 7: DATA-GRADIENT operand of a KS x KS convolution [Cout][Cin][KS][KS] with few input channels, folded like the tail:
 *           dst_f[(kh' Cout + c)][kw' Cin + ci] = src[c][ci][KS-1-kh'][KS-1-kw'] (a KS x 1 convolution over dy + tpgsr_shiftsum_nhwc)
 * blk0 = prefix sum of tpgsr_pack_blocks(...) over the preceding descriptors (ceil(numel / 256), except kind 0 with <= 9 taps and Cout * Cin >= 65536,
 * which is packed in 32 x 32-channel tiles through LDS: ceil(Cout / 32) * ceil(Cin / 32) workgroups); total_blocks = the full sum. */
typedef struct {
  const float* src;
  float* dst_f;
  float* dst_d;
  int Cout, Cin, KH, KW;
  int kind, f_ld, f_coff;
  float wscale;
  int numel, blk0;
  const float* src2;
  const float* src3;
  int d_ld;               /* kinds 0/3/4: row stride of dst_d (0 = Cin): a zero-padded dgrad operand */
  int cin_ld;             /* kinds 0/3/4: channel stride of dst_f's k index, k = tap*cin_ld + ci (0 = Cin): zero-padded input channels */
} tpgsr_pack_desc;
/* Chain rule of the composed GruBlock operand, for many blocks in one launch.  Given dWc [2*G][Cin] / dbc [2*G] (the
 * weight / bias gradient of the composed 1x1 conv, both directions stacked, G = 96):
 *   dW1[u][ci] += sum_g Wih[g][u] dWc[g][ci]     db1[u] += sum_g Wih[g][u] dbc[g]
 *   dWih[g][u] += sum_ci dWc[g][ci] W1[u][ci] + dbc[g] b1[u]      dbih[g] += dbc[g]
 * (the conv output is W1 x + b1; Wih = [wih0; wih1], U = 64 conv outputs)
 * blk0 / total_blocks as above with tpgsr_compose_bwd_blocks(Cin) workgroups per descriptor. */
typedef struct tpgsr_compose_bwd_desc {
  const float* dWc;
  const float* dbc;
  const float* W1;
  const float* b1;
  const float* wih0;
  const float* wih1;
  float* dW1;
  float* db1;
  float* dwih0;
  float* dwih1;
  float* dbih0;
  float* dbih1;
  int Cin, U, G, blk0;
} tpgsr_compose_bwd_desc;
int tpgsr_compose_bwd_blocks(int Cin, int U, int G);
int tpgsr_compose_bwd_program(const tpgsr_compose_bwd_desc* descs_dev, int ndesc, int total_blocks, void* stream);
int tpgsr_pack_blocks(int kind, int Cout, int Cin, int KH, int KW, long long numel);
int tpgsr_pack_program(const tpgsr_pack_desc* descs_dev, int ndesc, int total_blocks, void* stream);
/* Split every packed fp32 MFMA operand of a network into bf16 planes for the bf16 matrix-core path, in ONE launch right
 * after tpgsr_pack_program: src fp32 [K][ld] (k-major, N <= ld columns used) -> dst bf16 planes
 * [3][ceil(N/32)][kp/16][64][8] (element (k, n) of term t at lane ((k>>3)&1)*32 + (n&31), slot k&7 of block (n/32, k/16)),
 * kp = 32*ceil(K/32), x = term0 + term1 + term2 exactly, zero padded.  blk0 = prefix sum of tpgsr_split_bf_blocks(K, N). */
typedef struct tpgsr_split_desc {
  const float* src;
  void* dst;
  int K, N, ld, kp, blk0;
  int cin;                /* > 0 (a multiple of 32 dividing K): write the rows in channel-block order, see tpgsr_conv_args.wt_bf_cin */
} tpgsr_split_desc;
int tpgsr_split_bf_blocks(int K, int N);
int tpgsr_split_bf_program(const tpgsr_split_desc* descs_dev, int ndesc, int total_blocks, void* stream);
/* ------------------------------------------------------------------------------------------------
 * ASTER evaluation recognizer, greedy decode (model/recognizer/*, interfaces/base.py:844-864): csrc/aster.hip
 * ---------------------------------------------------------------------------------------------- */
/* parse_aster_data: out[n][oh][ow][c] = scale * bicubic(in[n][c])(oh, ow) + shift for the first C of Ctot NCHW planes
 * (F.interpolate(mode='bicubic'), align_corners False; interfaces/base.py:852-858 with scale 2, shift -1) */
int tpgsr_bicubic_resize(const float* in_nchw, int N, int Ctot, int C, int H, int W, int OH, int OW, float scale, float shift,
                         float* out_nhwc, void* stream);
/* AttentionUnit + context (attention_recognition_head.py:196-218, :258-260): xproj [N][T][A], sproj [N][A], wv [A], bv [1], x [N][T][D]
 * -> alpha [N][T], context [N][D] */
int tpgsr_aster_attention(const float* xproj, const float* sproj, const float* wv, const float* bv, const float* x, int N, int T, int A,
                          int D, float* alpha, float* context, void* stream);
/* out[n] = [ emb[ids[n]] | ctx[n] ]  (tgt_embedding + torch.cat, :262-264) */
int tpgsr_embed_concat(const int* ids, const float* emb, int V, int E, const float* ctx, int D, int N, float* out, void* stream);
/* nn.GRU cell gate math from the two projections gi = W_ih x + b_ih, gh = W_hh h + b_hh ([N][3 Hd], gate order r, z, n) */
int tpgsr_gru_cell(const float* gi, const float* gh, const float* h, int N, int Hd, float* hnew, void* stream);
/* greedy decision (:60-61): ids[n*ld+col] = first arg-max of logits[n], score[n*ld+col] = its softmax probability; ids_next (optional) [N] */
int tpgsr_softmax_max(const float* logits, int N, int C, int* ids, float* score, int ld, int col, int* ids_next, void* stream);

/* the whole-CU halo kernel (csrc/conv_halo3.hip: one workgroup per CU on three 64-pixel tiles at once, T <= 2) takes the KH x KW > 1 x 1
 * convolutions whose 192-pixel halo fits LDS and that fill the chip; 0 sends them back to the two-workgroup halo kernel (tests, A/B) */
void tpgsr_halo3_set_enabled(int on);
/* the tile-loop weight gradient with THREE 64-row k-blocks per workgroup (csrc/conv_xbf.hip: conv_wgrad_xbf3_kernel, round 6) takes the
 * launches of tpgsr_conv_wgrad whose K is a multiple of 192 (plain / affine loader, dense dy, terms 1 or 2); 0 sends them back to the
 * one-block kernel (tests, A/B).  Same slabs, same summation order. */
void tpgsr_wgrad3_set_enabled(int on);
/* tuning knob of the halo forward kernel: weight-plane bytes above which tiles are walked column-major per XCD (keeps an XCD's slice of
 * the weights in its L2); -1 never, 0 whenever the column-tile count allows, default 3 MB.  Results do not depend on it. */
void tpgsr_halo_set_colmajor_min_bytes(long long v);
/* smallest tap count (KH * KW) the halo forward kernel takes: 2 (default) or 1 (1x1 convolutions with Cin % 32 == 0 as well;
 * TPGSR_XBF_HALO_MINTAPS=1 at load time).  Results do not depend on it beyond fp32 summation order. */
void tpgsr_halo_set_min_taps(int v);
void tpgsr_halo_set_ne9(int on);   /* halos of 225..288 entries on the halo forward kernel as well (default off: slower than the tile loop; TPGSR_XBF_HALO_NE9) */
/* The row-panel kernel of the 1x1 convolutions with K <= 192 over many pixels (csrc/conv_panel.hip: the GruBlock projections,
 * model/tsrn.py:491-508, and their data gradients): on / off (TPGSR_XBF_PANEL), and the smallest pixel count it takes (default 32768;
 * TPGSR_XBF_PANEL_MIN_M).  Placement of a launch on this kernel or on the tile loop only affects speed. */
void tpgsr_panel_set_enabled(int on);
void tpgsr_panel_set_min_m(long long m);
void tpgsr_panel_set_k192(int on);   /* K = 192 -> <= 64 columns on the panel kernel as well (default off: the tile loop is faster in x3; TPGSR_XBF_PANEL_K192) */
/* host-only: the halo kernels' LDS entry capacity for this geometry = an upper bound of the halo length of any tile of 64
 * consecutive output pixels (reads OH, OW, KH, KW) */
int tpgsr_halo_capacity(const tpgsr_conv_args* a);
/* diagnostic: time line of the halo convolution kernel.  buf = 8 * 8 * 256 uint64 of device memory ([workgroup][wave][slot],
 * 100 MHz wall-clock stamps, see csrc/conv_xbf.hip) or NULL to switch it off.  Not thread safe; off by default. */
int tpgsr_halo_trace(unsigned long long* buf);
/* diagnostic: out[lane*4 + j] = what ds_read_b64_tr_b16 hands lane `lane` as element j from a [16 rows][16] image of
 * consecutive integers, addressed like the weight-gradient fragment fetch (64 lanes, 256 ints) */
int tpgsr_tr_probe(int* out, void* stream);
/* diagnostic: d = (a b + ...) applied `reps` times with ONE v_mfma_f32_32x32x16_bf16: a [32][16] / b [16][32] bf16 bit patterns, c / d [32][32] f32 */
int tpgsr_mfma_bf16_probe(const void* a, const void* b, const float* c, float* d, int reps, void* stream);
/* diagnostic: `blocks` workgroups x 4 waves x 2*iters register-only v_mfma_f32_32x32x2_f32 (8192 FLOP each per wave) */
int tpgsr_mfma_probe(float* out, int blocks, int iters, void* stream);
int tpgsr_copy(const float* src, float* dst, long long n, void* stream);   /* async D2D copy (graph memcpy node) */
int tpgsr_zero(float* dst, long long n, void* stream);                     /* async memset */

/* ------------------------------------------------------------------------------------------------
 * Train-mode BatchNorm pieces (nn.BatchNorm2d model/tsrn.py:376,380,153; stn_head.py:19; crnn.py:48).
 * ---------------------------------------------------------------------------------------------- */
/* From the conv epilogue partials: batch mean / biased var -> scale = gamma*rstd, shift = beta - mean*scale,
 * save_mean / save_rstd, and the running-stat update (momentum 0.1, unbiased var). bias (optional) is the conv
 * bias the partials were shifted by.  eval != 0: scale/shift from the running stats, nothing updated. */
int tpgsr_bn_finalize(const float* partial, int nblk, int C, long long count, const float* conv_bias,
                      const float* gamma, const float* beta, float* running_mean, float* running_var,
                      float momentum, float eps, int eval, float* scale, float* shift, float* save_mean,
                      float* save_rstd, void* stream);
/* Per-channel partial stats of an arbitrary [M][C] tensor (used where no conv epilogue produced them). */
int tpgsr_bn_stats(const float* x, long long M, int C, int ld, float* partial, int nblk, void* stream);
/* BN(+activation) backward, pass 1: dz = da * act'(scale*y+shift); partial sums of dz and dz*xhat. */
int tpgsr_bn_bwd_reduce(const float* da, const float* da2, const float* y, long long M, int C, const float* scale,
                        const float* shift, const float* save_mean, const float* save_rstd, int act,
                        float* partial, int nblk, void* stream);
/* finalize: dgamma, dbeta (accumulate flag) and the per-channel coefficients of pass 2 */
int tpgsr_bn_bwd_finalize(const float* partial, int nblk, int C, long long count, const float* gamma,
                          const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta,
                          int accumulate, float* coef /* [3][C] */, void* stream);
/* pass 2: dy = coef0*dz + coef1*y + coef2 (dz recomputed from da(+da2), y) */
int tpgsr_bn_bwd_apply(const float* da, const float* da2, const float* y, long long M, int C, const float* scale,
                       const float* shift, int act, const float* coef, float* dy, void* stream);
/* ------------------------------------------------------------------------------------------------
 * BatchNorm finalize inside the launch that consumes it (round 5; csrc/bn_derive.h).  The reduction of the producing convolution's
 * partial rows that nn.BatchNorm2d's batch statistics (model/tsrn.py:376,380; model/stn_head.py:15; forward) and batch_norm_backward's
 * coefficients (backward) need is done by the FIRST CONSUMER's launch instead of a launch of its own: the first ceil(C / 16) workgroups
 * of its grid each sum the rows of 16 channels (fp64, fixed order), publish by write-through stores and arrive on `flag`; every workgroup
 * waits for the flag (its first loads already in flight) and reads the published values with L1-bypassing loads.  A wait that never ends
 * poisons the outputs with NaN.  tpgsr_bn_finalize / tpgsr_bn_bwd_finalize stay for consumers without the prologue.
 * C: 8, or a multiple of 16 up to 512; the grid must have at least ceil(C / 16) workgroups (M C >= 1024 ceil(C / 16)).
 *   forward  (tpgsr_affine_act_bnd, tpgsr_affine_act_pool_bnd): reads rows / count / bias / gamma / beta, writes scale / shift
 *            (+ save_mean / save_rstd / running statistics when set) exactly as tpgsr_bn_finalize does
 *   backward (tpgsr_bn_bwd_apply_bnd): reads rows ([.][0][c] = sum dz, [.][1][c] = sum dz * xhat) / count / gamma / save_mean /
 *            save_rstd, writes coef and dgamma / dbeta (+= when accumulate) exactly as tpgsr_bn_bwd_finalize does
 * ---------------------------------------------------------------------------------------------- */
typedef struct tpgsr_bn_derive {
  const float* rows;      /* [nrows][2][C] */
  int nrows, C;
  long long count;        /* elements per channel the rows were summed over */
  const float* bias;      /* forward: the convolution's bias when it is not part of the stored map (shifts the mean only) or NULL */
  const float* gamma;     /* [C] */
  const float* beta;      /* forward */
  float* running_mean;    /* forward, optional: updated with `momentum` (unbiased variance) */
  float* running_var;
  float momentum, eps;
  float* scale;           /* forward out: folded scale / shift [C] */
  float* shift;
  float* save_mean;       /* forward: out (optional); backward: in */
  float* save_rstd;
  float* dgamma;          /* backward, optional */
  float* dbeta;
  float* coef;            /* backward, optional out [3][C] */
  int accumulate;         /* backward: 1 = add to dgamma / dbeta */
  int reserved;
  unsigned* flag;         /* one word of device memory, ZERO when the launch starts (tpgsr_zero earlier on the stream): the arrival counter
                             of the launch's deriver workgroups; the launch leaves ceil(C / 16) in it */
} tpgsr_bn_derive;
/* tpgsr_affine_act with the BatchNorm finalized in the launch: out = act(scale[c] * x + shift[c]) */
int tpgsr_affine_act_bnd(const tpgsr_bn_derive* d, const float* x, long long M, int act, float* out, void* stream);
/* tpgsr_affine_act_pool likewise (STN head: conv -> BN -> ReLU -> max-pool, model/stn_head.py:34-45) */
int tpgsr_affine_act_pool_bnd(const tpgsr_bn_derive* d, const float* x, int N, int H, int W, int act, int pool_h, int pool_w,
                              float* out, void* stream);
/* tpgsr_bn_bwd_finalize + tpgsr_bn_bwd_apply in one launch: dy = coef0 * dz + coef1 * y + coef2, dz = (da + da2) * act'(scale * y + shift) */
int tpgsr_bn_bwd_apply_bnd(const tpgsr_bn_derive* d, const float* da, const float* da2, const float* y, long long M,
                           const float* scale, const float* shift, int act, float* dy, void* stream);

/* out = act(scale[c]*x + shift[c]) over an [M][C] tensor (scale/shift optional, C % 4 == 0): materialises an activation
 * once where re-applying it per filter tap in the consumer's loader would cost more (mish before a 3x3 / 9x1 conv). */
int tpgsr_affine_act(const float* x, long long M, int C, const float* scale, const float* shift, int act, float* out,
                     void* stream);
/* Materialise act(scale*x+shift) (+ optional 2x2 / 1x2 max-pool, STN head model/stn_head.py:34-45) */
int tpgsr_affine_act_pool(const float* x, int N, int H, int W, int C, const float* scale, const float* shift,
                          int act, int pool_h, int pool_w, float* out, void* stream);
/* backward of the above: routes dout to the arg-max position and multiplies by act' and scale -> d(pre-affine)=dz;
 * writes dz (the BN-backward passes then run with act = NONE on dz). */
int tpgsr_affine_act_pool_bwd(const float* x, const float* dout, int N, int H, int W, int C, const float* scale,
                              const float* shift, int act, int pool_h, int pool_w, float* dz, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Elementwise
 * ---------------------------------------------------------------------------------------------- */
/* PReLU with one shared slope (model/tsrn.py:29-31) */
int tpgsr_prelu_fwd(const float* x, const float* alpha, long long n, float* y, void* stream);
int tpgsr_prelu_bwd(const float* x, const float* alpha, const float* dy, const float* dy2, long long n, float* dx,
                    float* dalpha_partial, int nblk, void* stream);
/* out = a + b (b optional scale) ; out = a*mish'(x) etc. */
int tpgsr_add(const float* a, const float* b, long long n, float* out, void* stream);
int tpgsr_act_bwd(const float* x, const float* dy, long long n, int act, float* dx, void* stream);
int tpgsr_nchw_to_nhwc(const float* in, int N, int C, int H, int W, float* out, void* stream);
int tpgsr_nhwc_to_nchw(const float* in, int N, int C, int H, int W, float* out, void* stream);
/* sum_z part[z][n] -> out[n] (+=) ; used for small parameter gradients (PReLU slope, GRU b_hh ...) */
int tpgsr_reduce_partials(const float* part, int Z, int n, float* out, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Bidirectional GRU (hidden 32) over one spatial axis of an NHWC map -- GruBlock, model/tsrn.py:491-508.
 * gi [P][192] = input projections (+b_ih) for both directions, column = dir*96 + gate*32 + j, gate order
 * (r,z,n); h_out [P][64], column = dir*32 + j.  axis 0: sequences run along W (one per (n,row));
 * axis 1: along H (one per (n,col)) -- the `.transpose(-1,-2)` of model/tsrn.py:391 without a copy.
 * ---------------------------------------------------------------------------------------------- */
int tpgsr_bigru_fwd(const float* gi, const float* w_hh /* [2][96][32] */, const float* b_hh /* [2][96] */,
                    int N, int H, int W, int axis, float* h_out, float* gates /* [P][256] or NULL */, void* stream);
/* GruBlock FORWARD in one launch (round 5, csrc/gru_proj.hip; model/tsrn.py:491-508 with the 1x1 convolution and nn.GRU's input
 * projection composed, :419-426 for the text-prior variant): gi = loader(x) Wc^T + bc is computed by the wave that owns the sequence, on
 * the matrix cores, into LDS, and the scan of tpgsr_bigru_fwd runs from there -- the [P][192] projection never touches HBM.
 *   c       the projection as tpgsr_conv_args: in / in_ld, the loader (in_scale + in_shift, in2, in_b + cin_a), Cin 64 | 96, Cout 192,
 *           bias = bc [192], terms 1..3 with wt_bf / kp = the operand Wc [Cin][192] split by tpgsr_split_bf_program; wt / out are not read
 *   w_hh [2][96][32], b_hh [2][96], h_out [P][64], gates [P][256] or NULL (inference), axis as tpgsr_bigru_fwd
 * tpgsr_bigru_proj_supported() says whether a GruBlock is this kernel's (scan length 16 or 64, the loaders above); results equal
 * tpgsr_conv_fwd + tpgsr_bigru_fwd up to the accumulation order of the projection. */
typedef struct tpgsr_bigru_proj_args {
  tpgsr_conv_args c;
  const float* w_hh;
  const float* b_hh;
  float* h_out;
  float* gates;
  int axis;
  int reserved;
} tpgsr_bigru_proj_args;
int tpgsr_bigru_proj_supported(const tpgsr_bigru_proj_args* p);
int tpgsr_bigru_proj_fwd(const tpgsr_bigru_proj_args* p, void* stream);
void tpgsr_bigru_proj_set_enabled(int on);
/* Backward through time from the gate values the forward pass stored: gates [P][256], column = dir*128 + q*32 + j,
 * q = (r, z, n, W_hn h + b_hn) -- pass gates = NULL to tpgsr_bigru_fwd at inference.  dh = dh_out (+ dh_out2 if
 * non-NULL).  Writes dgi [P][192] = (dr, dz, dn) pre-activation gradients of the input side and
 * dgh [P][192] = (dr, dz, dn*r) of the hidden side, from which dW_ih/db_ih/d(input) and dW_hh/db_hh follow
 * as plain GEMMs / column sums (tpgsr_conv_wgrad against the input resp. the one-step-shifted states). */
int tpgsr_bigru_bwd(const float* gates, const float* h_out, const float* dh_out, const float* dh_out2,
                    const float* w_hh, int N, int H, int W, int axis, float* dgi, float* dgh, void* stream);
/* tpgsr_bigru_bwd with the hidden-side gradient written compactly: dghn [P][64] = dn_pre * r of both directions (column = dir*32 + j) --
 * the r and z planes of dgh equal dgi's, and tpgsr_gru_wgrad reads them there. */
int tpgsr_bigru_bwd2(const float* gates, const float* h_out, const float* dh_out, const float* dh_out2,
                     const float* w_hh, int N, int H, int W, int axis, float* dgi, float* dghn, void* stream);
/* test hook: the recurrence's own sigmoid / tanh (csrc/gru.hip: compensated v_exp_f32, v_rcp_f32 + one Newton step) over n values */
int tpgsr_gru_gate_math_probe(const float* x, float* sg, float* th, int n, void* stream);
/* ALL weight gradients of one GruBlock in one launch (csrc/gru_wgrad.hip; model/tsrn.py:491-508, the backward pass of GruBlock.forward):
 *   c      the A side of the composed 1x1 projection gi = loader(x) Wc^T + bc, as tpgsr_conv_args (in / in_ld / in_coff, optional
 *          in_scale + in_shift, in2 (residual add) or in_b (concatenated [N][W][Cb] strip, cin_a); N, H, W, Cin = 64 | 96, Cout = 192,
 *          1x1, OH = H, OW = W; terms = 1 | 2 | 3: split-bf16 matrix-core path only)
 *   dgi [P][192], dghn [P][64] from tpgsr_bigru_bwd2; h [P][64] the BiGRU's output (h_prev = h one step against the scan direction)
 *   zsplits = tpgsr_gru_wgrad_splits(P) = Z; slabs for tpgsr_wgrad_reduce(_program):
 *   partC [Z][Cin][192], dbC [Z][192]   -> dWc, dbc (composed operand; chain rule: tpgsr_compose_bwd_program)
 *   partH [2][Z][32][96], dbH [2][Z][96] -> weight_hh_l0(_reverse) [96][32] (layout 0), bias_hh_l0(_reverse) */
typedef struct tpgsr_gru_wgrad_args {
  tpgsr_conv_args c;
  const float* dgi;
  const float* dghn;
  const float* h;
  float* partC;
  float* dbC;
  float* partH;
  float* dbH;
  int axis;
  int zsplits;
} tpgsr_gru_wgrad_args;
int tpgsr_gru_wgrad_splits(long long P);
int tpgsr_gru_wgrad(const tpgsr_gru_wgrad_args* w, void* stream);
/* look-ahead, in time steps, of the operand prefetch rings of tpgsr_bigru_fwd / _bwd: 4, 8 (default) or 12 (TPGSR_GRU_PF); speed only */
void tpgsr_gru_set_prefetch(int steps);

/* ------------------------------------------------------------------------------------------------
 * STN / TPS rectification -- model/tps_spatial_transformer.py:97-112, grid_sample :10-18
 * ---------------------------------------------------------------------------------------------- */
/* ctrl [N][NC][2] -> src [N][HW][2] (pre-clamp source coordinates, optional output) and the sampling grid
 * [N][HW][2] = 2*clamp(src,0,1)-1.  inv_kernel [NC+3][NC+3], coord_repr [HW][NC+3] are the registered
 * buffers `tps.inverse_kernel` / `tps.target_coordinate_repr`. */
int tpgsr_tps_grid_fwd(const float* ctrl, const float* inv_kernel, const float* coord_repr, int N, int HW,
                       int NC, float* grid, float* src, void* stream);
/* dctrl [N][NC][2] from dgrid; src (saved by the forward) decides the clamp mask. */
int tpgsr_tps_grid_bwd(const float* dgrid, const float* src, const float* inv_kernel, const float* coord_repr,
                       int N, int HW, int NC, float* dctrl, void* stream);
/* bilinear, zeros padding; NHWC in/out with C channels (C = 3 or 4). */
int tpgsr_grid_sample_fwd(const float* in, const float* grid, int N, int H, int W, int C, int OH, int OW,
                          int align_corners, float* out, void* stream);
int tpgsr_grid_sample_bwd(const float* in, const float* grid, const float* dout, int N, int H, int W, int C,
                          int OH, int OW, int align_corners, float* din /* optional; zero-filled by callee */,
                          float* dgrid /* optional */, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Text-prior strip (TSRN_TL): F.interpolate(strip, (H, W), bilinear, align_corners=True) of a [N][1][Win][C] strip
 * (model/tsrn.py:196) is a 1-D resample along W (rows are copies): out [N][Wout][C] = lerp of relu(scale*in+shift).
 * ---------------------------------------------------------------------------------------------- */
int tpgsr_strip_resample_fwd(const float* in, const float* scale, const float* shift, int act, int N, int Win, int Wout,
                             int C, float* out, void* stream);
/* d(pre-activation in) from dout [N][Wout][C] (dz = d relu(scale*in+shift) wrt its argument; BN backward follows) */
int tpgsr_strip_resample_bwd(const float* in, const float* scale, const float* shift, int act, const float* dout, int N,
                             int Win, int Wout, int C, float* dz, void* stream);
/* dstrip[n][w][c] (+)= sum_h d[n][h][w][c]  (gradient of the H-broadcast) */
int tpgsr_hsum(const float* d, int N, int H, int W, int C, float* dstrip, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Text-prior generator (CRNN) pieces -- model/crnn/crnn.py:29-90, interfaces/base.py:806-829,
 * loss/semantic_loss.py:21-39, interfaces/super_resolution.py:316-321,372-382
 * ---------------------------------------------------------------------------------------------- */
/* parse_crnn_data: F.interpolate(x[:, :3], (OH, OW), 'bicubic') (A=-0.75, align_corners False) + 0.299R+0.587G+0.114B.
 * in: NCHW [N][Ctot>=3][H][W]  ->  out [N][OH][OW] (= NHWC with C = 1).  bwd: adjoint into din (NCHW, zero-filled here). */
int tpgsr_bicubic_gray_fwd(const float* in_nchw, int N, int Ctot, int H, int W, int OH, int OW, float* out, void* stream);
int tpgsr_bicubic_gray_bwd(const float* dout, int N, int Ctot, int H, int W, int OH, int OW, float* din_nchw, void* stream);
/* CRNN conv0 (nn.Conv2d(1, 64, 3, 1, 1), model/crnn/crnn.py:45-46) as a 1x1 conv over a 12-channel neighbourhood map:
 * col[n][y][x][kh*3+kw] = in[n][y+kh-1][x+kw-1] (zero padded; channels 9..11 = 0); col2im is its gather-form adjoint. */
int tpgsr_im2col3x3_c1(const float* in, int N, int H, int W, float* col /* [N*H*W][12] */, void* stream);
int tpgsr_col2im3x3_c1(const float* dcol, int N, int H, int W, float* din, void* stream);
/* dst[m][c] = c < Cs ? src[m][c] : 0: pads the 37-class tensors (logits gradient, text prior) to 40 channels so their
 * consumers (Linear(512,37) crnn.py:12, InfoGen tconv1 tsrn.py:89) run on the 16-byte loaders */
int tpgsr_pad_channels(const float* src, long long M, int Cs, int Cd, float* dst, void* stream);
/* nn.MaxPool2d((KH,KW),(SH,SW),(PH,PW)) of act(scale*x+shift), NHWC; bwd returns d(pre-activation) (first arg-max wins) */
int tpgsr_pool2d_fwd(const float* x, int N, int H, int W, int C, const float* scale, const float* shift, int act, int KH,
                     int KW, int SH, int SW, int PH, int PW, float* out, void* stream);
int tpgsr_pool2d_bwd(const float* x, const float* dout, int N, int H, int W, int C, const float* scale, const float* shift,
                     int act, int KH, int KW, int SH, int SW, int PH, int PW, float* dz, void* stream);
/* Recurrent projection of one BiLSTM time step (nn.LSTM, model/crnn/crnn.py:10), both directions in one launch, split
 * over K into S slabs: slab[sp][d][n][c] = sum_{k in slice sp} A_d[n][k] * B_d[k][c], n < Nrows <= 64 (the batch),
 * A_d row n at a_d + n*a_stride, B_d row-major [K][Nc] (K %% 32 == 0, Nc %% 64 == 0).  out: [S][2][Nrows][Nc]. */
int tpgsr_lstm_rec_gemm(const float* a0, const float* a1, long long a_stride, const float* b0, const float* b1, int Nrows,
                        int K, int Nc, int S, float* out, void* stream);
/* One BiLSTM time step for both directions (gate order i,f,g,o).  G [N][T][2][4Hh]: input projections in, activated
 * gates out (fwd) / gate gradients out (bwd); gh [nsplit][2][N][4Hh] = W_hh h_prev of this step as K-split slabs
 * (tpgsr_lstm_rec_gemm; summed in slab order; unused at step 0); Cst [N][T][2][Hh]; out [N][T][2Hh].
 * bwd: dhc [nsplit][2][N][Hh] = W_hh^T dG of the previous backward step, dcc [N][2][Hh]. */
int tpgsr_lstm_step_fwd(float* G, const float* gh, int nsplit, const float* bhh /* [2][4Hh], optional */, float* Cst, float* out,
                        int N, int T, int Hh, int step, void* stream);
/* BiLSTM time step as ONE launch: recurrent projection (bf16 matrix cores, split operands, full K per workgroup) + gate math, Hh == 256,
 * N <= 64; replaces tpgsr_lstm_rec_gemm + tpgsr_lstm_step_fwd.  wfr = tpgsr_lstm_wfrag(whhT) once per pass (tpgsr_lstm_wfrag_bytes()
 * bytes); hx as for tpgsr_lstm_seq_fwd (zeroed once; step s reads parity (s-1)&1 and writes s&1). */
long long tpgsr_lstm_wfrag_bytes(void);
int tpgsr_lstm_wfrag(const float* whhT /* [2][Hh][4Hh] */, void* wfr, int Hh, void* stream);
int tpgsr_lstm_stepx_fwd(float* G, const void* wfr, const float* bhh, float* Cst, float* out, void* hx, int N, int T, int Hh, int step,
                         void* stream);
/* BiLSTM forward as ONE persistent launch (replaces the T x (tpgsr_lstm_rec_gemm + tpgsr_lstm_step_fwd) loop; model/crnn/crnn.py:10,
 * nn.LSTM(bidirectional=True)): Hh == 256, N <= 64.  G / Cst / out as for tpgsr_lstm_step_fwd, whhT [2][Hh][4Hh] = W_hh^T of both
 * directions, bhh [2][4Hh] or NULL.  hx: exchange buffer of tpgsr_lstm_seq_hx_bytes() bytes, ZEROED ONCE by the caller (reusable across
 * calls on one stream); sync: 4 x u32 scratch, zeroed by the call; sync[2] != 0 afterwards = a step's hand-off timed out (the 64
 * workgroups were not co-resident within seconds) and the result is invalid.  csrc/lstm_seq.hip: per-step exchange by write-through
 * stores + one relaxed arrival counter, no fences. */
int tpgsr_lstm_seq_fwd(float* G, const float* whhT, const float* bhh, float* Cst, float* out, void* hx, unsigned* sync, int N, int T,
                       int Hh, void* stream);
long long tpgsr_lstm_seq_hx_bytes(void);
/* Can the persistent BiLSTM launches (tpgsr_lstm_seq_*) run here?  They need their 2 x 32 workgroups resident TOGETHER; this launches
 * 64 workgroups with the larger kernel's footprint that wait for each other for at most ~50 ms.  1: yes; 0: no (record the per-step
 * launches tpgsr_lstm_rec_gemm + tpgsr_lstm_step_* instead: a timed-out hand-off would poison the step with NaN); < 0: error.
 * Synchronises `stream`.  words: 2 u32 of device memory. */
int tpgsr_lstm_seq_probe(unsigned* words, void* stream);
/* The same forward recurrence with a DATA-TAGGED hand-off (8-byte {three bf16 terms of h, tag} granules, no counter, no wait for the
 * stores; csrc/lstm_seq.hip).  hg: tpgsr_lstm_seq_hg_bytes() bytes and sync: 8 x u32, both ZEROED ONCE by the caller and then owned by
 * these launches (the launch epoch lives in sync[4..5]); T <= 31.  sync[2] != 0: a hand-off timed out. */
int tpgsr_lstm_seq_fwdg(float* G, const float* whhT, const float* bhh, float* Cst, float* out, void* hg, unsigned* sync, int N, int T,
                        int Hh, void* stream);
long long tpgsr_lstm_seq_hg_bytes(void);
/* BiLSTM backward recurrence (BPTT of nn.LSTM, model/crnn/crnn.py:10) as ONE persistent launch (replaces the T x (tpgsr_lstm_rec_gemm +
 * tpgsr_lstm_step_bwd) loop): G activated gates in, gate gradients out; dout [N][T][2Hh] = dL/dh; w0 / w1 = weight_hh_l0 / _reverse
 * [4Hh][Hh] as stored; px: exchange buffer of tpgsr_lstm_seq_px_bytes() bytes (no initialisation); sync as above. */
int tpgsr_lstm_seq_bwd(float* G, const float* Cst, const float* dout, const float* w0, const float* w1, void* px, unsigned* sync, int N,
                       int T, int Hh, void* stream);
long long tpgsr_lstm_seq_px_bytes(void);
/* The backward recurrence with the data-tagged hand-off (8-byte {fp32 partial sum, tag} granules): pg of tpgsr_lstm_seq_pg_bytes() bytes and
 * sync (8 x u32) ZEROED ONCE by the caller, then owned by these launches; T <= 255. */
int tpgsr_lstm_seq_bwdg(float* G, const float* Cst, const float* dout, const float* w0, const float* w1, void* pg, unsigned* sync, int N,
                        int T, int Hh, void* stream);
long long tpgsr_lstm_seq_pg_bytes(void);
int tpgsr_lstm_step_bwd(float* G, const float* Cst, const float* dout, const float* dhc, int nsplit, float* dcc, int N, int T,
                        int Hh, int step, void* stream);
/* p = softmax(logits [N][T][C]); prior (N,C,1,T) = p with samples [0, drop_n) zeroed (prior dropout); with q: partial
 * sums of SemanticLoss = mean|q-p| + KLDivLoss('mean')(log(p+1e-20), q+1e-20).  bwd: dlogits from dprior (+dp_in) and
 * the semantic loss weighted by wsem. */
int tpgsr_softmax_prior_fwd(const float* logits, const float* q, int N, int T, int C, int drop_n, float* p, float* prior_nchw,
                            float* partial, int nblk, void* stream);
int tpgsr_semantic_loss_finalize(const float* partial, int nblk, long long count, float w, float* loss, void* stream);
int tpgsr_softmax_prior_bwd(const float* p, const float* q, const float* dprior_nchw, const float* dp_in, int N, int T, int C,
                            int drop_n, float wsem, float* dlogits, int nblk, void* stream);
/* CTC loss of the text-prior generator's logits (`--use_label`: ctc_loss = torch.nn.CTCLoss(blank=0, reduction='none') on
 * label_vecs_logits.log_softmax(2), weighted by weighted_tics and averaged; interfaces/super_resolution.py:40, :347-366).
 * nll[n]; dlogits (optional; element (n, t, c) at n * sn + t * st + c like logits) (+)= scale * weight[n] * d nll[n] / d logits.
 * targets: concatenated int32 labels, sample n at [tgt_off[n], tgt_off[n] + tgt_len[n]); T <= 32, C <= 64, max_len (the longest target) <= 31. */
int tpgsr_ctc_loss(const float* logits, int sn, int st, const int* targets, const int* tgt_off, const int* tgt_len, const float* weight,
                   int N, int T, int C, int blank, float scale, float* nll, float* dlogits, int accumulate, int max_len, void* stream);
/* SemanticLoss.forward(pred, gt) on probability tensors (loss/semantic_loss.py:21-39, the nn.Module entry point):
 * partial [nblk][2] = (sum|q-p|, sum q'(log q' - log p')) -> tpgsr_semantic_loss_finalize; bwd: dp = dloss*(-sign(q-p) - q'/p')/n */
int tpgsr_semantic_loss_fwd(const float* p, const float* q, long long n, float* partial, int nblk, void* stream);
int tpgsr_semantic_loss_bwd(const float* p, const float* q, const float* dloss, long long n, float* dp, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Layout / resampling glue of the standalone operator API and the _TL baseline backbones (csrc/glue.hip), NHWC:
 * torch.cat / channel slicing (model/srresnet.py:213, model/srcnn.py:98-104, model/rdn.py:147,197, model/vdsr.py:31),
 * F.interpolate nearest (model/srcnn.py:91, model/vdsr.py:205) and bilinear align_corners=True (model/srresnet.py:153),
 * zero-dilation / sub-sampling (ConvTranspose2d model/srresnet.py:174-183 and the strided conv4_1 of
 * model/crnn/modules/feature_extraction.py:232 expressed through the stride-1 MFMA conv), mean over H
 * (nn.AdaptiveAvgPool2d((None, 1)), model/crnn/model.py:46,71).  Backward kernels are gathers (deterministic).
 * ---------------------------------------------------------------------------------------------- */
int tpgsr_copy_strided(const float* src, int src_ld, int src_coff, float* dst, int dst_ld, int dst_coff, long long M, int C,
                       int accumulate, void* stream);
int tpgsr_resize_nearest_fwd(const float* in, int N, int H, int W, int C, int s, float* out, void* stream);
int tpgsr_resize_nearest_bwd(const float* dout, int N, int H, int W, int C, int s, float* din, void* stream);
int tpgsr_resize_bilinear_fwd(const float* in, int N, int H, int W, int C, int OH, int OW, float* out, void* stream);
int tpgsr_resize_bilinear_bwd(const float* dout, int N, int H, int W, int C, int OH, int OW, float* din, void* stream);
int tpgsr_dilate2d(const float* in, int N, int H, int W, int C, int sh, int sw, float* out, void* stream);
int tpgsr_subsample2d(const float* in, int N, int H, int W, int C, int sh, int sw, float* out, void* stream);
int tpgsr_hreduce(const float* in, int N, int H, int W, int C, float scale, float* out, void* stream);
int tpgsr_hbroadcast(const float* dout, int N, int H, int W, int C, float scale, float* din, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Tail: out = tanh(bias + sum_kw P[h][w+kw-4][kw][co])  (model/tsrn.py:159,213), NCHW output
 * ---------------------------------------------------------------------------------------------- */
int tpgsr_tail_shiftsum_tanh(const float* P, const float* bias, int N, int H, int W, int Co, int KS,
                             float* out_nchw, void* stream);
/* out [N][H][W][Co] = sum_kw P[n][h][w+kw-KS/2][kw*Co+co]: the same column-group sum without bias / tanh into an NHWC map -- the
 * second half of block1's data gradient (model/tsrn.py:28, 64 -> 4 over 9 x 9 taps) run as a 9 x 1 convolution with 36 columns */
int tpgsr_shiftsum_nhwc(const float* P, int N, int H, int W, int Co, int KS, float* out, void* stream);
/* dP[h][x][kw][co] = dpre[h][x-kw+4][co], dpre = dout*(1-out^2); also bias-grad partials */
int tpgsr_tail_bwd(const float* out_nchw, const float* dout_nchw, int N, int H, int W, int Co, int KS,
                   float* dP, float* dbias_partial /* optional [nblk][Co] */, int nblk, void* stream);
int tpgsr_tail_bwd_blocks(int N, int H, int W, int Co, int KS); /* the nblk tpgsr_tail_bwd expects */

/* ------------------------------------------------------------------------------------------------
 * Losses -- loss/image_loss.py:10-51, loss/semantic_loss.py:10-39; NCHW tensors like the reference
 * ---------------------------------------------------------------------------------------------- */
/* loss = w0*MSE(all C) + w1*L1(gradmag(out[:, :3]), gradmag(tgt[:, :3])); writes partial sums [nblk][2] */
int tpgsr_image_loss_fwd(const float* out, const float* tgt, int N, int C, int H, int W, int gradient,
                         float* partial, int nblk, void* stream);
int tpgsr_image_loss_finalize(const float* partial, int nblk, long long n_mse, long long n_gp, float w0, float w1,
                              float* loss, void* stream);
int tpgsr_image_loss_bwd(const float* out, const float* tgt, const float* dloss, int N, int C, int H, int W,
                         int gradient, float w0, float w1, float* dout, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Input pipeline (dataset/dataset.py:615-632 resizeNormalize, called by alignCollate_real* :1226-1323): Pillow's 8-bit
 * bicubic `img.resize` + ToTensor + luminance-threshold mask for a batch of variable-size uint8 HWC images, bit-exact.
 * tpgsr_resample_coeffs (HOST) fills Pillow's per-output-index bounds [out][2] and 22-bit fixed-point taps
 * [out][tpgsr_resample_ksize]; the batch's tables are concatenated into one int32 device buffer, every image's descriptor
 * holds the offsets (in ints).  pixels: the images' bytes back to back; tmp [N][maxH][OW][3], res8 [N][OH][OW][3] scratch;
 * out [N][3 + mask][OH][OW] float.
 * ---------------------------------------------------------------------------------------------- */
typedef struct tpgsr_image_desc {
  long long offset;       /* byte offset of the image's [H][W][3] pixels */
  int H, W;
  int xb_off, xk_off, kx; /* horizontal pass: bounds table, tap table, taps per output column */
  int yb_off, yk_off, ky; /* vertical pass */
} tpgsr_image_desc;
int tpgsr_resample_ksize(int in_size, int out_size);
int tpgsr_resample_coeffs(int in_size, int out_size, int* bounds, int* kk);
int tpgsr_resize_normalize(const unsigned char* pixels, const tpgsr_image_desc* descs_dev, const int* tables_dev, int N, int OH, int OW,
                           int maxH, int mask, unsigned char* tmp, unsigned char* res8, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Evaluation path (interfaces/super_resolution.py:540-900): CTC greedy decoding of the recogniser's logits
 * (utils/metrics.py:71-88 get_string_crnn; logits batch-major [N][T][C], labels [N][T] collapsed + blank-free, -1 padded),
 * PSNR and SSIM of NCHW images over their first 3 channels (utils/ssim_psnr.py:9-15, :18-78; window = the reference's
 * 11x11 Gaussian, passed in).  partial: nblk doubles of scratch.
 * ---------------------------------------------------------------------------------------------- */
int tpgsr_ctc_greedy_decode(const float* logits, int N, int T, int C, int* labels, int* lengths, void* stream);
int tpgsr_psnr(const float* a, const float* b, int N, int Ctot, int H, int W, double* partial, int nblk, float* out, void* stream);
int tpgsr_ssim(const float* a, const float* b, const float* window, int KS, int N, int Ctot, int H, int W, double* partial, int nblk,
               float* out, void* stream);
/* SSIM as a loss (`--ssim_loss`, interfaces/super_resolution.py:388-391: loss_ssim = (1 - ssim(sr, hr).mean()) * 10): the gradient of the SSIM
 * map's sum with respect to the first image, da[:, :min(Ctot,3)] (+)= mult * coef[0] * d(sum ssim_map)/da (NCHW; channels >= 3 untouched).
 * gm: scratch of 3 * N * min(Ctot,3) * H * W floats; coef: optional device scalar (upstream gradient).  utils/ssim_psnr.py:30-50. */
int tpgsr_ssim_bwd(const float* a, const float* b, const float* window, int KS, int N, int Ctot, int H, int W, float* gm,
                   const float* coef, float mult, float* da, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Optimiser -- clip_grad_norm_(0.25) + Adam(lr 1e-3, betas (0.5,0.999)) over flat arenas
 * (interfaces/super_resolution.py:419-424, interfaces/base.py:449-450)
 * ---------------------------------------------------------------------------------------------- */
int tpgsr_sumsq_partial(const float* x, long long n, float* partial, int nblk, void* stream);
/* coef[0] = min(1, max_norm / (sqrt(sum partial) + 1e-6)); norm_out[0] = sqrt(sum) */
int tpgsr_clip_coef(const float* partial, int nblk, float max_norm, float* coef, float* norm_out, void* stream);
/* g' = g * (*gscale if gscale else 1);  Adam update with bias correction from the device step counter */
int tpgsr_adam_step(float* p, const float* g, float* m, float* v, long long n, const float* gscale, float lr,
                    float beta1, float beta2, float eps, const int* step_dev, void* stream);
int tpgsr_step_inc(int* step_dev, void* stream);
/* tpgsr_clip_coef (partial == NULL: skipped) and tpgsr_step_inc of up to eight counters in ONE single-wave launch -- the optimiser's launches
 * sit on the step's exposed tail.  `steps`: a HOST array of nsteps distinct device pointers (copied into the launch's arguments). */
int tpgsr_clip_coef_steps(const float* partial, int nblk, float max_norm, float* coef, float* norm_out, int* const* steps, int nsteps, void* stream);
int tpgsr_scale_(float* x, long long n, const float* coef, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Launch plans: a recorded sequence of the launches above, replayed by one call (the host-side analogue of
 * the reference's per-iteration Python loop body, interfaces/super_resolution.py:336-424, with the interpreter
 * taken out of the per-kernel path).  `symbol` is the name of any launch entry point of this header (the ones
 * whose last parameter is the stream); `args` are its parameters in order, WITHOUT the stream: pointers in .p,
 * integers in .i, floats in .f.  For tpgsr_conv_fwd / tpgsr_conv_wgrad args[0].p points to the argument struct,
 * which is copied.  side != 0 puts the launch on the side stream given to tpgsr_plan_run; fork orders the side
 * stream after everything recorded so far on the main stream, join orders the main stream after the side stream.
 * add_* return the op index (>= 0) or a negative error.
 * ---------------------------------------------------------------------------------------------- */
typedef union tpgsr_plan_arg {
  const void* p;
  long long i;
  double f;
} tpgsr_plan_arg;
void* tpgsr_plan_create(void);
void tpgsr_plan_destroy(void* plan);
int tpgsr_plan_size(const void* plan);
int tpgsr_plan_add_launch(void* plan, const char* symbol, const tpgsr_plan_arg* args, int nargs, int side);
int tpgsr_plan_add_fork(void* plan);
int tpgsr_plan_add_join(void* plan);
int tpgsr_plan_set_arg(void* plan, int op, int arg, const tpgsr_plan_arg* value);   /* patch a per-step pointer / scalar */
int tpgsr_plan_run(void* plan, void* main_stream, void* side_stream);
/* A third stream ("leaf"): launches recorded with side == 2 run on it; tpgsr_plan_add_edge(plan, src, dst) orders stream dst (0 main,
 * 1 side, 2 leaf) after everything recorded so far on stream src.  The SR network's STN-head backward -- a long chain of small
 * launches nothing but the optimiser waits for (model/stn_head.py, model/tsrn.py:183) -- runs there, next to the text-prior
 * generator's backward pass on the main stream. */
int tpgsr_plan_add_edge(void* plan, int src, int dst);
int tpgsr_plan_run3(void* plan, void* main_stream, void* side_stream, void* leaf_stream);
/* Execution modes of tpgsr_plan_run3, process-wide -- test and diagnostic instruments, no product entry point switches them on:
 *   serial != 0      : every launch goes to the caller's stream in RECORDING order and the stream edges are dropped: the reference
 *                      schedule the three-stream replay must equal bit for bit (tests/test_schedule_gpu.py);
 *   fuzz_max_us > 0  : around every fork / join / edge a one-wave spin kernel of random length (0 .. fuzz_max_us microseconds,
 *                      xorshift64* seeded with `seed`) delays a random stream, the edge's source and its destination -- an ordering
 *                      that only holds because one stream happens to be ahead of another breaks under it;
 *   noise_blocks > 0 : ... and a busy kernel of noise_blocks workgroups co-runs on a stream of its own (wave timing inside the
 *                      step's kernels changes: a reduction whose order follows arrival would show).
 * tpgsr_plan_fuzz_point(stream): the same perturbation for stream edges a caller makes outside a plan.
 * tpgsr_spin: the spin kernel itself (blocks x threads busy-waiting `us` microseconds of the 100 MHz wall clock; work != 0: half of
 * the waves run FMAs meanwhile). */
void tpgsr_plan_set_mode(int serial, int fuzz_max_us, unsigned long long seed, int noise_blocks);
int tpgsr_plan_get_mode(void);
int tpgsr_plan_fuzz_point(void* stream);
int tpgsr_spin(int blocks, int threads, float us, int work, void* stream);
/* Stamp mode (un-profiled per-op time line): with tpgsr_plan_set_stamp(1) every launch of a plan is followed by a timing event on its
 * stream; tpgsr_plan_stamp_epoch(stream) records the common origin; after a device synchronisation tpgsr_plan_read_stamps fills
 * ms_out[i] = origin -> end of op i of the plan's last run (-1 for stream edges) and stream_out[i] = the HIP stream it ran on. */
int tpgsr_plan_set_stamp(int on);
int tpgsr_plan_stamp_epoch(void* stream);
int tpgsr_plan_read_stamps(void* plan, float* ms_out, long long* stream_out, int cap);
/* A HIP stream for the side-stream role, optionally confined to the compute units whose bits are set in cu_mask
 * (n_words 32-bit words, bit i of word w = CU 32*w + i; NULL / 0 = all CUs).  Returns NULL on failure. */
void* tpgsr_stream_create(const unsigned int* cu_mask, int n_words);
int tpgsr_stream_destroy(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TPGSR_HIP_H */
