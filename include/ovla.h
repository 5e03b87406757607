/*
 * ovla.h -- C-ABI of libovla_hip.so: the MI355X (gfx950) kernels behind the OpenVLA-OFT
 * parallel-decoding action-chunk forward/backward path.
 *
 * The reference (ciccio42/openvla-oft) has NO native interface for this path: every FLOP is delegated to
 * PyTorch / timm / transformers / peft.  Each entry point below therefore cites the reference
 * *Python call site* whose arithmetic it replaces (paths relative to the reference root).
 *
 * Conventions
 *  - extern "C", plain pointers and sizes; no torch types.  All device buffers are caller-owned.
 *  - every op is `int ovla_<op>(const ovla_<op>_args*, void* hip_stream)`; returns 0 or a negative OVLA_E* code,
 *    `ovla_last_error()` returns a thread-local message.  Nothing is allocated, nothing synchronises the host;
 *    all kernels are stream-ordered on the given stream.
 *  - bf16 tensors are passed as `const void*` holding IEEE bfloat16 bit patterns; row-major; leading dimensions
 *    ("ld*") are in ELEMENTS.
 *  - "rows" of a token matrix are flattened (batch, position): row = b * S + s.
 */
#ifndef OVLA_H_
#define OVLA_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OVLA_OK 0
#define OVLA_EINVAL (-1)  /* malformed arguments (shape/alignment/null) */
#define OVLA_EARCH (-2)   /* device is not gfx950 */
#define OVLA_ELAUNCH (-3) /* HIP launch / runtime error */

#define OVLA_ABI_VERSION 1

const char* ovla_last_error(void);
int ovla_abi_version(void);
/* sha256 (first 32 hex digits) of the sources, headers and build script this library was compiled from (csrc/build.sh); the Python
 * loader compares it with the files next to the library and refuses a stale build. */
const char* ovla_build_hash(void);
/* Checks that `device` is a gfx950 part; returns OVLA_OK / OVLA_EARCH / OVLA_ELAUNCH. */
int ovla_check_device(int device);

/* ------------------------------------------------------------------------------------------------------------------
 * GEMM  C[M,N] = epilogue( A[M,K] . B[N,K]^T  (+ A2[M,K2] . B2[N,K2]^T) )          bf16 in, fp32 accumulate
 *
 * Replaces every nn.Linear on the path:
 *   Llama q/k/v/o/gate/up/down  (transformers LlamaDecoderLayer, call site prismatic/extern/hf/modeling_prismatic.py:632-643)
 *   timm Block qkv/proj/fc1/fc2 and patch_embed-as-GEMM (call site modeling_prismatic.py:201-227)
 *   PrismaticProjector fc1/fc2/fc3 (modeling_prismatic.py:250-262), ProprioProjector / NoisyActionProjector
 *   (prismatic/models/projectors.py:19-24,44-49), MLPResNet fc1 / block Linear (prismatic/models/action_heads.py:72-81),
 *   FiLM scale/shift (prismatic/models/film_vit_wrapper.py:65-67),
 * and, through the (A2,B2) "K-extension", peft's LoRA update  y += (alpha/r) * B(A x)  (vla-scripts/finetune.py:862-871):
 * A2 = scaled t = s * x A^T  [M, G*r],  B2 = stacked LoRA-B [N, r]; the column block of A2 used by an output tile is
 * (n0 / k2_group_n) * K2, so fused q|k|v and gate|up linears share one launch.
 * Because frozen weights are kept both as W [out,in] and W^T [in,out] in HBM, the same "NT" kernel computes the
 * backward data gradient dX = dY . W (A = dY, B = W^T).
 *
 * Epilogue order (each step rounds to bf16 like the reference's separate PyTorch ops do):
 *   v = alpha * acc;  v += bias[n];  [C_pre = v];  v = act(v);  v *= colscale[n];  v += residual[m,n];
 *   v = v * (1 + film_gamma[m / film_rows, n]) + film_beta[...]  ->  C
 * Backward epilogues (the data-gradient GEMM applies the producing activation's derivative instead of a separate pass):
 *   dact_mode 1: C[m,n] = v * act'(dact_src[m,n])               (dact_act: GELU / ReLU / SiLU / GELU-tanh; timm Mlp, projector)
 *   dact_mode 2: SwiGLU: g = dact_src[m,n], u = dact_src[m,n+N]; C[m,n] = v*u*silu'(g), C[m,n+N] = v*silu(g); C is [M,2N]
 *     (LlamaMLP down_proj(silu(gate) * up): the gradient of the fused gate|up pre-activations)
 * Requirements: K % 8 == 0, K2 % 8 == 0, N % 8 == 0, ld* % 8 == 0, 16-byte aligned base pointers.
 * split_k > 1 needs `workspace` of ovla_gemm_workspace_bytes() bytes.
 */
enum { OVLA_ACT_NONE = 0, OVLA_ACT_GELU = 1, OVLA_ACT_RELU = 2, OVLA_ACT_SILU = 3, OVLA_ACT_GELU_TANH = 4,
       /* ovla_gemm_bf16 only, tile 22 / 122 only: B = [gate; up] stacked ([N = 2 F, K], F % 128 == 0), C is [M, F] = bf16(bf16(silu(g)) * u) with g | u the
        * bf16-rounded projection outputs -- HF LlamaMLP's act_fn(gate_proj(x)) * up_proj(x) (modeling_llama.py) without the [M, 2 F] intermediate; nothing else
        * in the epilogue but alpha and the RMSNorm-fold row scale (rowscale_part).  Same bits as ovla_gemm_bf16 + ovla_swiglu_fwd. */
       OVLA_ACT_SWIGLU = 5 };

typedef struct {
  const void* A; int64_t lda;
  const void* B; int64_t ldb;
  const void* A2; int64_t lda2;  /* optional K-extension (LoRA) */
  const void* B2; int64_t ldb2;
  void* C; int64_t ldc;          /* bf16 [M,N] */
  void* C_pre;                   /* optional bf16 [M,N] (ldc): value before the activation; with FiLM: before the modulation */
  const void* bias;              /* optional bf16 [N] */
  const void* colscale;          /* optional bf16 [N]  (timm LayerScale) */
  const void* residual; int64_t ldr; /* optional bf16 [M,N] */
  const void* film_gamma;        /* optional bf16 [M/film_rows, N]  (FiLM) */
  const void* film_beta;
  int32_t film_rows;
  int32_t M, N, K, K2;
  int32_t k2_group_n;            /* 0: A2 column offset 0 for every tile */
  int32_t a_group_n;             /* >0: block-diagonal GEMM -- output columns [g*a_group_n, (g+1)*a_group_n) contract A columns
                                    [g*K, (g+1)*K) with B rows of the same column range (per-group LoRA dt = dy_g . B_g) */
  int32_t act;
  int32_t split_k;               /* <=1: none */
  int32_t tile;                  /* 0: auto; otherwise forces a tile configuration (tests / tuning) */
  float alpha;                   /* scales the accumulator (LoRA alpha/r); 0 means 1 */
  const void* dact_src; int64_t ld_dact;   /* optional saved pre-activations for the backward epilogues above */
  int32_t dact_mode, dact_act;
  const void* rope_cos; const void* rope_sin;  /* optional bf16 [>= rope_S, 64] tables: RoPE (head_dim 128, HF rotate_half, position = */
  int32_t rope_S, rope_cols;                   /* row % rope_S) applied to output columns [0, rope_cols) -- the q | k heads of a fused
                                                  q|k|v projection (modeling_llama apply_rotary_pos_emb).  Excludes the other epilogues;
                                                  fused into the 256x256 config's epilogue, otherwise one extra ovla_rope launch. */
  /* RMSNorm folded around the GEMM (the decoder's input_layernorm / post_attention_layernorm, HF LlamaRMSNorm at modeling_prismatic.py:632-643,
   * on the merged inference path; the norm WEIGHT is folded into B offline, B' = bf16(B * w[k])):
   *   producer side  rowsq_out [M, N/64] fp32: sum of squares of every 64-column group of the bf16 OUTPUT row (plain stores, one writer per
   *                  slot: summed later in slot order -- deterministic);
   *   consumer side  rowscale_part [M, rowscale_slots] fp32 = the producer's rowsq_out for this GEMM's A operand (rowscale_slots * 64 = K):
   *                  C = epilogue(rstd[m] * alpha * acc), rstd[m] = rsqrt(sum(slots) / K + rowscale_eps); rowscale_r [M] fp32 scratch
   *                  receives rstd (the hybrid-remainder reduce reads it).
   * Supported on the 128x128 tile only (tile 1 / 101, what the batch-1 shapes M <= ~1k resolve to); any other schedule is an error. */
  float* rowsq_out;
  const float* rowscale_part; int32_t rowscale_slots; float rowscale_eps; float* rowscale_r;
  /* In-launch reduce of the hybrid schedule's K-split remainder tiles: device array of n_hybrid_counters uint32, ALL ZERO on entry (the kernel
   * leaves it all zero).  The last of a tile's K-part workgroups to arrive sums the slabs in slab order and runs the epilogue, so no separate
   * reduce launch is needed; same bits either way.  NULL / too few counters: a gemm_hybrid_reduce launch follows as before. */
  uint32_t* hybrid_counters; int32_t n_hybrid_counters;
  void* workspace;               /* fp32 scratch: [split_k, M, N] when split_k > 1; also enables the auto schedules */
  int64_t workspace_bytes;       /* (hybrid remainder split, skinny-N split-K) when tile == 0; may be NULL/0 */
} ovla_gemm_args;

/* The schedule `tile = 0` would pick for a dense GEMM (no skinny / small-M special case): tile id (17 = 256x256, 1 = 128x128,
 * 2 = 64x128, 5 = 128x32), how many tiles run whole, how many are split along K and how many ways, and the cost model's estimate.
 * (A launch that resolves to 17 runs the 256x256 tile's 4-wave configuration, id 18, when K >= 4096 is a multiple of 64, the LoRA rank is 0 or 32
 * and the epilogue is alpha / bias / residual only -- same tile grid, same remainder split, sums added in a different order.)
 * Host-only (no launch): lets callers and tests inspect the decision. */
int ovla_gemm_plan(int32_t M, int32_t N, int32_t K, int32_t K2, int32_t k2_group_n, int64_t workspace_bytes, int32_t* tile,
                   int32_t* full_tiles, int32_t* rem_tiles, int32_t* rem_splits, double* est_seconds);
/* The tile configuration ovla_gemm_bf16 would run for exactly these arguments (same checks, same decision code, no launch): what profilers and
 * bench.py use to name the kernel instance of a launch.  6 = the single-launch skinny kernel. */
int ovla_gemm_resolved_tile(const ovla_gemm_args* a, int32_t* tile);
int64_t ovla_gemm_workspace_bytes(int32_t M, int32_t N, int32_t split_k);
int ovla_gemm_bf16(const ovla_gemm_args* a, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * "TN" GEMM (weight gradients)   C[P,Q] (+)= alpha * X[M,P]^T . Y[M,Q]     bf16 in, fp32 accumulate
 * Replaces autograd's weight-gradient matmuls for the trainable tensors: LoRA A/B (peft, finetune.py:862-871),
 * L1RegressionActionHead / ProprioProjector / NoisyActionProjector / FiLM Linears (finetune.py:894-932).
 * out_mode 0: fp32 atomicAdd into C (C must hold the running sum; the M range is split over workgroups)
 *          1: fp32 store    2: bf16 store   (modes 1/2: no M split)
 */
typedef struct {
  const void* X; int64_t ldx;
  const void* Y; int64_t ldy;
  void* C; int64_t ldc;
  int32_t M, P, Q;
  float alpha;
  int32_t out_mode;
} ovla_gemm_tn_args;
int ovla_gemm_tn_bf16(const ovla_gemm_tn_args* a, void* stream);
/* Up to OVLA_TN_MAX_GROUP independent TN problems in ONE launch (the dA / dB_g products of one LoRA linear): the small
 * problems then overlap on the chip instead of running back to back. */
#define OVLA_TN_MAX_GROUP 4
int ovla_gemm_tn_grouped(const ovla_gemm_tn_args* problems, int32_t n, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * LoRA backward, the two products that stream dy, in ONE pass over it (peft LoRA r = 32 on every nn.Linear, finetune.py:862-871; autograd's
 * backward of `result + lora_B(lora_A(x)) * scaling`):
 *     dt[:, g r : (g+1) r]  = bf16(scale * dy_g . B_g)        dy_g = dy[:, g gn : (g+1) gn], B_g [gn, r] given transposed (Bt rows g r ..)
 *     dB[g gn : (g+1) gn]  += dy_g^T . t[:, g r : (g+1) r]    fp32, atomically accumulated (dB must hold the running sum)
 * for the G fused groups of one adapted Linear (q|k|v: 3, gate|up: 2).  dt partial sums of the 256-column chunks go through the fp32
 * workspace and are added in a fixed order: dt is bit-reproducible run to run.  r must be 32. */
typedef struct {
  const void* dy; int64_t ld_dy;   /* bf16 [M, G*gn] */
  const void* Bt; int64_t ld_bt;   /* bf16 [G*r, gn]: B_g^T stacked */
  const void* t;  int64_t ld_t;    /* bf16 [M, G*r]: the forward's saved  scale * x . A_g^T */
  void* dt; int64_t ld_dt;         /* out bf16 [M, G*r] */
  float* dB; int64_t ld_db;        /* in/out fp32 [G*gn, r] */
  int32_t M, gn, G, r;
  float scale;
  void* workspace; int64_t workspace_bytes;   /* >= ovla_lora_bwd_workspace_bytes(M, gn, G), 16-byte aligned */
} ovla_lora_bwd_args;
int64_t ovla_lora_bwd_workspace_bytes(int32_t M, int32_t gn, int32_t G);
int ovla_lora_bwd(const ovla_lora_bwd_args* a, void* stream);

/* column sums  out[n] (+)= sum_m X[m,n]   (bias gradients).  out fp32, atomic accumulate. */
typedef struct { const void* X; int64_t ldx; float* out; int32_t M, N; } ovla_colsum_args;
int ovla_colsum_bf16(const ovla_colsum_args* a, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Attention (no KV paging; whole short context; K/V tiles staged through LDS)
 *   O[b,s,h,:] = softmax_k( scale * Q[b,s,h,:] . K[b,k,h,:]  + mask ) V[b,k,h,:]
 * Replaces F.scaled_dot_product_attention inside timm Attention (ViT, unmasked) and inside the transformers-fork
 * LlamaAttention (bidirectional + key padding: reference pyproject.toml:50, modeling_prismatic.py:742; `causal`
 * selects the stock lower-triangular mask).  Key padding is right padding only (reference collator
 * prismatic/util/data_utils.py:115): keys >= kv_len[b] are masked; kv_len == NULL means all S keys.
 * Q/K/V/O are addressed as  base + (b*S + s) * row_stride + h * head_dim  (elements), so the fused QKV GEMM output is
 * consumed in place.  head_dim in {64, 72, 128}.  lse: fp32 [B, H, S] (natural-log sum-exp of the scaled scores).
 */
typedef struct {
  const void* Q; const void* K; const void* V; int64_t q_stride, k_stride, v_stride;
  void* O; int64_t o_stride;
  float* lse;
  const int32_t* kv_len;  /* optional [B] */
  int32_t B, H, S, head_dim;
  int32_t causal;
  float scale;
} ovla_attn_fwd_args;
int ovla_attn_fwd(const ovla_attn_fwd_args* a, void* stream);

typedef struct {
  const void* Q; const void* K; const void* V; int64_t q_stride, k_stride, v_stride;
  const void* O; const void* dO; int64_t o_stride, do_stride;
  const float* lse;
  float* delta;            /* workspace fp32 [B,H,S]: rowsum(dO * O), written by the dQ kernel, read by the dK / dV kernel */
  void* dQ; void* dK; void* dV; int64_t dq_stride, dk_stride, dv_stride;
  const int32_t* kv_len;
  int32_t B, H, S, head_dim;
  int32_t causal;
  float scale;
  const void* rope_cos;    /* optional bf16 [S, head_dim/2] tables (ovla_rope_table): dQ and dK are written with the INVERSE RoPE rotation */
  const void* rope_sin;    /* applied (the gradient w.r.t. the pre-RoPE q / k: saves the separate ovla_rope(inverse) pass); head_dim 128 */
} ovla_attn_bwd_args;
int ovla_attn_bwd(const ovla_attn_bwd_args* a, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Normalisation.  RMSNorm: transformers LlamaRMSNorm (fp32 math, eps 1e-5).  LayerNorm: timm blocks (eps 1e-6) and
 * MLPResNet (action_heads.py:64,69, eps 1e-5).  x/y bf16 [rows, dim]; weight/bias bf16 [dim] (bias NULL for RMS).
 * rstd (and mean) fp32 [rows] are saved for the backward.  The backward returns dx (bf16); dw/db (fp32, atomic
 * accumulate) only when non-NULL (they are frozen on the LoRA path except in the action head).
 * If `dx_accum` != 0 the backward ADDS into dx (residual-stream gradient) instead of overwriting it.
 */
typedef struct {
  const void* x; void* y; const void* weight; const void* bias;
  float* mean; float* rstd;
  int32_t rows, dim; float eps; int32_t is_rms;
} ovla_norm_fwd_args;
int ovla_norm_fwd(const ovla_norm_fwd_args* a, void* stream);

typedef struct {
  const void* x; const void* dy; const void* weight;
  const float* mean; const float* rstd;
  void* dx; float* dweight; float* dbias;
  int32_t rows, dim; int32_t is_rms; int32_t dx_accum;
} ovla_norm_bwd_args;
int ovla_norm_bwd(const ovla_norm_bwd_args* a, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * RoPE, in place on the q and k heads of a fused [rows, ld] buffer (HF "rotate_half" convention; position = row % S).
 * cos/sin tables are bf16 [S, head_dim/2], built once by ovla_rope_table (fp32 angles, then cast to bf16 exactly as
 * transformers LlamaRotaryEmbedding does before apply_rotary_pos_emb).  `inverse` applies the transposed rotation
 * (backward).  n_heads = q heads + k heads, contiguous from column 0.
 */
int ovla_rope_table(void* cos_table, void* sin_table, int32_t S, int32_t head_dim, float theta, void* stream);
typedef struct {
  void* qk; int64_t ld; int32_t rows, S, n_heads, head_dim;
  const void* cos_table; const void* sin_table; int32_t inverse;
} ovla_rope_args;
int ovla_rope(const ovla_rope_args* a, void* stream);

/* SwiGLU:  h = silu(g) * u  with gu = [g | u] as [rows, 2F];  backward writes dgu from dh and gu. */
typedef struct { const void* gu; void* h; int32_t rows, F; } ovla_swiglu_fwd_args;
int ovla_swiglu_fwd(const ovla_swiglu_fwd_args* a, void* stream);
typedef struct { const void* gu; const void* dh; void* dgu; int32_t rows, F; } ovla_swiglu_bwd_args;
int ovla_swiglu_bwd(const ovla_swiglu_bwd_args* a, void* stream);

/* activation backward:  dz = dh * act'(z)   (z = saved pre-activation), elementwise on [n] bf16 */
typedef struct { const void* z; const void* dh; void* dz; int64_t n; int32_t act; } ovla_act_bwd_args;
int ovla_act_bwd(const ovla_act_bwd_args* a, void* stream);

/* y = alpha*x (+ y if accum)   and friends: small elementwise helpers on bf16 buffers */
typedef struct { const void* a; const void* b; void* out; int64_t n; } ovla_add_args;
int ovla_add_bf16(const ovla_add_args* a, void* stream);  /* out = a + b (b may be NULL: copy) */
typedef struct { const void* x; const void* scale; void* out; int32_t rows, dim; } ovla_colscale_args;
int ovla_colscale_bf16(const ovla_colscale_args* a, void* stream); /* out[m,n] = x[m,n]*scale[n] */

/* ------------------------------------------------------------------------------------------------------------------
 * Vision front end.
 * im2col for the 14x14/stride-14 patch embedding (timm PatchEmbed Conv2d as GEMM): pixel_values bf16 NCHW
 * [B, C_total, H, W]; image `img` (0..n_img-1) of a batch row uses channels [c0 + img*img_cstride, +3)
 * (modeling_prismatic.py:210-224: 6 channels per image, DINOv2 first, SigLIP second) -> rows [(B*n_img)*gh*gw, ldo] with
 * K = 3*p*p real columns (c, py, px) and zero pad.  n_img == 0 means 1.
 */
typedef struct { const void* pixels; void* out; int64_t ldo; int32_t B, C_total, c0, H, W, patch, n_img, img_cstride; } ovla_im2col_args;
int ovla_im2col(const ovla_im2col_args* a, void* stream);

/* Inference-side image preparation as ONE kernel (experiments/robot/openvla_utils.py:542-622 `center_crop_image`:
 * convert_image_dtype(u8 -> f32) = x * (1/255); tf.image.crop_and_resize of the central sqrt(crop_scale) box back to out x out
 * (TF 2.15 CropAndResize arithmetic in fp32: in_y = y1 (H-1) + i (y2-y1)(H-1)/(out-1), lerp form a + (b-a) w); clip; back to uint8
 * as trunc(x * 255.5); then prismatic/extern/hf/processing_prismatic.py:128-145 `apply_transform`: to_tensor (/255) and
 * per-backbone normalise, channel-stacked).  src uint8 [n_img, H, W, 3] (HWC, device) -> dst bf16 [n_img * 6, out, out]:
 * image i -> channels [6 i, 6 i + 3) normalised with mean[0..2]/std[0..2] (DINOv2) and [6 i + 3, 6 i + 6) with mean[3..5]/std[3..5]
 * (SigLIP).  crop == 0: the pixels are taken as they are (requires H == W == out).  Every fp32 operation is individually
 * rounded (no FMA contraction): bit-exact against the host restatement (image_prep.center_crop_image + apply_transform). */
typedef struct { const void* src; void* dst; int32_t n_img, H, W, out, crop; float crop_scale; float mean[6]; float std[6]; } ovla_image_prep_args;
int ovla_image_prep(const ovla_image_prep_args* a, void* stream);

/* Separable resampling with per-output-pixel spans (TF 2.15 `scale_and_translate_op.cc`, the kernel behind
 * tf.image.resize(method="lanczos3", antialias=True) that experiments/robot/openvla_utils.py:516-540 `resize_image_for_policy` and
 * dlimp's `resize_image` (rlds/obs_transforms.py:83) call): rows first into an fp32 intermediate [n, out_h, W, 3], then columns;
 * out = sum_k weights[o, k] * in[starts[o] + k] accumulated in span order from 0, each multiply and add rounded on its own; then
 * tf.round (half to even), clip to [0, 255], uint8.  The spans (starts int32 [out], weights fp32 [out, span]) are computed by the host
 * (image_prep.lanczos3_spans) and passed as device arrays.  The reference's JPEG encode/decode round trip before the resize is
 * ovla_jpeg_roundtrip below; PARITY UNPINNED against TensorFlow, bit-compared with oracle/data_oracle.py. */
typedef struct {
  const void* src; void* dst; void* workspace; int64_t workspace_bytes;
  const int32_t* row_starts; const float* row_weights; const int32_t* col_starts; const float* col_weights;
  int32_t n_img, H, W, out_h, out_w, row_span, col_span;
} ovla_image_resize_args;
int64_t ovla_image_resize_workspace_bytes(int32_t n_img, int32_t W, int32_t out_h);
int ovla_image_resize(const ovla_image_resize_args* a, void* stream);

/* The JPEG encode -> decode round trip of `resize_image_for_policy` (experiments/robot/openvla_utils.py:532-533: tf.image.encode_jpeg(img)
 * then tf.io.decode_image(...), default arguments = libjpeg-turbo baseline 4:2:0 at quality 95, accurate integer DCT, fancy upsampling) as
 * two launches, without the (lossless) entropy coder: per 16 x 16 MCU RGB -> YCbCr, 2x2 chroma box filter, 8x8 forward DCT, quantise,
 * dequantise, inverse DCT -> component planes in `workspace`; then triangle-filter chroma upsampling + YCbCr -> RGB per pixel.  Integer
 * arithmetic throughout, bit-identical to oracle/jpeg_oracle.py, which is pinned to libjpeg-turbo's own output (tests/golden/g12).
 * src / dst uint8 [n_img, H, W, 3] (any H, W >= 1: edges are padded by replication as the library does); quality 1..100. */
typedef struct { const void* src; void* dst; void* workspace; int64_t workspace_bytes; int32_t n_img, H, W, quality; } ovla_jpeg_roundtrip_args;
int64_t ovla_jpeg_roundtrip_workspace_bytes(int32_t n_img, int32_t H, int32_t W);
int ovla_jpeg_roundtrip(const ovla_jpeg_roundtrip_args* a, void* stream);

/* Training-time image path as two launches (replaces the TF/dlimp frame transform + PIL/torchvision processor of the reference's data
 * loader: prismatic/vla/datasets/rlds/obs_transforms.py:18-45 `augment` -> dlimp `augment_image` with the kwargs of
 * prismatic/vla/datasets/datasets.py:159-174, then prismatic/extern/hf/processing_prismatic.py:128-145 `apply_transform`).
 * Per image, fp32, every operation individually rounded (no FMA contraction), `clip(0, 1)` after every enabled op:
 *   x = u8 / 255
 *   bit 0  crop_and_resize(box = params[0..3] = y1, x1, y2, x2, to out x out)   TF CropAndResize, bilinear, extrapolation 0
 *   bit 1  x + params[4]                                                        tf.image.adjust_brightness
 *   bit 2  (x - mean_c) * params[5] + mean_c, mean over the image per channel   AdjustContrastv2 (mean accumulated in fp64 here)
 *   bit 3  rgb -> hsv, s = clamp(s * params[6], 0, 1), hsv -> rgb                adjust_saturation_op.cc (CPU kernel arithmetic)
 *   bit 4  rgb -> (h, v_min, v_max), h += 6 * params[7] wrapped to [0, 6), back  adjust_hue_op.cc (CPU kernel arithmetic)
 *   u8 = trunc(x * 255); then to_tensor (/255) and the two backbones' normalisations as in ovla_image_prep.
 * ops_mask == 0 is the evaluation path (quantise + normalise only; H == W == out required unless bit 0 is set).
 * src uint8 [n_img, H, W, 3] (device) ; params fp32 [n_img, 8] (device) ; dst bf16 [n_img, 6, out, out] ;
 * workspace: ovla_image_augment_workspace_bytes(n_img, out) bytes (fp32 intermediate image + per-block fp64 channel sums).
 * TensorFlow / dlimp are not available offline: PARITY UNPINNED against TF, bit-compared with oracle/data_oracle.py. */
typedef struct {
  const void* src; void* dst; const float* params; void* workspace; int64_t workspace_bytes;
  int32_t n_img, H, W, out, ops_mask; float mean[6]; float std[6];
} ovla_image_augment_args;
int64_t ovla_image_augment_workspace_bytes(int32_t n_img, int32_t out);
int ovla_image_augment(const ovla_image_augment_args* a, void* stream);

/* tokens[b, pre + i, :] = patches[b, i, :] + pos[i, :] ;  tokens[b, j, :] = prefix[j, :]  (cls / register tokens)
 * (timm VisionTransformer._pos_embed with no_embed_class=True) */
typedef struct { const void* patches; const void* pos; const void* prefix; void* tokens; int32_t B, n_patches, n_prefix, dim; } ovla_vit_embed_args;
int ovla_vit_embed(const ovla_vit_embed_args* a, void* stream);
/* strided row copy: dst[b, i, dst_col0 : dst_col0+dim] = src[b, src_row0 + i, :]   (drop prefix tokens + feature concat,
 * modeling_prismatic.py:221-227); `accumulate` adds instead (used by the backward). */
typedef struct {
  const void* src; void* dst; int32_t B, rows, dim;
  int64_t src_batch_stride, src_row0, src_ld, dst_batch_stride, dst_row0, dst_ld, dst_col0; int32_t accumulate;
} ovla_copy_rows_args;
int ovla_copy_rows(const ovla_copy_rows_args* a, void* stream);

/* FiLM backward (prismatic/models/film_vit_wrapper.py:72: x = x_pre * (1 + gamma[b]) + beta[b], b = row / rows_per_batch):
 *   dgamma[b,n] += sum_rows dy * x_pre ;  dbeta[b,n] += sum_rows dy ;  dy <- dy * (1 + gamma[b,n])   (in place)
 * dgamma / dbeta fp32 [B, dim] (accumulated), gamma bf16 [B, dim]. */
typedef struct { void* dy; const void* x_pre; const void* gamma; float* dgamma; float* dbeta; int32_t B, rows_per_batch, dim; } ovla_film_bwd_args;
int ovla_film_bwd(const ovla_film_bwd_args* a, void* stream);

/* mean pooling:  out[b,:] = mean over rows i with row_mask[b,i] != 0 of x[b, i, :]  (FiLM's average language embedding) */
typedef struct { const void* x; const uint8_t* row_mask; void* out; int32_t B, L, dim; } ovla_masked_mean_args;
int ovla_masked_mean(const ovla_masked_mean_args* a, void* stream);

/* FiLM's conditioning vector straight from the token ids (modeling_prismatic.py:575-583: `input_embeddings[~all_actions_mask]`
 * averaged over the sequence, film_vit_wrapper.py:243): out[b,:] = bf16(mean_i embed[ids[b,i], :]) over the text positions i whose label
 * is NOT an action token (labels[b,i] <= action_token_begin, which includes IGNORE_INDEX: BOS, prompt, stop and pad positions all
 * count, as in the reference).  The action mask is the same integer rule as ovla_assemble_multimodal's (an action token is always a
 * counted label, so the cumulative-count clause of train_utils.py:8-39 cannot change it).  fp32 sum of the bf16 rows, one rounding.
 * ids / labels int64 [B, L]; embed bf16 [vocab, D]; out bf16 [B, D].  No host synchronisation: capturable in a hipGraph. */
typedef struct { const int64_t* ids; const int64_t* labels; const void* embed_table; void* out; int32_t B, L, D, vocab; int64_t action_token_begin; } ovla_language_average_args;
int ovla_language_average(const ovla_language_average_args* a, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Multimodal sequence assembly (modeling_prismatic.py:571-629): one pass writes
 *   out[b, 0]            = embed[ids[b,0]]
 *   out[b, 1 .. P]       = patches[b, :]            (P = projected patches + proprio (+ timestep) tokens)
 *   out[b, P+1+i]        = action_mask[b,1+i] ? (noisy ? noisy[b, slot] : 0) : embed[ids[b, 1+i]]
 * action_mask is computed on device from labels exactly as train_utils.get_current_action_mask | get_next_actions_mask
 * (cumsum of labels != -100, labels > 31743); `action_pos` receives, per batch row and action slot, the flattened row
 * b*(P+L) + (P + i) - 1 of the assembled sequence whose final hidden state PREDICTS that slot (i = text index of the
 * slot; shift-by-one of finetune.py:385-394 / modeling_prismatic.py:915-920), ready for ovla_gather_rows.
 */
typedef struct {
  const int64_t* ids; const int64_t* labels;   /* [B, L] */
  const void* embed_table;                      /* bf16 [V, D] */
  const void* patches;                          /* bf16 [B, P, D] */
  const void* noisy;                            /* optional bf16 [B, A, D] */
  void* out;                                    /* bf16 [B, P+L, D] */
  int32_t* action_pos;                          /* optional int32 [B, A]: predicting row of each action slot */
  int32_t B, L, P, D, A, vocab;
  int32_t ignore_index, action_token_begin, action_dim;
} ovla_assemble_args;
int ovla_assemble_multimodal(const ovla_assemble_args* a, void* stream);

/* Per-row sums of squares of every 64-column group: out[m, j] = sum x[m, 64 j .. 64 j + 63]^2 (fp32, [rows, dim / 64]) -- the RMSNorm-fold
 * input of the FIRST decoder layer (ovla_gemm_args.rowscale_part); later layers get theirs from the producing GEMM's epilogue (rowsq_out). */
int ovla_row_sumsq(const void* x, int64_t ld, float* out, int32_t rows, int32_t dim, void* stream);

/* dst[i, :] = src[index[i], :]  /  scatter-add backward  */
typedef struct { const void* src; const int32_t* index; void* dst; int32_t n, dim; int64_t src_ld, dst_ld; int32_t scatter_add; } ovla_gather_rows_args;
int ovla_gather_rows(const ovla_gather_rows_args* a, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * L1 action head tail (prismatic/models/action_heads.py:69-81, finetune.py:400):
 *   pred[m, :adim] = x[m,:] . W[adim, dim]^T + b ;  loss = mean |pred - target|
 * and its backward (dpred = sign(pred-target)/count * dloss -> dx, dW, db).  The wide layers of the head run on
 * ovla_gemm_bf16 / ovla_norm_*; this is the N = ACTION_DIM tail that does not fit an MFMA tile.
 */
/* Next-token cross entropy on selected rows (the discrete action-token objective: vla-scripts/finetune.py:357-359 `loss =
 * output.loss`, i.e. the transformers LlamaForCausalLM loss: logits.float(), shift by one, CrossEntropyLoss(ignore_index=-100,
 * mean) -- the caller gathers the rows whose shifted label is not ignored, so only those rows reach lm_head).
 * logits bf16 [rows, ld] (read as fp32, as `.float()` does) ; targets int64 [rows] in [0, vocab)
 *   loss_rows[r] = logsumexp(logits[r, :vocab]) - logits[r, target[r]]   (fp32)
 *   argmax[r]    = lowest index of the row maximum                       (predicted_token_ids, finetune.py:358)
 *   dlogits[r,j] = bf16((softmax(logits[r])[j] - [j == target[r]]) * grad_scale)   (optional; may alias logits)
 * grad_scale = loss_scale / number of non-ignored tokens in the batch (the mean's 1/N and the grad-accumulation divide). */
typedef struct {
  const void* logits; int64_t ld; const int64_t* targets; float* loss_rows; int32_t* argmax; void* dlogits; int64_t ld_d;
  int32_t rows, vocab; float grad_scale;
} ovla_token_ce_args;
int ovla_token_ce(const ovla_token_ce_args* a, void* stream);

typedef struct {
  const void* x; const void* W; const void* b; void* pred;   /* bf16 */
  const void* target; float* loss_sum;                        /* optional: accumulates sum |pred-target| (or squared, mse) */
  int32_t rows, dim, adim;
  int32_t mse;
} ovla_head_out_fwd_args;
int ovla_head_out_fwd(const ovla_head_out_fwd_args* a, void* stream);
typedef struct {
  const void* x; const void* W; const void* pred; const void* target;
  const void* dpred;            /* optional bf16 [rows, adim]: explicit upstream gradient (pred/target/mse then ignored) */
  float dloss_scale;            /* dloss / (rows*adim) */
  int32_t mse;                  /* 0: L1 (sign), 1: MSE (2*(pred-target)) */
  void* dx; float* dW; float* db;
  int32_t rows, dim, adim;
} ovla_head_out_bwd_args;
int ovla_head_out_bwd(const ovla_head_out_bwd_args* a, void* stream);

/* The fused L1 / diffusion action-head tail (`north_star`: "a fused L1 action-head"): everything of MLPResNet.forward after fc1
 * (prismatic/models/action_heads.py:72-81, 49-56) and the loss of finetune.py:400 / :407 in ONE launch, for up to 64 rows (batch 8 x chunk 8):
 *     for b in 0, 1:   x <- x + relu(fc_b(LayerNorm_b(x)))           (MLPResNetBlock: ffn = LN -> Linear -> ReLU, residual outside)
 *     h2 = LayerNorm_2(x);  pred = fc2(h2);  loss_sum += sum |pred - target|   (or squared, mse)
 * dim / 16 <= 256 workgroups, all resident.  Five stages: LayerNorm rows of block 0 | GEMM 0 + epilogue | LayerNorm rows of block 1 | GEMM 1 +
 * epilogue | LayerNorm 2 + fc2 + loss.  LayerNorm stages deal ROWS over the workgroups and write the normalised rows (hb: the Linear's saved
 * input); GEMM stages deal 16-column strips: a workgroup streams ITS 16 weight rows from HBM once (each 33.5 MB matrix crosses HBM once,
 * spread over the chip) and reads the R x dim normalised rows from L2, one wave per 16-row MFMA tile.  The stages are separated by four
 * grid-wide barriers: agent-scope release / arrival counter / bounded spin / agent-scope acquire (the arrival counter sync[0] is zeroed by a memset
 * node ahead of the launch; a spin that runs out fills pred, loss_sum and every saved buffer with NaN and sets sync[1], which NO launch clears:
 * the host reads it where it already synchronises).  The launch is refused unless the whole grid can be co-resident
 * (ovla_head_tail_resident_blocks() = occupancy x compute units >= dim / 16 workgroups).
 * Arithmetic, rounding points AND summation orders are those of the unfused sequence (ovla_norm_fwd, ovla_gemm_bf16 with split_k = 2 and
 * its reduce epilogue, ovla_head_out_fwd): results are bit-identical to it (tests/test_kernels_gpu.py).
 * Everything the backward needs is written out when the pointers are given (training): per block the LayerNorm output hb (the Linear's
 * saved input), the pre-activation zb and mean / rstd; h2, mean2, rstd2.  rows = R must be a multiple of 16 (<= 64; pad with zero rows),
 * rows_real <= R rows enter pred / loss.  All [R, dim] buffers contiguous bf16; W_b [dim, dim] row-major [out, in]. */
typedef struct {
  const void* x0;                                   /* [R, dim] input of block 0 (= relu(fc1(LN1(.)))) */
  const void* ln_w[2]; const void* ln_b[2]; const void* W[2]; const void* bias[2];
  void* hb[2]; void* zb[2]; void* xo[2];            /* hb (LayerNorm output) and xo (block output) required; zb optional (training) */
  float* mean[2]; float* rstd[2];                   /* optional, [R] */
  const void* ln2_w; const void* ln2_b; void* h2; float* mean2; float* rstd2;   /* h2 required ([R, dim]); stats optional */
  const void* W2; const void* b2; void* pred; const void* target; float* loss_sum;   /* W2 [adim, dim]; target / loss_sum optional */
  uint32_t* sync;                                   /* device, >= 2 words: arrival counter (reset per launch), STICKY timeout flag */
  int32_t rows, rows_real, dim, adim, mse, ksplit;  /* ksplit 1 | 2: accumulate K in that many halves, summed in order (= split_k of the unfused GEMM) */
  float eps;
} ovla_head_tail_args;
int ovla_head_tail_resident_blocks(void);          /* workgroups of the fused tail that fit on the current device at once (0: unknown / no device) */
int ovla_head_tail_fwd(const ovla_head_tail_args* a, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Fused AdamW over a flat parameter buffer (torch.optim.AdamW as called at finetune.py:952: betas (0.9,0.999),
 * eps 1e-8, weight_decay 0.01).  bf16 variant reproduces torch's per-op bf16 rounding sequence
 * (param *= 1-lr*wd; m.lerp_(g,1-b1); v = v*b2 + (1-b2) g*g; denom = sqrt(v)/sqrt(bc2) + eps; p -= (lr/bc1) m/denom)
 * so results are bit-identical to torch on the same inputs.  grads are fp32 master-accumulated here and are rounded to
 * the parameter dtype first (what autograd would have stored).  grad_scale multiplies the gradient (1/world, 1/accum).
 */
typedef struct {
  void* param; void* exp_avg; void* exp_avg_sq;   /* bf16 or fp32 (is_bf16) */
  const float* grad;
  int64_t n; int32_t is_bf16; int32_t step;       /* step >= 1 (after increment, as torch) */
  double lr, beta1, beta2, eps, weight_decay;     /* Python floats, combined in double like torch/optim/adamw.py */
  float grad_scale;
} ovla_adamw_args;
int ovla_adamw(const ovla_adamw_args* a, void* stream);

/* fp32 -> bf16 convert with scale (publishes .grad views), and fill */
typedef struct { const float* src; void* dst; int64_t n; float scale; } ovla_cvt_args;
int ovla_cvt_f32_to_bf16(const ovla_cvt_args* a, void* stream);
int ovla_cvt_bf16_to_f32(const void* src, float* dst, int64_t n, float scale, void* stream);

/* transpose bf16 [rows, cols] -> [cols, rows] (W^T copies of the frozen weights, LoRA A^T/B^T refresh) */
typedef struct { const void* src; void* dst; int32_t rows, cols; int64_t lds, ldd; } ovla_transpose_args;
int ovla_transpose_bf16(const ovla_transpose_args* a, void* stream);
/* Many transposes in one launch: `table` is a DEVICE array of n ovla_transpose_args (built once: the LoRA factors and
 * their derived A^T / B^T copies never move); `tile_start` is a DEVICE int32 array of n+1 exclusive prefix sums of the
 * entries' 64x64 tile counts, total_tiles = tile_start[n]. */
int ovla_transpose_batched(const ovla_transpose_args* table, const int32_t* tile_start, int32_t n, int32_t total_tiles, void* stream);

/* self-test hook used by tests/: dumps the MFMA / transposed-LDS-read / LDS-DMA lane layouts the kernels rely on.
 * out: fp32 [4, 64, 16] device;  src: bf16 [512] device holding 0..511. */
int ovla_selftest_layouts(float* out, const void* src, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OVLA_H_ */
