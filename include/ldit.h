/*
 * ldit.h - C ABI of libldit_hip.so: the MI355X (gfx950) ViT / DiT encoder forward behind LayoutDiT's DiTBackbone.
 *
 * The reference has no FFI: its seam is the Python attribute `DiTBackbone.dit`, a HuggingFace `BeitModel`
 * (ref src/layoutdit/modeling/dit_backbone.py:26-31) whose only use is
 *     hs = self.dit(x).hidden_states            (ref src/layoutdit/modeling/dit_backbone.py:47)
 * Every entry point below replaces a piece of what that one line executes; the citations name the reference (ref:)
 * or the third-party code it delegates to (TF: = transformers models/beit/modeling_beit.py, pinned 4.49.0 at
 * ref uv.lock:1771-1772; line numbers from the installed 5.15.0 copy).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - plain C, no torch / HIP types in signatures; `ldit_stream` is a hipStream_t passed as void* (NULL = default stream)
 *   - every pointer is a DEVICE pointer owned by the caller, 16-byte aligned, fp32 unless stated
 *   - the library allocates nothing and only ENQUEUES work on `stream` (asynchronous to the host; the caller synchronises)
 *     - safe to capture in a hipGraph.  Its only mutable process state: one bit per (kernel, device ordinal) recording that
 *     the kernel's dynamic-LDS limit was raised on that device (set lazily on the CURRENT device of the calling thread, which
 *     must be the device `stream` belongs to), and the diagnostic switches below (read from the environment once)
 *   - returns LDIT_OK (0) or a negative LDIT_E* code; ldit_last_error() returns a thread-local message
 *   - there is NO CPU fallback in this library: without a HIP device every compute entry point fails with LDIT_EHIP
 */
#ifndef LDIT_H
#define LDIT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LDIT_ABI_VERSION 5
#define LDIT_MAX_TAPS 8

enum ldit_status {
    LDIT_OK = 0,
    LDIT_EINVAL = -1,       /* bad argument (null pointer, misaligned, inconsistent geometry) */
    LDIT_EWORKSPACE = -2,   /* workspace / packed buffer too small */
    LDIT_EHIP = -3,         /* HIP runtime error (message in ldit_last_error) */
    LDIT_EUNSUPPORTED = -4  /* geometry outside what the kernels handle (see each function) */
};

enum ldit_dtype {
    LDIT_F32 = 0,  /* everything fp32 (exact-fp32 MFMA) */
    LDIT_BF16 = 1, /* GEMM / attention operands bf16, fp32 accumulation, residual stream / LayerNorm / softmax fp32; taps fp32 */
    LDIT_FP8 = 3,  /* BASELINE.json configs[4]: the four GEMMs of a layer on fp8 e4m3 (OCP) operands with per-tensor scales,
                      fp32 accumulation; attention on bf16 q|k|v; residual stream / LayerNorm / softmax fp32; taps fp32.
                      Weight scales (one per output channel) are measured by ldit_pack_weights; the four per-tensor
                      activation scales per layer come from
                      ldit_set_fp8_act_scales (calibration is the caller's job) */
    LDIT_F32X3 = 4, /* fp32 forward with every GEMM operand held as TWO bf16 planes x ~= p0 + p1 (p0 = bf16(x), p1 = bf16(x - p0):
                      16 significant bits) and every product computed as p1.q0 + p0.q1 + p0.q0 on the bf16 MFMA (16x the fp32
                      matrix rate) with fp32 accumulation - "bf16x3" -, the two products of the attention included.  Residual stream, LayerNorm,
                      softmax, erf-GELU, biases and taps are fp32 exactly as in LDIT_F32.  Whole-path error vs float64: 4.4 - 5.6e-6
                      relative L2 (LDIT_F32: 7 - 9e-7), i.e. inside the fp32 build's own parity gates (2e-5), 180x inside the north-star 1e-3.
                      The error of a dot product is 2^-17 of its TERMS: with activation outliers (LayerNorm channels x 60) the worst
                      element reaches 1.2e-3 of max(|ref|, 1) while the relative L2 stays 1.6e-5 - check parity per checkpoint. */
    LDIT_F32X6 = 5  /* the same with THREE planes (24 significant bits) and the six plane products down to 2^-24: fp32-grade error
                      (6 - 8e-7 vs float64: under the fp32 MFMA path's on every tap, at that of ATen's CPU fp32 forward) at 6 bf16 MFMAs per product */
};

/* order of the per-layer activation scales handed to ldit_set_fp8_act_scales (scale = amax / 448) */
enum ldit_fp8_act {
    LDIT_FP8_A_LN1 = 0,   /* layernorm_before output -> q|k|v GEMM */
    LDIT_FP8_A_ATTN = 1,  /* attention output        -> o_proj GEMM */
    LDIT_FP8_A_LN2 = 2,   /* layernorm_after output  -> fc1 GEMM */
    LDIT_FP8_A_GELU = 3,  /* gelu(fc1) output        -> fc2 GEMM */
    LDIT_FP8_A_COUNT = 4
};

/* epilogues of ldit_linear_f32 (what is fused behind the matmul) */
enum ldit_epilogue {
    LDIT_EPI_BIAS = 0,       /* Y = X W^T + b                         nn.Linear; TF:305-307 (q,k,v), TF:349 */
    LDIT_EPI_BIAS_GELU = 1,  /* Y = gelu_erf(X W^T + b)               TF:353-354, TF:activations.py:70-89 */
    LDIT_EPI_SCALE_RESID = 2,/* Y = R + lam (.) (X W^T + b)           TF:432-434 and TF:440-442 (LayerScale + residual) */
    /* train step, ldit_linear_bf16_ex only: */
    LDIT_EPI_F32 = 4,        /* Y fp32 = X W^T (+ b)                  dgrad into LayerNorm backward; wgrad (split-K slabs) */
    LDIT_EPI_GELU_BWD = 5    /* Y bf16 = (X W^T) (.) aux              dgrad of fc2 times the saved GELU derivative */
};

typedef void *ldit_stream;

/* Encoder geometry.  Mirrors the BeitConfig fields the path reads (TF:configuration_beit.py:72-102). */
typedef struct ldit_cfg {
    int32_t hidden;   /* C   hidden_size */
    int32_t layers;   /* L   num_hidden_layers */
    int32_t heads;    /* H   num_attention_heads ; head_dim = C / H */
    int32_t mlp;      /* F   intermediate_size */
    int32_t patch;    /* p   patch_size (square) */
    int32_t in_ch;    /* num_channels (3) */
    int32_t img_h;    /* input height, multiple of patch */
    int32_t img_w;    /* input width,  multiple of patch */
    int32_t n_taps;   /* how many hidden states the caller wants (<= LDIT_MAX_TAPS) */
    int32_t taps[LDIT_MAX_TAPS]; /* hidden-state indices, 0 = embedding output, l = after layer l
                                    (ref dit_backbone.py:33-34: d/3, d/2, 2d/3, d) */
    float ln_eps;     /* layer_norm_eps, 1e-12 for BEiT */
    int32_t dtype;    /* enum ldit_dtype */
    int32_t flags;    /* reserved, 0 */
} ldit_cfg;

/* Per-layer parameters, each exactly the tensor nn.Module.state_dict() holds (row-major [out, in] for Linear). */
typedef struct ldit_layer_weights {
    const void *ln1_w, *ln1_b;        /* layernorm_before        [C]       TF:390,426 */
    const void *wq, *bq;              /* attention q_proj        [C,C],[C] TF:305 */
    const void *wk;                   /* attention k_proj        [C,C]     TF:306 (NO bias) */
    const void *wv, *bv;              /* attention v_proj        [C,C],[C] TF:307 */
    const void *wo, *bo;              /* attention o_proj        [C,C],[C] TF:308 */
    const void *lam1;                 /* lambda_1                [C]       TF:397-403 */
    const void *ln2_w, *ln2_b;        /* layernorm_after         [C]       TF:391,438 */
    const void *w1, *b1;              /* mlp.fc1                 [F,C],[F] TF:349 */
    const void *w2, *b2;              /* mlp.fc2                 [C,F],[C] TF:350 */
    const void *lam2;                 /* lambda_2                [C] */
} ldit_layer_weights;

typedef struct ldit_weights {
    const void *patch_w;              /* embeddings.patch_embeddings.projection.weight [C,in_ch,p,p]  TF:81 */
    const void *patch_b;              /* ....projection.bias [C] */
    const void *cls;                  /* embeddings.cls_token [C]                                    TF:168 */
    const void *pos;                  /* position table for THIS input grid, [1 + (img_h/p)*(img_w/p), C]
                                         (= embeddings.position_embeddings when the grid is the table's own;
                                         otherwise the caller resamples it bicubically first, TF:113-151) */
    const ldit_layer_weights *layer;  /* HOST array of `layers` entries */
} ldit_weights;

/* ---- library ------------------------------------------------------------------------------------------------ */
int ldit_abi_version(void);
const char *ldit_last_error(void);
/* Diagnostic: the LDIT_* environment switches that force a tiling (tests reach every kernel instantiation through them; see
 * layoutdit_amd/csrc/ldit_common.h: DiagSwitches) are read ONCE, at first use; this re-reads them after the caller changed
 * one.  Not thread-safe against concurrent launches. */
int ldit_debug_reload_env(void);

/* ---- whole path: replaces `self.dit(x).hidden_states` (ref dit_backbone.py:47 ; TF:515-560) ------------------- */

/* Bytes of the packed parameter block for this geometry (one allocation the forward streams from). */
size_t ldit_packed_bytes(const ldit_cfg *cfg);

/* Gather the caller's parameter tensors into `packed` (device), fusing q/k/v into one [3C,C] matrix with bias
 * [bq ; 0 ; bv] (the key projection has no bias, TF:306).  Call again whenever parameters change. */
int ldit_pack_weights(const ldit_cfg *cfg, const ldit_weights *w, void *packed, size_t packed_bytes, ldit_stream stream);

/* LDIT_FP8 only: store the activation scales (HOST array, layers x LDIT_FP8_A_COUNT floats, all > 0) into `packed`.
 * Until this has been called after ldit_pack_weights the fp8 forward's output is undefined (scales are zero). */
int ldit_set_fp8_act_scales(const ldit_cfg *cfg, void *packed, size_t packed_bytes, const float *act_scales,
                            ldit_stream stream);

/* Scratch bytes ldit_vit_forward needs for a batch of `batch` images. */
size_t ldit_workspace_bytes(const ldit_cfg *cfg, int32_t batch);

/* x: [batch, in_ch, img_h, img_w] NCHW contiguous.  tap_out[i]: [batch, 1+P, C] row-major receives hidden state
 * cfg->taps[i] (the raw residual stream: no final LayerNorm, TF:504-506,557).  The pooler (TF:558,563-572) is not
 * computed: LayoutDiT never reads pooler_output. */
int ldit_vit_forward(const ldit_cfg *cfg, const void *packed, const void *x, int32_t batch, void *const *tap_out,
                     void *workspace, size_t workspace_bytes, ldit_stream stream);

/* The same forward fed by the detector's image list BEFORE its input transform (ref src/layoutdit/modeling/model.py:50-54:
 * GeneralizedRCNNTransform, fixed_size = (img_h, img_w) of cfg, image_mean = image_std = 0.5): `images` / `heights` / `widths` are
 * HOST arrays of `batch` entries (device pointers to [in_ch, h_i, w_i] planar images in [0, 1]; half_in != 0: fp16).  In the bf16 /
 * fp8 / split-fp32 builds the pass that writes the patch-embedding operand evaluates (bilinear(img) - mean) / std itself - no fp32
 * batch is materialised; the fp32 build, whose GEMM gathers pixels by LDS-DMA, produces the batch in its workspace first.  Either
 * way the taps EQUAL ldit_preprocess_* followed by ldit_vit_forward (one shared statement, csrc/image_blend.h). */
int ldit_vit_forward_images(const ldit_cfg *cfg, const void *packed, const void *const *images, const int32_t *heights,
                            const int32_t *widths, int32_t half_in, float mean, float std, int32_t batch, void *const *tap_out,
                            void *workspace, size_t workspace_bytes, ldit_stream stream);

/* Same, but brackets every kernel launch with HIP events on `stream`, synchronises, and ADDS the elapsed
 * milliseconds / launch counts per kernel family into ms[LDIT_K_COUNT] / launches[LDIT_K_COUNT].
 * Measurement aid for bench.py's roofline block; not for production use (it blocks the host). */
enum ldit_kernel_family {
    LDIT_K_GEMM = 0,      /* all MFMA GEMMs: patch-embed, qkv, o_proj, fc1, fc2 */
    LDIT_K_ATTENTION = 1,
    LDIT_K_LAYERNORM = 2,
    LDIT_K_OTHER = 3,
    LDIT_K_COUNT = 4
};
int ldit_vit_forward_timed(const ldit_cfg *cfg, const void *packed, const void *x, int32_t batch, void *const *tap_out,
                           void *workspace, size_t workspace_bytes, ldit_stream stream, double *ms, int64_t *launches);

/* ---- the kernels, one entry point each (unit parity tests; also usable on their own) -------------------------- */

/* Y[M,N] = epilogue(X[M,K] . W[N,K]^T).  fp32 MFMA.  K % 32 == 0, lda % 4 == 0.
 * bias[N] may be NULL (= 0).  lam[N], R[M,N] (row stride ldy) only for LDIT_EPI_SCALE_RESID; R may alias Y.
 * Y2 (optional, same shape/stride as Y) receives a second copy of the result (hidden-state tap). */
int ldit_linear_f32(const void *X, int64_t lda, const void *W, const void *bias, void *Y, int64_t ldy, int64_t M,
                    int64_t N, int64_t K, int32_t epilogue, const void *lam, const void *R, void *Y2,
                    ldit_stream stream);

/* Row LayerNorm over the last axis, biased variance about the mean, y = (x-mu) rsqrt(var+eps) g + b.
 * C % 4 == 0, C <= 4096.  (nn.LayerNorm; TF:390-391,426,438) */
int ldit_layernorm_f32(const void *X, const void *gamma, const void *beta, void *Y, int64_t rows, int64_t C, float eps,
                       ldit_stream stream);

/* softmax(Q K^T * scale) V per head, no mask.  Q,K,V,O: [B, N, H*D] token-major with row strides ldq..ldo (floats);
 * head h occupies columns [h*D, (h+1)*D).  D == 64.  (TF:268-293, TF:323-338) */
int ldit_attention_f32(const void *Q, const void *K, const void *V, void *O, int64_t B, int64_t N, int64_t H,
                       int64_t D, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float scale, ldit_stream stream);

/* Patch embedding + [cls ; patches] + position table (TF:81-90, TF:153-176).
 * x [B,in_ch,img_h,img_w] NCHW -> out [B, 1+P, C].  (in_ch*p*p) % 32 == 0, p % 4 == 0, img_w % 4 == 0. */
int ldit_embed_f32(const void *x, const void *patch_w, const void *patch_b, const void *cls, const void *pos, void *out,
                   int64_t B, int64_t in_ch, int64_t img_h, int64_t img_w, int64_t p, int64_t C, ldit_stream stream);

/* DiTBackbone tap post-processing (ref dit_backbone.py:50-61): drop CLS, view tokens as a [C,Gh,Gw] map, bilinear
 * rescale by `scale` in {4, 2, 1, 0.5} (align_corners=False).  tap [B,1+Gh*Gw,C] -> out [B,C,Gh*scale,Gw*scale] NCHW. */
int ldit_tap_to_map_f32(const void *tap, void *out, int64_t B, int64_t Gh, int64_t Gw, int64_t C, float scale,
                        ldit_stream stream);

/* Adjoint of ldit_tap_to_map_f32 (training through DiTBackbone.forward, ref dit_backbone.py:50-61 under loss.backward()):
 * dmap [B,C,Gh*scale,Gw*scale] NCHW contiguous -> dtap [B,1+Gh*Gw,C] (CLS row = 0).  Gather form, no atomics. */
int ldit_tap_to_map_bwd_f32(const void *dmap, void *dtap, int64_t B, int64_t Gh, int64_t Gw, int64_t C, float scale,
                            ldit_stream stream);

/* ---- bf16 path (first build; BASELINE configs 3-5) ------------------------------------------------------------------
 * Y[M,N] = epilogue(X[M,K] . W[N,K]^T), X and W bf16 (K-contiguous), fp32 accumulation on v_mfma_f32_32x32x16_bf16.
 * K % 64 == 0, lda % 8 == 0.  bias / lam fp32.  LDIT_EPI_BIAS and LDIT_EPI_BIAS_GELU write bf16 Y;
 * LDIT_EPI_SCALE_RESID reads the fp32 residual R (may alias Y), writes fp32 Y and the optional fp32 copy Y2. */
int ldit_linear_bf16(const void *X, int64_t lda, const void *W, const void *bias, void *Y, int64_t ldy, int64_t M,
                     int64_t N, int64_t K, int32_t epilogue, const void *lam, const void *R, void *Y2,
                     ldit_stream stream);

/* softmax(Q K^T * scale) V per head; Q, K, V bf16 [B, N, H*D] token-major (row strides in bf16 elements), O bf16;
 * fp32 softmax and accumulation.  D == 64.  scale == 0 means "Q is already multiplied by scale * log2(e)" (what the packed
 * inference path delivers: ldit_pack_weights folds that factor into W_q / b_q of the bf16 and fp8 builds): the scores are then
 * exp2-domain exponents and the kernel subtracts the running maximum on the matrix pipe instead of per element. */
int ldit_attention_bf16(const void *Q, const void *K, const void *V, void *O, int64_t B, int64_t N, int64_t H, int64_t D,
                        int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float scale, ldit_stream stream);

/* dst[i] = bf16(src[i]) (round to nearest even), n elements. */
int ldit_cast_f32_bf16(const void *src, void *dst, int64_t n, ldit_stream stream);

/* ---- split-fp32 building blocks (LDIT_F32X3 / LDIT_F32X6) ------------------------------------------------------------------
 * ldit_split_f32_planes: dst bf16 [rows, planes * cols] = the bf16 planes p0 | p1 (| p2) of src fp32 [rows, cols] (row stride lds),
 *   p0 = bf16(x), p1 = bf16(x - p0), p2 = bf16(x - p0 - p1); planes = 2 or 3; cols % 4 == 0.
 * ldit_layernorm_f32_planes: ldit_layernorm_f32 with the result written as such planes, Y bf16 [rows, planes * C].
 * ldit_linear_planes: Y = epilogue(X . W^T) for fp32 X [M,K] and W [N,K] given as planes Xp [M, planes * K] (row stride lda), Wp [N,
 *   planes * K]: the sum of 3 (planes = 2: x1 w0 + x0 w1 + x0 w0) or 6 (planes = 3: + x2 w0 + x1 w1 + x0 w2) plane products on
 *   v_mfma_f32_32x32x16_bf16 - per 64-deep k-tile, smallest first -, ONE fp32 accumulation chain per output element.  K % 64 == 0.  LDIT_EPI_BIAS and
 *   LDIT_EPI_SCALE_RESID write fp32 Y (row stride ldy); LDIT_EPI_BIAS_GELU (exact erf-GELU) writes Y as planes, bf16 [M, planes * N]. */
/* ldit_attention_planes: softmax(Q K^T) V per head on plane operands (the attention of LDIT_F32X3 / LDIT_F32X6; planes = 2 / 3).  Q, K, V
 *   point at plane 0 of their column slices of a bf16 matrix with row stride ld_in; plane s of a row lies s * plane_in elements
 *   further.  Q must already be multiplied by scale * log2(e) (ldit_pack_weights folds it into W_q / b_q of those builds).  Both
 *   products are the 3 / 6 plane products of ldit_linear_planes on the bf16 MFMA, softmax fp32.  O: bf16 [B*N, ldo], plane s of the
 *   fp32 result at column s * H * D.  D == 64. */
int ldit_attention_planes(const void *Q, const void *K, const void *V, void *O, int64_t B, int64_t N, int64_t H, int64_t D,
                          int64_t ld_in, int64_t plane_in, int64_t ldo, int32_t planes, ldit_stream stream);
int ldit_split_f32_planes(const void *src, int64_t lds, void *dst, int64_t rows, int64_t cols, int32_t planes, ldit_stream stream);
int ldit_layernorm_f32_planes(const void *X, const void *gamma, const void *beta, void *Y, int64_t rows, int64_t C, float eps,
                              int32_t planes, ldit_stream stream);
int ldit_linear_planes(const void *Xp, int64_t lda, const void *Wp, const void *bias, void *Y, int64_t ldy, int64_t M, int64_t N,
                       int64_t K, int32_t epilogue, const void *lam, const void *R, void *Y2, int32_t planes, ldit_stream stream);

/* ldit_embed_f32's arithmetic on bf16 MFMA operands - the patch embedding of the bf16 / fp8 builds and of the train step
 * (TF:81-90, TF:153-176): x fp32 NCHW is rounded to a bf16 im2col matrix [B*P, in_ch*p*p] in `scratch` (that many bf16, 16-byte
 * aligned; one HBM-bound pass), multiplied with patch_w_bf16 [C, in_ch*p*p] (fp32 accumulation), bias and position rows added in
 * fp32, written to token rows 1.. of out fp32 [B, 1+P, C]; row 0 = cls + pos[0].  (in_ch*p*p) % 64 == 0, C % 4 == 0. */
int ldit_embed_bf16(const void *x, const void *patch_w_bf16, const void *patch_b, const void *cls, const void *pos, void *out,
                    void *scratch, int64_t B, int64_t in_ch, int64_t img_h, int64_t img_w, int64_t p, int64_t C, ldit_stream stream);

/* The same embedding fed by the detector's ragged image list instead of the resized batch (the input transform
 * of ref src/layoutdit/modeling/model.py:50-54 fused into the patch-embed load): `images` / `heights` / `widths` as for
 * ldit_preprocess_f32 (half_in != 0: fp16 images), img_h x img_w = the transform's fixed_size.  The pass that builds the bf16
 * im2col matrix evaluates (bilinear(img) - mean) / std itself - the statement of ldit_preprocess_f32, bit for bit - so the
 * result EQUALS ldit_preprocess_* followed by ldit_embed_bf16, without the fp32 batch in between (one launch and a write + read
 * of B * in_ch * img_h * img_w * 4 bytes less). */
int ldit_embed_bf16_images(const void *const *images, const int32_t *heights, const int32_t *widths, int32_t half_in, float mean,
                           float std, const void *patch_w_bf16, const void *patch_b, const void *cls, const void *pos, void *out,
                           void *scratch, int64_t B, int64_t in_ch, int64_t img_h, int64_t img_w, int64_t p, int64_t C, ldit_stream stream);

/* ---- fp8 (OCP e4m3) building blocks of the fp8 build (BASELINE.json configs[4]) -----------------------------------------
 * Symmetric scaling: activations per tensor (T ~= scale_T * q, scale_T = amax(T) / 448), weights per output channel
 * (W[n,:] ~= w_scales[n] * q[n,:]).
 *
 * ldit_linear_fp8:  Y = epilogue(ab_scale * w_scales[n] * (X8 . W8^T) + bias)  with X8 [M,K] (row stride lda BYTES =
 * elements) and W8 [N,K] fp8 e4m3, fp32 accumulation (v_mfma_f32_32x32x64_f8f6f4).  w_scales: device fp32 [N] or NULL
 * (then ab_scale = scale_X * scale_W, per tensor).  K % 128 == 0, lda % 16 == 0.  bias / lam fp32.  LDIT_EPI_BIAS writes bf16 Y; LDIT_EPI_BIAS_GELU writes fp8 Y =
 * sat(gelu(.) * out_inv_scale) (ldy in elements); LDIT_EPI_SCALE_RESID reads the fp32 residual R (may alias Y),
 * writes fp32 Y and the optional fp32 copy Y2. */
int ldit_linear_fp8(const void *X, int64_t lda, const void *W, const void *bias, void *Y, int64_t ldy, int64_t M, int64_t N,
                    int64_t K, int32_t epilogue, const void *lam, const void *R, void *Y2, float ab_scale,
                    float out_inv_scale, const void *w_scales, ldit_stream stream);

/* Per-output-channel weight quantisation: scales[n] = max|W[n,:]| / 448 (never 0), codes[n,:] = fp8(W[n,:] / scales[n]).
 * W fp32 [N,K] row-major, K % 4 == 0. */
int ldit_quant_rows_f32_fp8(const void *W, void *codes, void *scales, int64_t N, int64_t K, ldit_stream stream);

/* dst[i] = fp8_e4m3(src[i] * inv_scale), round to nearest even, SATURATING at +-448 (torch's cast yields NaN above
 * 464 instead); n elements, src 16-byte aligned, dst 4-byte aligned. */
int ldit_quant_f32_fp8(const void *src, void *dst, int64_t n, float inv_scale, ldit_stream stream);

/* *out (one device float) = max |src[i]|, i < n (0 for n == 0).  Enqueues a 4-byte memset + one kernel. */
int ldit_amax_f32(const void *src, int64_t n, void *out, ldit_stream stream);

/* Detector input transform (the step that produces `x`; ref src/layoutdit/modeling/model.py:50-54 configures
 * torchvision's GeneralizedRCNNTransform with fixed_size = (224, 224), image_mean = image_std = 0.5): for each image
 * [in_ch, h_i, w_i] in [0,1]:  (img - mean) / std, then bilinear resize (align_corners = False, no antialias) to
 * out_h x out_w, written as row i of the NCHW batch `out` [B, in_ch, out_h, out_w].  `images`, `heights`, `widths`
 * are HOST arrays of B entries (device pointers / sizes); at most 65535 images per call. */
int ldit_preprocess_f32(const void *const *images, const int32_t *heights, const int32_t *widths, int32_t B, int32_t in_ch,
                        float mean, float std, int32_t out_h, int32_t out_w, void *out, ldit_stream stream);

/* Same with fp16 images (the reference's trainer casts images with .half() before the detector's transform,
 * ref src/layoutdit/training/trainer.py:153-155); arithmetic and output fp32. */
int ldit_preprocess_f16(const void *const *images, const int32_t *heights, const int32_t *widths, int32_t B, int32_t in_ch,
                        float mean, float std, int32_t out_h, int32_t out_w, void *out, ldit_stream stream);

/* fp16 <-> fp32 conversion of a pixel batch / of the returned taps for an fp16 caller (round to nearest even), n elements;
 * the fp16 side 8-byte aligned, the fp32 side 16-byte aligned. */
int ldit_cast_f16_f32(const void *src, void *dst, int64_t n, ldit_stream stream);
int ldit_cast_f32_f16(const void *src, void *dst, int64_t n, ldit_stream stream);

/* ---- FPN on top of the taps (ref src/layoutdit/modeling/dit_backbone.py:65-90: torchvision FeaturePyramidNetwork([C]*4,
 * 256, extra_blocks=LastLevelMaxPool()); torchvision's source is not available offline -> parity unpinned, checked against a
 * torch restatement of its documented forward).  The 1x1 lateral convolutions run as ldit_linear_f32 on the TOKENS of each
 * tap (a 1x1 convolution commutes with the bilinear rescale of dit_backbone.py:55-59), then per level:
 *   ldit_fpn_merge_f32:    inner [B, Gh*s, Gw*s, Ch] (NHWC) = bilinear_s(lat tokens [B, 1+Gh*Gw, Ch]) + nearest(top), top NHWC
 *                          [B, top_h, top_w, Ch] or NULL (coarsest level);  s in {4, 2, 1, 0.5};  Ch % 4 == 0
 *   ldit_conv3x3_nhwc_f32: y [B, H, W, Cout] (NHWC) = conv3x3(x [B, H, W, Cin] NHWC, padding 1) + bias as an implicit-im2col
 *                          fp32 MFMA GEMM; w [Cout, 3, 3, Cin] (= torch weight.permute(0, 2, 3, 1)); Cin % 32 == 0;
 *                          zeros: >= 128 bytes of device zeros (the padding taps' DMA source).
 * LastLevelMaxPool (kernel 1, stride 2) is a strided view of the last output, no kernel. */
int ldit_fpn_merge_f32(const void *lat, const void *top, void *out, int64_t B, int64_t Gh, int64_t Gw, int64_t Ch, float scale,
                       int64_t top_h, int64_t top_w, ldit_stream stream);
int ldit_conv3x3_nhwc_f32(const void *x, const void *w, const void *bias, void *y, int64_t B, int64_t H, int64_t W, int64_t Cin,
                          int64_t Cout, const void *zeros, ldit_stream stream);

/* ---- FPN backward (training through `self.fpn(feats)`, ref dit_backbone.py:87-90 under ref trainer.py:169-178).  The MFMA work
 * reuses entry points above and below (layoutdit_amd/modeling/dit_fpn.py sequences them): dgrad of a 3x3 convolution =
 * ldit_conv3x3_nhwc_f32 on the flipped / in-out-swapped weight; wgrad of a 3x3 convolution = nine ldit_linear_bf16_tr (wgrad
 * form) on zero-padded bf16 NHWC copies, one per tap (a tap is a constant row offset of the padded input); laterals =
 * ldit_linear_f32 / ldit_linear_bf16_tr.  New here:
 *   ldit_fpn_merge_bwd_f32:  adjoint of ldit_fpn_merge_f32.  d_inner [B, Gh*s, Gw*s, Ch] NHWC ->
 *                            d_lat [B, 1+Gh*Gw, Ch] (written; CLS row = 0; may be NULL) and d_top [B, top_h, top_w, Ch]
 *                            (ACCUMULATED into: it already holds the coarser level's own 3x3 dgrad; may be NULL).  Gather form.
 *   ldit_pad_nhwc_f32_bf16:  dst bf16 [B, H+2, W+2, C] = zero-padded copy of src fp32 [B, H, W, C] (every element written)
 *   ldit_colsum_f32:         out[n] = sum_m x[m * ldx + n] (bias gradients), two stages in a fixed order;
 *                            scratch: ldit_colsum_scratch_bytes(M, N) */
int ldit_fpn_merge_bwd_f32(const void *d_inner, void *d_lat, void *d_top, int64_t B, int64_t Gh, int64_t Gw, int64_t Ch, float scale,
                           int64_t top_h, int64_t top_w, ldit_stream stream);
int ldit_pad_nhwc_f32_bf16(const void *src, void *dst, int64_t B, int64_t H, int64_t W, int64_t C, ldit_stream stream);
size_t ldit_colsum_scratch_bytes(int64_t M, int64_t N);
int ldit_colsum_f32(const void *x, int64_t M, int64_t N, int64_t ldx, void *out, void *scratch, size_t scratch_bytes, ldit_stream stream);
/* out[n] = max_m |x[m * ldx + n]|: per-channel activation ranges for the fp8 build's calibration (the SmoothQuant-style fold of
 * DiTEncoder.calibrate_fp8: per-channel factors move LayerNorm-output outliers into the next GEMM's weight columns). Same scratch. */
int ldit_colamax_f32(const void *x, int64_t M, int64_t N, int64_t ldx, void *out, void *scratch, size_t scratch_bytes, ldit_stream stream);

/* ==== train step (BASELINE.json configs[2]: ViT-B/16 bs=64 bf16 forward + backward + AdamW) =============
 * Replaces, for the encoder, what the reference's loop runs through torch.autograd and torch.optim:
 *     loss_dict = self.model(images, targets) ; loss.backward() ; optimizer.step()     ref trainer.py:169-180
 *     optimizer = AdamW(params, lr = 1e-4, weight_decay = 0)                           ref trainer.py:62-68
 * bf16 build only (cfg.dtype = LDIT_BF16; BASELINE asks bf16 where the reference uses fp16 autocast + GradScaler).
 *
 * Parameters, gradients and the two AdamW moments are FLAT fp32 device blocks of ldit_flat_param_bytes(cfg) bytes, laid out
 * like the fp32 packed block: patch_w, patch_b, cls, pos, then per layer ln1_w, ln1_b, wqkv [3C,C] = [Wq;Wk;Wv],
 * bqkv [3C] = [bq;0;bv] (no key bias, TF:306: that third stays zero and receives a zero gradient), wo, bo, lam1, ln2_w,
 * ln2_b, w1, b1, w2, b2, lam2.  ldit_flat_param_layout writes the 4 + 14 L + 1 float offsets in that order (last = total).
 * The `pos` slot is the position table FOR THE INPUT GRID of cfg ([1 + P, C]); its gradient is with respect to that table (a caller
 * that resamples embeddings.position_embeddings bicubically chains the resample's own adjoint behind it, as layoutdit_amd.training does). */
size_t ldit_flat_param_bytes(const ldit_cfg *cfg);
int ldit_flat_param_layout(const ldit_cfg *cfg, int64_t *offsets, int32_t n);

/* bytes of: the activations kept between forward and backward / the backward's scratch */
size_t ldit_train_saved_bytes(const ldit_cfg *cfg, int32_t batch);
size_t ldit_train_workspace_bytes(const ldit_cfg *cfg, int32_t batch);

/* The bf16 MIRROR of the flat parameter block: element i = bf16(flat[i]), ldit_train_mirror_bytes(cfg) bytes.  It is the
 * `packed` argument of the two train-step entry points below: the forward reads a matrix K-contiguous at half its fp32
 * offset, the dgrad GEMMs read the SAME copy reduction-major through transposing LDS reads (no transposed copy exists),
 * the wgrad GEMMs need no weight; fp32 vectors are read from the flat block itself.  ldit_pack_train rebuilds it from
 * the flat block in one pass (after loading weights, or after a foreign optimizer changed them); ldit_adamw_step keeps it
 * current by itself when handed the mirror. */
size_t ldit_train_mirror_bytes(const ldit_cfg *cfg);
int ldit_pack_train(const ldit_cfg *cfg, const void *flat_params, void *mirror, size_t mirror_bytes, ldit_stream stream);

/* Training forward: as ldit_vit_forward, and keeps in `saved` what the backward needs (LayerNorm inputs and outputs, q|k|v,
 * the attention output and its log-sum-exp, the pre-LayerScale branch outputs, the MLP hidden after GELU and the GELU derivative at its pre-activation).
 * drop_scales: device fp32 [layers][2][batch] or NULL - stochastic depth (TF:360-378,432-434,440-442): the factor the
 * residual branch (0 = attention, 1 = MLP) of a layer is multiplied with for each sample, 0 or 1 / keep_prob; drawing
 * them is the caller's job.  ms / launches: as ldit_vit_forward_timed (both NULL = plain enqueue, capturable). */
int ldit_vit_forward_train(const ldit_cfg *cfg, const void *packed, const void *flat_params, const void *x, int32_t batch,
                           void *const *tap_out, const void *drop_scales, void *saved, size_t saved_bytes, ldit_stream stream,
                           double *ms, int64_t *launches);

/* Backward through stages stage_hi .. stage_lo (stage l >= 1 = encoder layer l, stage 0 = embeddings); a full backward is
 * (layers, 0).  Splitting it into consecutive descending ranges lets the caller start the gradient all-reduce of finished
 * layers while earlier ones are still being differentiated; the running gradient of the residual stream lives in
 * `workspace` between calls, so the ranges of one backward pass run in descending order on ONE workspace and every call gets the
 * same dtaps array (the gradient arriving at hidden state s < layers is summed in by the call that finishes stage s + 1, in the same
 * pass over the rows as that layer's LayerNorm backward).  dtaps[i] (device fp32 [batch, 1+P, C] or NULL) = gradient of the loss with
 * respect to hidden state cfg->taps[i].  grads: flat fp32 block, OVERWRITTEN (not accumulated) for every parameter of the stages processed.
 * drop_scales: the pointer given to the forward (NULL there = NULL here).  x: the forward's input (patch-embedding wgrad). */
int ldit_vit_backward(const ldit_cfg *cfg, const void *flat_params, const void *packed, const void *x, int32_t batch, void *const *dtaps,
                      const void *drop_scales, const void *saved, size_t saved_bytes, void *grads, size_t grads_bytes,
                      void *workspace, size_t workspace_bytes, int32_t stage_hi, int32_t stage_lo, ldit_stream stream,
                      double *ms, int64_t *launches);

/* Fused AdamW over n fp32 elements (torch.optim.AdamW semantics, decoupled weight decay), step counts from 1:
 *   g = grads * grad_scale ; p *= 1 - lr wd ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
 *   p -= lr / (1 - b1^step) * m / (sqrt(v) / sqrt(1 - b2^step) + eps).   n % 4 == 0, all pointers 16-byte aligned.
 * bf16_mirror (optional, n bf16 elements): receives bf16(p) of the updated parameters in the same pass. */
int ldit_adamw_step(void *params, const void *grads, void *exp_avg, void *exp_avg_sq, int64_t n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, int32_t step, float grad_scale, void *bf16_mirror,
                    ldit_stream stream);

/* ---- the kernels of the backward, one entry point each (unit parity tests) ---- */
/* ldit_attention_bf16 that also writes lse[b][h][q] = log2 sum_k exp2(scale log2(e) q.k)  (fp32 [B, H, N]) */
int ldit_attention_fwd_lse_bf16(const void *Q, const void *K, const void *V, void *O, void *lse, int64_t B, int64_t N, int64_t H,
                                int64_t D, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float scale, ldit_stream stream);
/* dQ, dK, dV (bf16, row stride lddqkv) from Q, K, V (row stride ldqkv), O, dO and the forward's lse.  D == 64.  N <= 256: one
 * LDS-resident workgroup per (image, head); longer sequences: blocks of 256 rows, one launch per role (dK/dV, dQ). */
int ldit_attention_bwd_bf16(const void *Q, const void *K, const void *V, const void *O, const void *dO, const void *lse, void *dQ,
                            void *dK, void *dV, int64_t B, int64_t N, int64_t H, int64_t D, int64_t ldqkv, int64_t ldo, int64_t lddo,
                            int64_t lddqkv, float scale, ldit_stream stream);
/* LayerNorm backward: dh += d/dx ; dgamma = sum_rows dy xhat ; dbeta = sum_rows dy.  scratch: ldit_layernorm_bwd_scratch_bytes. */
size_t ldit_layernorm_bwd_scratch_bytes(int64_t rows, int64_t C);
int ldit_layernorm_bwd_f32(const void *dy, const void *x, const void *gamma, void *dh, int64_t rows, int64_t C, float eps,
                           void *dgamma, void *dbeta, void *scratch, size_t scratch_bytes, ldit_stream stream);
/* ldit_linear_bf16 with the train step's extras: Ypre (bf16 [M,N], stride ldy) receives X W^T + b before GELU / LayerScale;
 * (LDIT_EPI_BIAS_GELU: gelu'(X W^T + b) instead); rowscale (fp32 [M]) multiplies lam per row; aux (bf16 [M,N], stride
 * ldaux) is the factor LDIT_EPI_GELU_BWD multiplies the accumulator with (= that saved gelu');
 * splits > 1 (LDIT_EPI_F32 only): K is cut into `splits` ranges, range s writes its own fp32 slab Y + s M ldy, to be summed
 * with ldit_reduce_slabs_f32. */
int ldit_linear_bf16_ex(const void *X, int64_t lda, const void *W, const void *bias, void *Y, int64_t ldy, int64_t M, int64_t N,
                        int64_t K, int32_t epilogue, const void *lam, const void *R, void *Y2, void *Ypre, const void *rowscale,
                        const void *aux, int64_t ldaux, int32_t splits, ldit_stream stream);
int ldit_reduce_slabs_f32(const void *slabs, void *out, int64_t n, int32_t count, ldit_stream stream);
/* The backward's two GEMM forms on reduction-major operands (gemm_bf16_tr.hip; operands gathered by ds_read_b64_tr_b16):
 *   a_reduction_major = 0 (dgrad): Y[M,N] = epi( A[M,K] . W[K,N] ), A K-contiguous (K % 64 == 0), W row stride ldw;
 *                                  epilogues LDIT_EPI_F32, LDIT_EPI_BIAS (bf16 out), LDIT_EPI_GELU_BWD (x aux, stride N)
 *   a_reduction_major = 1 (wgrad): Y[M,N] = A[K,M]^T . W[K,N], any K (rows past K read zeros), fp32 out, optional split-K
 * zeros: >= 128 bytes of device zeros.  M, N, lda, ldw multiples of 8. */
int ldit_linear_bf16_tr(const void *A, int64_t lda, int32_t a_reduction_major, const void *W, int64_t ldw, void *Y, int64_t ldy,
                        int64_t M, int64_t N, int64_t K, int32_t epilogue, const void *aux, int32_t splits, const void *zeros,
                        ldit_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* LDIT_H */
