// Shared internals of libldit_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "ldit.h"

namespace ldit {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// thread-local error text behind ldit_last_error()
char *err_buf();
int fail(int code, const char *fmt, ...);

#define LDIT_HIP_CHECK(expr)                                                                              \
    do {                                                                                                  \
        hipError_t e__ = (expr);                                                                          \
        if (e__ != hipSuccess) return ::ldit::fail(LDIT_EHIP, "%s: %s", #expr, hipGetErrorString(e__));   \
    } while (0)

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, DEVICE): a process that drives several GPUs must set it
// on each of them before the first launch there with more than 64 KB of dynamic LDS.  `done` (one per kernel instantiation,
// a function-local static of its launcher) holds one bit per device ordinal; ordinals >= 64 set the attribute every time.
// The attribute is idempotent, so two threads racing on the same bit at worst both set it.  This is the library's only
// mutable process state besides the diagnostic switches below.
int ensure_dynamic_lds(const void *kern, int bytes, std::atomic<unsigned long long> &done);
#define LDIT_DYN_LDS(kern, bytes)                                                                               \
    do {                                                                                                        \
        static std::atomic<unsigned long long> done__{0};                                                       \
        if (int rc__ = ::ldit::ensure_dynamic_lds(reinterpret_cast<const void *>(kern), (bytes), done__)) return rc__; \
    } while (0)

// Compute units of the current device (hipDeviceAttributeMultiprocessorCount; 256 on MI355X), cached per device ordinal: the
// persistent GEMM launches size their grid with it.  Read-mostly process state like the dynamic-LDS bits above.
int compute_units();

// Diagnostic switches (tests and experiments force every tiling through them; none is needed in production).  Read from the
// environment ONCE, when the library is first used - not per launch (a getenv on the 12 us-per-launch serving path) - and
// again only by ldit_debug_reload_env() (tests call it after changing a variable).
struct DiagSwitches {
    int gemm_tile = -1;          // LDIT_GEMM_TILE 0..7 (fp32 GEMM tiling)
    long thin_tiles = 192;       // LDIT_GEMM_THIN_TILES
    bool panel_r16_vec = false;  // LDIT_PANEL_R16=vec
    bool panel_persist = false;  // LDIT_GEMM_PERSIST=1: persistent tile loop of the fp32 panel GEMM (measured equal: off by default)
    int bf16_tile = -1;          // LDIT_GEMM_BF16_TILE 2..5 (-1 = picker; any value also disables the small-M kernel)
    bool bf16_tile_env = false;  //   the variable is present at all
    int bf16_tr_tile = -1;       // LDIT_GEMM_BF16_TR_TILE
    bool bf16_tail_launch = false;   // LDIT_GEMM_BF16_TAIL_LAUNCH=1: the peeled tail of a bf16 GEMM as its own launch (round 3) instead of riding in the main one
    int fp8_tile = -1;           // LDIT_GEMM_FP8_TILE 0..4
    bool fp8_k16 = false, fp8_noskinny = false;   // LDIT_GEMM_FP8_K16, LDIT_GEMM_FP8_NOSKINNY
    bool direct_epi = false;     // LDIT_GEMM_DIRECT_EPILOGUE=1
    bool attn_bf16_kt4 = false;  // LDIT_ATTN_BF16_KT=4
    int attn_bf16_nw = -1;       // LDIT_ATTN_BF16_NW=8: eight query tiles per workgroup (one staging of a chunk per 256 queries)
    long planes_tail_waves = 700;    // LDIT_PLANES_TAIL_WAVES: split-fp32 GEMMs of at most this many 32 x 32 output tiles run on the one-wave-per-tile kernel
    int seg_order = -1;          // LDIT_GEMM_SEG_ORDER 0 = plane segments outermost (whole K per segment), 1 = innermost (per k-tile)
};
const DiagSwitches &diag();
void reload_diag();

// erf to < 1 ulp in ~25 instructions, branch-free (both ranges evaluated, one selected): odd polynomial in x below
// 0.927734375, 1 - exp(poly(|x|)) above (coefficients: N. Juffa's single-precision erff).  libdevice's erff costs
// ~90 instructions per call with divergent ranges - 28 us per fc1 tile round when it sat in the GEMM epilogue.
__device__ __forceinline__ float erf_fast(float a)
{
    const float t = fabsf(a), s = a * a;
    float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = fmaf(r, s, u);
    r = fmaf(r, t, -1.06777877e-1f);
    r = fmaf(r, t, -6.34846687e-1f);
    r = fmaf(r, t, -1.28717512e-1f);
    r = fmaf(r, t, -t);
    const float big = copysignf(1.0f - __expf(r), a);
    float q = -5.96761703e-4f;
    q = fmaf(q, s, 4.99119423e-3f);
    q = fmaf(q, s, -2.67681349e-2f);
    q = fmaf(q, s, 1.12819925e-1f);
    q = fmaf(q, s, -3.76125336e-1f);
    q = fmaf(q, s, 1.28379166e-1f);
    const float small = fmaf(q, a, a);
    return t > 0.927734375f ? big : small;
}

// exact (erf) GELU of TF:activations.py:70-89
__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erf_fast(v * 0.70710678118654752440f)); }

// GELU of the bf16 / fp8 builds (round 3): the erf-GELU of TF:activations.py:70-89 evaluated through its logistic form
//   gelu(v) ~= v * sigmoid(a v + b v^3) = v / (1 + exp2(-(a v + b v^3) log2 e)),   a = 2 sqrt(2/pi), b = 0.044715 a
// (the "tanh" GELU written with one exponential).  |deviation from the exact erf-GELU| <= 4.8e-4 everywhere (maximum at
// |v| ~ 2.7, where the result is 2.7 and half a bf16 ulp is 7.8e-3; in the negative tail, results of -0.17 .. -0.004, it stays
// <= 2.2e-4) - at or under the rounding the result receives right after, and far inside the builds' parity gates (bf16
// 2e-2, fp8 1e-1; the whole-path errors the tests print did not move).  5 plain VALU + v_exp_f32 + v_rcp_f32 = 36 issue cycles
// per element against 60 for the Abramowitz-Stegun erf of rounds 1-2: the fc1 epilogue is VALU-bound (17 k cycles per
// 256 x 256 tile, a third of its K = 1024 main loop): ViT-L forward 14.70 -> 14.36 ms, fp8 ViT-B bs=32 1.74 -> 1.67 ms.
// The fp32 build keeps gelu_erf (< 1 ulp).
__device__ __forceinline__ float gelu_lp(float v)
{
    const float w = v * __builtin_fmaf(v * v, -0.10294324f, -2.30220819f);      // -(a v + b v^3) log2(e)
    return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(w));
}

// The same function and ITS OWN derivative from one set of sub-expressions (train forward: the fc1 epilogue stores both, the
// backward multiplies by the saved derivative - forward and backward differentiate the function that was actually computed):
//   s = sigmoid(u), u = a v + b v^3:   gelu = v s,   gelu' = s + v s (1 - s) (a + 3 b v^2)
__device__ __forceinline__ void gelu_and_grad_lp(float v, float &gelu, float &grad)
{
    const float v2 = v * v;
    const float w = v * __builtin_fmaf(v2, -0.10294324f, -2.30220819f);
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(w));
    const float vs = v * s;
    gelu = vs;
    grad = __builtin_fmaf(vs - vs * s, __builtin_fmaf(v2, 0.21406445f, 1.59576912f), s);
}

// 4 floats -> 4 fp8 e4m3 (round to nearest even, saturating at +-448), packed little-endian in one dword
__device__ __forceinline__ unsigned pack_fp8x4(float a, float b, float c, float d)
{
    const float lim = 448.0f;
    a = fminf(fmaxf(a, -lim), lim); b = fminf(fmaxf(b, -lim), lim);
    c = fminf(fmaxf(c, -lim), lim); d = fminf(fmaxf(d, -lim), lim);
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned)w;
}

// ---- GEMM ----------------------------------------------------------------------------------------------------
enum AMode { A_ROWMAJOR = 0, A_PATCH = 1,
             A_CONV3 = 2 };   // fp32 general kernel only: implicit im2col of a 3x3 / pad 1 convolution over an NHWC map
enum Epi { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_SCALE_RESID = 2, EPI_EMBED = 3,
           EPI_F32 = 4,        // bf16 GEMM only: Y fp32 = acc (+ bias); dgrad into LayerNorm backward, wgrad slabs
           EPI_GELU_BWD = 5,   // bf16 GEMM only: Y bf16 = acc * aux, aux = gelu'(pre-activation) saved by the forward
           EPI_GELU_SPLIT = 6, // bf16 GEMM, split-fp32 build only: Y = split_bf16(gelu_erf(acc + bias)), `nsplit_out` bf16 planes per row
           EPI_BIAS_SPLIT = 7 }; // the same without the GELU (q|k|v of the split-fp32 build, consumed as planes by attention_planes)

// Optional operands of the bf16 GEMM used by the train step (training.hip); all null / 1 for inference.
struct GemmExtra {
    void *Ypre = nullptr;              // bf16 [M,N], row stride ldy, saved for backward: EPI_SCALE_RESID: acc + bias before
                                       // LayerScale; EPI_BIAS_GELU: gelu'(acc + bias), the factor of the GELU backward
    const float *rowscale = nullptr;   // fp32 [M]: per-row factor on lam (stochastic depth, TF:360-378), EPI_SCALE_RESID
    const void *aux = nullptr;         // bf16 [M,N], row stride ldaux: EPI_GELU_BWD multiplies the accumulator with it
    int ldaux = 0;
    int splits = 1;                    // EPI_F32: K split into `splits` slabs, slab s at (float*)Y + s * M * ldy
    const void *zeros = nullptr;       // gemm_bf16_tr.hip: >= 128 bytes of device zeros (source of reduction rows past the end)
    // EPI_EMBED (bf16 GEMM on the bf16 im2col of the batch, launch_patches_rows): row m = (image b, patch i) lands in token row
    // b * (patches + 1) + 1 + i of the fp32 output with pos[1 + i] added (TF:153-176; the CLS rows are launch_cls_rows')
    const float *pos = nullptr;        // fp32 [patches + 1, N], row stride ldy
    int patches = 0;
    // Split-fp32 build (LDIT_F32X3 / LDIT_F32X6, api.hip): an fp32 operand x is held as S bf16 PLANES x ~= p0 + p1 (+ p2), p0 = bf16(x),
    // p1 = bf16(x - p0), ... stored side by side per row (A: [M, S K], W: [N, S K]).  The product is the sum of `nseg` plane
    // products in a fixed order: the k-loop walks segment g = 0 .. nseg-1 over the K columns of A-plane seg_a(g) and W-plane
    // seg_w(g) (4 bits each, segment 0 in the low nibble) - ONE accumulation chain per output element whatever tile it falls in.
    // K in the launch arguments stays the per-plane depth.  nseg = 0: ordinary bf16 GEMM.
    int nseg = 0;
    unsigned seg_a = 0, seg_w = 0;
    int seg_inner = 0;                 // 1: the segments are walked per 64-deep k-tile (k-tile outermost) instead of per whole K
    int nsplit_out = 0;                // EPI_GELU_SPLIT: planes of the output row (row stride ldy = nsplit_out * N)
    // gemm_bf16_tr.hip (round 4): column sums of the bf16 OUTPUT as the epilogue writes it - the bias gradient that a separate pass
    // over the tensor used to collect.  colsum[r][n], r = (row tile index * waves along M + wave row): one fp32 partial row per
    // wave row of every tile, `colsum_rows(M, BM, WM)` of them, summed afterwards by launch_reduce_jobs; row stride = N.
    float *colsum = nullptr;
    // gemm_bf16.hip: rows M .. M + tail_rows - 1 of the same operands (the peeled ragged tail) are computed by extra workgroups of the
    // SAME launch, one 32 x 16 tile per wave (0 = no tail rides along)
    int tail_rows = 0;
};

// x ~= p[0] + p[1] (+ p[2]): the bf16 planes of the split-fp32 build.  Every subtraction is exact in fp32.
template <int S>
__device__ __forceinline__ void split_bf16_planes(float x, __bf16 (&p)[S])
{
#pragma unroll
    for (int s = 0; s < S; ++s) {
        p[s] = (__bf16)x;
        x -= (float)p[s];
    }
}

struct GemmArgs {
    const float *A;      // [M, K] row-major (A_ROWMAJOR) or the NCHW image batch (A_PATCH)
    const float *W;      // [N, K] row-major (nn.Linear.weight)
    float *Y;            // [M, N] (EPI_EMBED: token rows, see below)
    float *Y2;           // optional second copy of Y (hidden-state tap)
    const float *bias;   // [N] or null
    const float *lam;    // [N]   (EPI_SCALE_RESID)
    const float *R;      // [M,N] (EPI_SCALE_RESID), row stride ldy, may alias Y
    const float *pos;    // [tokens, N] (EPI_EMBED)
    int M, N, K;
    int lda, ldy;        // row strides in floats
    // A_PATCH / EPI_EMBED geometry: row m = (image b, patch gy*gw + gx); k = (ch, dy, dx)
    int img_h, img_w, gw, patches, patch, tokens;
    // A_CONV3: A = NHWC map [B, conv_h, conv_w, conv_c]; row m = pixel (b, y, x); k = (ky, kx, c); taps outside the map read
    // `zeros` (>= 128 bytes of device zeros: LDS-DMA has no predication, so padding is a source address)
    int conv_h, conv_w, conv_c;
    const float *zeros;
#ifdef LDIT_GEMM_STAMPS
    unsigned long long *stamps;   // diagnostic build only: 8 words per block (see scripts/gemm_stamps.py)
#endif
};

int launch_gemm(const GemmArgs &a, int epi, int amode, hipStream_t stream);
int launch_gemm_panel(const GemmArgs &a, int epi, int amode, hipStream_t stream, int height = 304);   // (32 T + 16) x 128 panels (gemm_panel_f32.hip)
int launch_gemm_thin(const GemmArgs &a, int epi, hipStream_t stream);               // serving-size M (gemm_thin_f32.hip)
bool gemm_thin_prefers(int M, int N);

// ---- other kernels ---------------------------------------------------------------------------------------------
int launch_layernorm(const float *X, const float *g, const float *b, float *Y, int64_t rows, int C, float eps,
                     hipStream_t stream);
int launch_layernorm_bf16out(const float *X, const float *g, const float *b, void *Y, int64_t rows, int C, float eps,
                             hipStream_t stream);
int launch_attention_bf16(const void *Q, const void *K, const void *V, void *O, int B, int N, int H, int D, int ldq, int ldk,
                          int ldv, int ldo, float scale, hipStream_t stream);
int launch_attention(const float *Q, const float *K, const float *V, float *O, int B, int N, int H, int D, int ldq,
                     int ldk, int ldv, int ldo, float scale, hipStream_t stream);
int launch_cls_rows(const float *cls, const float *pos, float *out, int B, int tokens, int C, hipStream_t stream);
int launch_tap_to_map(const float *tap, float *out, int B, int Gh, int Gw, int C, float scale, hipStream_t stream);
int launch_tap_to_map_bwd(const float *dmap, float *dtap, int B, int Gh, int Gw, int C, float scale, hipStream_t stream);
int launch_preprocess(const void *const *images, bool half_in, const int *heights, const int *widths, int B, int in_ch, float mean,
                      float std, int out_h, int out_w, float *out, hipStream_t stream);
struct ImageList;
int fill_image_list(ImageList &l, const void *const *images, const int *heights, const int *widths, int B, int first, float mean, float std);
// the bf16 (or plane) im2col rows of the patch embedding straight from a ragged image list: the input transform fused into the load
int launch_patches_rows_images(const void *const *images, bool half_in, const int *heights, const int *widths, int B, int in_ch,
                               float mean, float std, int out_h, int out_w, int p, void *out, hipStream_t stream, int planes = 1);
int launch_cast_f16(const void *src, void *dst, size_t n, bool widen, hipStream_t stream);
int launch_fpn_merge(const float *lat, const float *top, float *out, int B, int Gh, int Gw, int Ch, float scale, int top_h,
                     int top_w, hipStream_t stream);
// FPN backward (fpn_bwd.hip)
int launch_fpn_merge_bwd(const float *din, float *dlat, float *dtop, int B, int Gh, int Gw, int Ch, float scale, int top_h, int top_w,
                         hipStream_t stream);
int launch_pad_nhwc_bf16(const float *src, void *dst, int B, int H, int W, int C, hipStream_t stream);
size_t colsum_scratch_bytes(int64_t M, int64_t N);
int launch_colsum_f32(const float *x, int64_t M, int N, int64_t ld, float *out, float *scratch, size_t scratch_bytes, bool absmax,
                      hipStream_t stream);
int launch_gemm_bf16(const void *A, int lda, const void *W, const float *bias, void *Y, int ldy, int M, int N, int K, int epi,
                     const float *lam, const float *R, float *Y2, hipStream_t stream);
int launch_gemm_bf16_ex(const void *A, int lda, const void *W, const float *bias, void *Y, int ldy, int M, int N, int K, int epi,
                        const float *lam, const float *R, float *Y2, const GemmExtra &x, hipStream_t stream);
int launch_gemm_bf16_tr(const void *A, int lda, bool a_reduction_major, const void *W, int ldw, const float *bias, void *Y, int ldy,
                        int M, int N, int K, int epi, const GemmExtra &x, hipStream_t stream);
int gemm_bf16_tr_colsum_rows(int M, int N, int *bm = nullptr);     // partial rows written through GemmExtra::colsum
int launch_cvt_bf16(const float *src, void *dst, size_t n, hipStream_t stream, float mul = 1.0f);
// split-fp32 build: dst bf16 [rows, planes * cols] = the `planes` bf16 planes of src fp32 [rows, cols] (row stride lds) side by side
int launch_split_planes(const float *src, int lds, void *dst, int rows, int cols, int planes, hipStream_t stream, float mul = 1.0f);
// f32x3 build: attention on bf16-plane operands (attention_planes.hip); queries pre-multiplied by scale * log2(e)
int launch_attention_planes2(const void *Q, const void *K, const void *V, void *O, int B, int N, int H, int D, int ld_in, int plane_in,
                             int ldo, hipStream_t stream);
int launch_attention_planes3(const void *Q, const void *K, const void *V, void *O, int B, int N, int H, int D, int ld_in, int plane_in,
                             int ldo, hipStream_t stream);     // three planes, six products (f32x6)
int launch_layernorm_splitout(const float *X, const float *g, const float *b, void *Y, int64_t rows, int C, float eps, int planes,
                              hipStream_t stream);
// fp8: the per-tensor part of the accumulator's dequantisation is d_act[0] when d_act is non-null (device), else the host
// value ab_scale; d_wrow (device, [N], optional) multiplies per-output-channel weight scales onto it; d_out (device, 1
// float: scale of the fp8 output) overrides out_inv_scale.
int launch_gemm_fp8(const void *A, int lda, const void *W, const float *bias, void *Y, int ldy, int M, int N, int K, int epi,
                    const float *lam, const float *R, float *Y2, float ab_scale, float out_inv_scale, const float *d_act,
                    const float *d_wrow, const float *d_out, hipStream_t stream);
int launch_quant_rows_fp8(const float *W, void *dst, float *scales, int N, int K, hipStream_t stream, float mul = 1.0f);
int launch_quant_fp8(const float *src, void *dst, size_t n, float inv_scale, const float *d_scale, hipStream_t stream);
int launch_amax_f32(const float *src, size_t n, float *out, bool accumulate, hipStream_t stream);
int launch_amax_to_scale(float *p, int n, hipStream_t stream);   // p[i] = max(p[i], tiny) / 448
int launch_layernorm_fp8out(const float *X, const float *g, const float *b, void *Y, int64_t rows, int C, float eps,
                            const float *qscale, hipStream_t stream);
int launch_attention_bf16_fp8out(const void *Q, const void *K, const void *V, void *O, int B, int N, int H, int D, int ldq,
                                 int ldk, int ldv, int ldo, float scale, const float *qscale, hipStream_t stream);
int launch_pack_qkv_bias(const float *bq, const float *bv, float *dst, int C, hipStream_t stream, float qmul = 1.0f);

// ---- train step (train_ops.hip, attention_bwd_bf16.hip, api_train.hip) ----------------------------------------------------
// second stage of the column reductions / split-K slab sums: out[j][n] = sum_{p < P[j]} part[j][p * stride[j] + n]
struct ReduceJobs {
    int n = 0;
    int first_block[16];
    const float *part[16];
    float *out[16];
    int64_t N[16];
    int P[16];
    int64_t stride[16];
    bool full() const { return n == 16; }
    void add(const float *part_, float *out_, int64_t N_, int P_, int64_t stride_)
    {
        part[n] = part_; out[n] = out_; N[n] = N_; P[n] = P_; stride[n] = stride_; ++n;
    }
};
int launch_reduce_jobs(ReduceJobs &jobs, hipStream_t stream);     // resets jobs.n
int launch_attention_bf16_lse(const void *Q, const void *K, const void *V, void *O, float *lse, int B, int N, int H, int D, int ldq,
                              int ldk, int ldv, int ldo, float scale, hipStream_t stream);
int launch_attention_bwd_bf16(const void *Q, const void *K, const void *V, const void *O, const void *dO, const float *lse,
                              void *dQ, void *dK, void *dV, int B, int N, int H, int D, int ldqkv, int ldo, int lddo, int lddqkv,
                              float scale, hipStream_t stream);
int launch_transpose_bf16(const void *src, bool src_f32, void *dst, int M, int N, int ld_src, int Mp, int skip_tokens,
                          float *colsum_part, hipStream_t stream, void *rowmajor_copy = nullptr);
int launch_resid_bwd(const float *dh, const void *z, const float *lam, const float *rowscale, void *dz, void *dzT, int M, int C,
                     int Mp, float *dlam_part, float *db_part, hipStream_t stream);      // dzT may be null
int launch_colsum_bf16(const void *src, int M, int N, int ld, float *part, hipStream_t stream);   // part[ceil(M/64)][N]
int launch_rows_to_bf16(const float *src, void *dst, int M, int N, int skip_tokens, hipStream_t stream);
int layernorm_bwd_blocks(int64_t rows);
int layernorm_bwd_resid_blocks(int64_t rows);
int launch_layernorm_bwd(const float *dy, const float *x, const float *g, float *dh, int64_t rows, int C, float eps,
                         float *dg_part, float *db_part, hipStream_t stream, const float *add = nullptr);   // add: one more [rows, C] gradient summed into dh
int launch_layernorm_bwd_resid(const float *dy, const float *x, const float *g, float *dh, int64_t rows, int C, float eps,
                               float *dg_part, float *db_part, const void *z, const float *lam, const float *rowscale, void *dz,
                               float *dlam_part, float *dzb_part, hipStream_t stream);
int launch_add_inplace(float *a, const float *b, size_t n, hipStream_t stream);
int launch_expand_rowscale(const float *drop, float *rowscale, int B, int T, int nvec, hipStream_t stream);
int launch_embed_bwd_small(const float *dh0, float *dpos, float *dcls, float *dpb, int B, int T, int C, hipStream_t stream);
int launch_patches_transposed(const float *x, void *out, int B, int in_ch, int img_h, int img_w, int p, int Mp, hipStream_t stream);
int launch_patches_rows(const float *x, void *out, int B, int in_ch, int img_h, int img_w, int p, hipStream_t stream, int planes = 1);
int launch_adamw(float *p, const float *g, float *m, float *v, size_t n, float lr, float b1, float b2, float eps, float wd,
                 int step, float grad_scale, void *mirror, hipStream_t stream);

}  // namespace ldit
