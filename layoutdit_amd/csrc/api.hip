// C-ABI entry points of libldit_hip.so (declared in include/ldit.h).  Host-side sequencing only: every byte of
// arithmetic happens in the HIP kernels of this directory; there is no CPU fallback.
#include "api_internal.h"

namespace ldit {

char *err_buf()
{
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

int ensure_dynamic_lds(const void *kern, int bytes, std::atomic<unsigned long long> &done)
{
    int dev = 0;
    LDIT_HIP_CHECK(hipGetDevice(&dev));
    const unsigned long long bit = dev >= 0 && dev < 64 ? 1ull << dev : 0ull;
    if (bit && (done.load(std::memory_order_relaxed) & bit)) return LDIT_OK;
    LDIT_HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    if (bit) done.fetch_or(bit, std::memory_order_relaxed);
    return LDIT_OK;
}

int compute_units()
{
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    const bool slot = dev >= 0 && dev < 64;
    if (slot) {
        const int c = cached[dev].load(std::memory_order_relaxed);
        if (c > 0) return c;
    }
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    if (slot) cached[dev].store(n, std::memory_order_relaxed);
    return n;
}

namespace {
DiagSwitches read_switches()
{
    DiagSwitches d;
    auto digit = [](const char *name, int lo, int hi) {
        const char *e = getenv(name);
        return (e && e[0] >= '0' + lo && e[0] <= '0' + hi && e[1] == 0) ? e[0] - '0' : -1;
    };
    auto is = [](const char *name, char c) { const char *e = getenv(name); return e && e[0] == c; };
    d.gemm_tile = digit("LDIT_GEMM_TILE", 0, 7);
    if (const char *e = getenv("LDIT_GEMM_THIN_TILES")) d.thin_tiles = atol(e);
    d.panel_r16_vec = is("LDIT_PANEL_R16", 'v');
    d.panel_persist = is("LDIT_GEMM_PERSIST", '1');
    d.bf16_tile = digit("LDIT_GEMM_BF16_TILE", 2, 7);
    d.bf16_tile_env = getenv("LDIT_GEMM_BF16_TILE") != nullptr;
    d.bf16_tr_tile = digit("LDIT_GEMM_BF16_TR_TILE", 0, 9);
    d.bf16_tail_launch = is("LDIT_GEMM_BF16_TAIL_LAUNCH", '1');
    d.fp8_tile = digit("LDIT_GEMM_FP8_TILE", 0, 6);
    d.fp8_k16 = is("LDIT_GEMM_FP8_K16", '1');
    d.fp8_noskinny = is("LDIT_GEMM_FP8_NOSKINNY", '1');
    d.direct_epi = is("LDIT_GEMM_DIRECT_EPILOGUE", '1');
    d.attn_bf16_kt4 = is("LDIT_ATTN_BF16_KT", '4');
    d.seg_order = digit("LDIT_GEMM_SEG_ORDER", 0, 1);
    if (const char *e = getenv("LDIT_PLANES_TAIL_WAVES")) d.planes_tail_waves = atol(e);
    d.attn_bf16_nw = digit("LDIT_ATTN_BF16_NW", 4, 8);
    return d;
}
DiagSwitches &switches()
{
    static DiagSwitches d = read_switches();
    return d;
}
}  // namespace

const DiagSwitches &diag() { return switches(); }
void reload_diag() { switches() = read_switches(); }

namespace {

int embed(const Geo &g, const float *x, const float *pw, const float *pb, const float *cls, const float *pos, float *out,
          int batch, int img_h, int img_w, hipStream_t stream, Probe &probe)
{
    GemmArgs a{};
    a.A = x; a.W = pw; a.Y = out; a.Y2 = nullptr; a.bias = pb; a.pos = pos;
    a.M = batch * g.P; a.N = g.C; a.K = g.Kp;
    a.lda = g.in_ch * img_h * img_w;   // per-image stride in patch mode
    a.ldy = g.C;
    a.img_h = img_h; a.img_w = img_w; a.gw = g.gw; a.patches = g.P; a.patch = g.p; a.tokens = g.T;
    LDIT_RUN(probe, LDIT_K_GEMM, launch_gemm(a, EPI_EMBED, A_PATCH, stream));
    LDIT_RUN(probe, LDIT_K_OTHER, launch_cls_rows(cls, pos, out, batch, g.T, g.C, stream));
    return LDIT_OK;
}

int linear(const float *X, int lda, const float *W, const float *bias, float *Y, int ldy, int M, int N, int K, int epi,
           const float *lam, const float *R, float *Y2, hipStream_t stream, Probe &probe)
{
    GemmArgs a{};
    a.A = X; a.W = W; a.Y = Y; a.Y2 = Y2; a.bias = bias; a.lam = lam; a.R = R;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldy = ldy;
    LDIT_RUN(probe, LDIT_K_GEMM, launch_gemm(a, epi, A_ROWMAJOR, stream));
    return LDIT_OK;
}

// the pixels as the detector holds them before its input transform: a ragged list of [in_ch, h_i, w_i] images in [0, 1]
// (ldit_vit_forward_images: the transform is evaluated by the kernel that produces the patch-embedding operand)
struct ImgSrc {
    const void *const *images;
    const int32_t *heights, *widths;
    bool half_in;
    float mean, std;
};

int forward(const ldit_cfg *cfg, const void *packed, const void *x, int32_t batch, void *const *tap_out, void *workspace,
            size_t ws_bytes, hipStream_t stream, Probe &probe, const ImgSrc *imgs = nullptr)
{
    Geo g;
    LDIT_TRY(geometry(cfg, g));
    if (batch <= 0) return fail(LDIT_EINVAL, "batch %d must be positive", batch);
    if (!packed || (!x && !imgs) || !workspace) return fail(LDIT_EINVAL, "null packed / x / workspace pointer");
    if (!aligned16(packed) || (x && !aligned16(x)) || !aligned16(workspace)) return fail(LDIT_EINVAL, "pointers must be 16-byte aligned");
    if (imgs && (!imgs->images || !imgs->heights || !imgs->widths || !(imgs->std > 0.0f) || batch > 65535))
        return fail(LDIT_EINVAL, "image list: null array, non-positive std or more than 65535 images");
    if (cfg->n_taps && !tap_out) return fail(LDIT_EINVAL, "tap_out is null");
    for (int i = 0; i < cfg->n_taps; ++i)
        if (!tap_out[i] || !aligned16(tap_out[i])) return fail(LDIT_EINVAL, "tap_out[%d] is null or misaligned", i);
    if ((int64_t)batch * g.T * (int64_t)(g.F > 3 * g.C ? g.F : 3 * g.C) >= (1ll << 31))
        return fail(LDIT_EUNSUPPORTED, "batch %d: activation index space exceeds 2^31 elements, split the batch", batch);
    const Workspace wm = workspace_map(g, batch, cfg->dtype);
    if (ws_bytes < wm.total) return fail(LDIT_EWORKSPACE, "workspace %zu bytes < required %zu", ws_bytes, wm.total);
    const PackedMap pm = packed_map(g, cfg->dtype);
    const char *P = static_cast<const char *>(packed);
    auto F32 = [&](size_t off) { return reinterpret_cast<const float *>(P + off); };
    char *ws = static_cast<char *>(workspace);
    float *h = reinterpret_cast<float *>(ws + wm.h);
    float *y = reinterpret_cast<float *>(ws + wm.y);
    float *big = reinterpret_cast<float *>(ws + wm.big);
    const int M = batch * g.T, C = g.C, F = g.F;
    const size_t act_bytes = (size_t)M * C * 4;

    auto tap_for = [&](int hidden_idx) -> float * {
        for (int i = 0; i < cfg->n_taps; ++i)
            if (cfg->taps[i] == hidden_idx) return static_cast<float *>(tap_out[i]);
        return nullptr;
    };
    auto extra_taps = [&](int hidden_idx, const float *src, float *first) -> int {
        // the same hidden state requested more than once: copy to the remaining destinations
        for (int i = 0; i < cfg->n_taps; ++i)
            if (cfg->taps[i] == hidden_idx && tap_out[i] != first)
                LDIT_HIP_CHECK(hipMemcpyAsync(tap_out[i], src, act_bytes, hipMemcpyDeviceToDevice, stream));
        return LDIT_OK;
    };

    // embeddings (TF:153-176).  fp32 build: the fp32 GEMM gathers the NCHW pixels itself (LDS-DMA source addresses).  bf16 / fp8
    // builds: one HBM-bound pass rounds the batch to a bf16 im2col matrix [B P, 3 p p] (into `big`, free until layer 0) and the
    // bf16 MFMA GEMM multiplies it - 16x the fp32 matrix rate for 1.8 - 4.5 % of those steps (the fp32 kernel took 266 us of
    // the 14.6 ms ViT-L/512 step, 60 us of the 1.6 ms fp8 step).
    if (cfg->dtype != LDIT_F32 && g.Kp % 64 == 0) {
        // (split-fp32 builds: the im2col rows hold the bf16 planes of the pixels, the GEMM walks the plane products)
        const int Se = split_planes_of(cfg->dtype) ? split_planes_of(cfg->dtype) : 1;
        char *patches = ws + wm.big;
        if (imgs)       // normalise + bilinear resize of the ragged list INSIDE this pass - no fp32 batch in between
            LDIT_RUN(probe, LDIT_K_OTHER, launch_patches_rows_images(imgs->images, imgs->half_in, imgs->heights, imgs->widths, batch, g.in_ch,
                                                                    imgs->mean, imgs->std, cfg->img_h, cfg->img_w, g.p, patches, stream, Se));
        else
            LDIT_RUN(probe, LDIT_K_OTHER, launch_patches_rows(static_cast<const float *>(x), patches, batch, g.in_ch, cfg->img_h, cfg->img_w,
                                                             g.p, stream, Se));
        GemmExtra xe{};
        xe.pos = F32(pm.pos); xe.patches = g.P;
        if (Se == 2) { xe.nseg = 3; xe.seg_a = 0x001u; xe.seg_w = 0x010u; }
        if (Se == 3) { xe.nseg = 6; xe.seg_a = 0x001012u; xe.seg_w = 0x010210u; }
        xe.seg_inner = 1;
        LDIT_RUN(probe, LDIT_K_GEMM, launch_gemm_bf16_ex(patches, Se * g.Kp, P + pm.patch_w16, F32(pm.patch_b), h, C, batch * g.P, C, g.Kp,
                                                        EPI_EMBED, nullptr, nullptr, nullptr, xe, stream));
        LDIT_RUN(probe, LDIT_K_OTHER, launch_cls_rows(F32(pm.cls), F32(pm.pos), h, batch, g.T, C, stream));
    } else {
        const float *xp = static_cast<const float *>(x);
        if (imgs) {
            // the fp32 GEMM gathers the NCHW pixels on its LDS-DMA operand path, where nothing can be blended: the batch is produced
            // first (same statement, image_blend.h), into `big` - free until layer 0
            const size_t pix = (size_t)batch * g.in_ch * cfg->img_h * cfg->img_w * 4;
            if (wm.total - wm.big < pix) return fail(LDIT_EUNSUPPORTED, "image list: the pixel batch does not fit the workspace of this geometry");
            LDIT_RUN(probe, LDIT_K_OTHER, launch_preprocess(imgs->images, imgs->half_in, imgs->heights, imgs->widths, batch, g.in_ch, imgs->mean,
                                                           imgs->std, cfg->img_h, cfg->img_w, big, stream));
            xp = big;
        }
        LDIT_TRY(embed(g, xp, F32(pm.patch_w), F32(pm.patch_b), F32(pm.cls), F32(pm.pos), h, batch, cfg->img_h, cfg->img_w, stream, probe));
    }
    if (float *t0 = tap_for(0)) {
        LDIT_HIP_CHECK(hipMemcpyAsync(t0, h, act_bytes, hipMemcpyDeviceToDevice, stream));
        LDIT_TRY(extra_taps(0, h, t0));
    }

    const float scale = 1.0f / sqrtf((float)g.D);
    const bool bf16 = cfg->dtype == LDIT_BF16, fp8 = cfg->dtype == LDIT_FP8;
    const int S = split_planes_of(cfg->dtype);
    // split-fp32 builds: the plane products of one output element, SMALLEST FIRST (their sum is formed at its own magnitude
    // before the leading p0.q0 term arrives): bf16x3 = a1 w0 + a0 w1 + a0 w0; six products = a2 w0 + a1 w1 + a0 w2 + a1 w0 + a0 w1 + a0 w0
    GemmExtra xs{};
    if (S == 2) { xs.nseg = 3; xs.seg_a = 0x001u; xs.seg_w = 0x010u; }
    if (S == 3) { xs.nseg = 6; xs.seg_a = 0x001012u; xs.seg_w = 0x010210u; }
    // per 64-deep k-tile (k-tile outermost): the operand tiles one product has just pulled through L2 serve the next one - 1-5 % on the
    // q|k|v and fc1 GEMMs against whole-K segments (profiles/r03_planes_segment_order.txt), never slower
    xs.seg_inner = 1;
    for (int l = 0; l < g.L; ++l) {
        const PackedLayer &pl = pm.layer[l];
        float *tap = tap_for(l + 1);
        if (S) {
            // fp32 forward on split operands (ldit.h, LDIT_F32X3 / LDIT_F32X6): every GEMM on the bf16 MFMA over the plane products of
            // its operands, fp32 accumulation; LayerNorm, attention (the fp32 kernel), erf-GELU, LayerScale + residual in fp32.
            char *ys = ws + wm.y, *bb = ws + wm.big;
            GemmExtra xg = xs;
            xg.nsplit_out = S;
            LDIT_RUN(probe, LDIT_K_LAYERNORM, launch_layernorm_splitout(h, F32(pl.ln1_w), F32(pl.ln1_b), ys, M, C, cfg->ln_eps, S, stream));
            {
                // q|k|v leave their GEMM as bf16 planes [M, S * 3C] (q pre-multiplied by scale log2 e at pack time) and the attention
                // runs on the plane products too (attention_planes.hip: 3 products for two planes, 6 for three)
                const __bf16 *qp = reinterpret_cast<const __bf16 *>(bb);
                LDIT_RUN(probe, LDIT_K_GEMM, launch_gemm_bf16_ex(ys, S * C, P + pl.wqkv, F32(pl.bqkv), bb, S * 3 * C, M, 3 * C, C, EPI_BIAS_SPLIT,
                                                                nullptr, nullptr, nullptr, xg, stream));
                LDIT_RUN(probe, LDIT_K_ATTENTION,
                         (S == 2 ? launch_attention_planes2 : launch_attention_planes3)(qp, qp + C, qp + 2 * C, ys, batch, g.T, g.H, g.D,
                                                                                        S * 3 * C, 3 * C, S * C, stream));
            }
            LDIT_RUN(probe, LDIT_K_GEMM, launch_gemm_bf16_ex(ys, S * C, P + pl.wo, F32(pl.bo), h, C, M, C, C, EPI_SCALE_RESID, F32(pl.lam1), h,
                                                            nullptr, xs, stream));
            LDIT_RUN(probe, LDIT_K_LAYERNORM, launch_layernorm_splitout(h, F32(pl.ln2_w), F32(pl.ln2_b), ys, M, C, cfg->ln_eps, S, stream));
            LDIT_RUN(probe, LDIT_K_GEMM, launch_gemm_bf16_ex(ys, S * C, P + pl.w1, F32(pl.b1), bb, S * F, M, F, C, EPI_GELU_SPLIT, nullptr, nullptr,
                                                            nullptr, xg, stream));
            LDIT_RUN(probe, LDIT_K_GEMM, launch_gemm_bf16_ex(bb, S * F, P + pl.w2, F32(pl.b2), h, C, M, C, F, EPI_SCALE_RESID, F32(pl.lam2), h,
                                                            tap, xs, stream));
        } else if (fp8) {
            // fp8 build: LayerNorm, the attention epilogue and the GELU epilogue quantise straight to e4m3 with the
            // calibrated per-tensor scales; q|k|v leave their GEMM as bf16 for the bf16 attention kernel.
            char *y8 = ws + wm.y, *bb = ws + wm.big;
            const float *sc = F32(pl.scales);
            LDIT_RUN(probe, LDIT_K_LAYERNORM, launch_layernorm_fp8out(h, F32(pl.ln1_w), F32(pl.ln1_b), y8, M, C, cfg->ln_eps, sc + 0, stream));
            LDIT_RUN(probe, LDIT_K_GEMM, launch_gemm_fp8(y8, C, P + pl.wqkv, F32(pl.bqkv), bb, 3 * C, M, 3 * C, C, EPI_BIAS, nullptr,
                                                        nullptr, nullptr, 0.f, 0.f, sc + 0, F32(pl.sw_qkv), nullptr, stream));
            LDIT_RUN(probe, LDIT_K_ATTENTION,
                     launch_attention_bf16_fp8out(bb, bb + 2 * (size_t)C, bb + 4 * (size_t)C, y8, batch, g.T, g.H, g.D, 3 * C, 3 * C,
                                                  3 * C, C, 0.0f /* q pre-scaled at pack time */, sc + 2, stream));
            LDIT_RUN(probe, LDIT_K_GEMM, launch_gemm_fp8(y8, C, P + pl.wo, F32(pl.bo), h, C, M, C, C, EPI_SCALE_RESID, F32(pl.lam1), h,
                                                        nullptr, 0.f, 0.f, sc + 2, F32(pl.sw_o), nullptr, stream));
            LDIT_RUN(probe, LDIT_K_LAYERNORM, launch_layernorm_fp8out(h, F32(pl.ln2_w), F32(pl.ln2_b), y8, M, C, cfg->ln_eps, sc + 4, stream));
            LDIT_RUN(probe, LDIT_K_GEMM, launch_gemm_fp8(y8, C, P + pl.w1, F32(pl.b1), bb, F, M, F, C, EPI_BIAS_GELU, nullptr, nullptr,
                                                        nullptr, 0.f, 0.f, sc + 4, F32(pl.sw_1), sc + 6, stream));
            LDIT_RUN(probe, LDIT_K_GEMM, launch_gemm_fp8(bb, F, P + pl.w2, F32(pl.b2), h, C, M, C, F, EPI_SCALE_RESID, F32(pl.lam2), h,
                                                        tap, 0.f, 0.f, sc + 6, F32(pl.sw_2), nullptr, stream));
        } else if (!bf16) {
            // y = LN1(h)                                                               TF:426
            LDIT_RUN(probe, LDIT_K_LAYERNORM, launch_layernorm(h, F32(pl.ln1_w), F32(pl.ln1_b), y, M, C, cfg->ln_eps, stream));
            // big[:, 0:3C] = y . [Wq;Wk;Wv]^T + [bq;0;bv]                               TF:319-321
            LDIT_TRY(linear(y, C, F32(pl.wqkv), F32(pl.bqkv), big, 3 * C, M, 3 * C, C, EPI_BIAS, nullptr, nullptr, nullptr, stream, probe));
            // y = softmax(q k^T / sqrt(D)) v, heads merged token-major                 TF:323-338
            LDIT_RUN(probe, LDIT_K_ATTENTION,
                     launch_attention(big, big + C, big + 2 * C, y, batch, g.T, g.H, g.D, 3 * C, 3 * C, 3 * C, C, scale, stream));
            // h += lam1 (.) (y . Wo^T + bo)                                             TF:339, 432-434
            LDIT_TRY(linear(y, C, F32(pl.wo), F32(pl.bo), h, C, M, C, C, EPI_SCALE_RESID, F32(pl.lam1), h, nullptr, stream, probe));
            // y = LN2(h)                                                               TF:438
            LDIT_RUN(probe, LDIT_K_LAYERNORM, launch_layernorm(h, F32(pl.ln2_w), F32(pl.ln2_b), y, M, C, cfg->ln_eps, stream));
            // big[:, 0:F] = gelu(y . W1^T + b1)                                         TF:353-354
            LDIT_TRY(linear(y, C, F32(pl.w1), F32(pl.b1), big, F, M, F, C, EPI_BIAS_GELU, nullptr, nullptr, nullptr, stream, probe));
            // h += lam2 (.) (big . W2^T + b2)  (+ tap copy of the new hidden state)      TF:355, 440-442
            LDIT_TRY(linear(big, F, F32(pl.w2), F32(pl.b2), h, C, M, C, F, EPI_SCALE_RESID, F32(pl.lam2), h, tap, stream, probe));
        } else {
            // bf16 build: residual stream h, LayerNorm statistics, softmax and every accumulation stay fp32; the GEMM /
            // attention operands (LN output, q|k|v, attention output, MLP hidden, the four weight matrices) are bf16.
            char *yb = ws + wm.y, *bb = ws + wm.big;                 // bf16 buffers
            LDIT_RUN(probe, LDIT_K_LAYERNORM, launch_layernorm_bf16out(h, F32(pl.ln1_w), F32(pl.ln1_b), yb, M, C, cfg->ln_eps, stream));
            LDIT_RUN(probe, LDIT_K_GEMM, launch_gemm_bf16(yb, C, P + pl.wqkv, F32(pl.bqkv), bb, 3 * C, M, 3 * C, C, EPI_BIAS,
                                                         nullptr, nullptr, nullptr, stream));
            LDIT_RUN(probe, LDIT_K_ATTENTION,
                     launch_attention_bf16(bb, bb + 2 * (size_t)C, bb + 4 * (size_t)C, yb, batch, g.T, g.H, g.D, 3 * C, 3 * C, 3 * C,
                                           C, 0.0f /* q pre-scaled at pack time */, stream));
            LDIT_RUN(probe, LDIT_K_GEMM, launch_gemm_bf16(yb, C, P + pl.wo, F32(pl.bo), h, C, M, C, C, EPI_SCALE_RESID, F32(pl.lam1),
                                                         h, nullptr, stream));
            LDIT_RUN(probe, LDIT_K_LAYERNORM, launch_layernorm_bf16out(h, F32(pl.ln2_w), F32(pl.ln2_b), yb, M, C, cfg->ln_eps, stream));
            LDIT_RUN(probe, LDIT_K_GEMM, launch_gemm_bf16(yb, C, P + pl.w1, F32(pl.b1), bb, F, M, F, C, EPI_BIAS_GELU, nullptr,
                                                         nullptr, nullptr, stream));
            LDIT_RUN(probe, LDIT_K_GEMM, launch_gemm_bf16(bb, F, P + pl.w2, F32(pl.b2), h, C, M, C, F, EPI_SCALE_RESID, F32(pl.lam2),
                                                         h, tap, stream));
        }
        if (tap) LDIT_TRY(extra_taps(l + 1, h, tap));
    }
    return LDIT_OK;
}

}  // namespace
}  // namespace ldit

using namespace ldit;

extern "C" {

int ldit_abi_version(void) { return LDIT_ABI_VERSION; }

/* diagnostic: re-read the LDIT_* switches of ldit_common.h from the environment (they are otherwise read once) */
int ldit_debug_reload_env(void)
{
    reload_diag();
    return LDIT_OK;
}

const char *ldit_last_error(void) { return err_buf(); }

size_t ldit_packed_bytes(const ldit_cfg *cfg)
{
    Geo g;
    if (geometry(cfg, g) != LDIT_OK) return 0;
    return packed_map(g, cfg->dtype).total;
}

int ldit_pack_weights(const ldit_cfg *cfg, const ldit_weights *w, void *packed, size_t packed_bytes, ldit_stream stream_)
{
    Geo g;
    LDIT_TRY(geometry(cfg, g));
    if (!w || !packed) return fail(LDIT_EINVAL, "null weights / packed pointer");
    if (!aligned16(packed)) return fail(LDIT_EINVAL, "packed must be 16-byte aligned");
    if (g.L && !w->layer) return fail(LDIT_EINVAL, "weights->layer is null");
    const PackedMap pm = packed_map(g, cfg->dtype);
    if (packed_bytes < pm.total) return fail(LDIT_EWORKSPACE, "packed buffer %zu bytes < required %zu", packed_bytes, pm.total);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    char *P = static_cast<char *>(packed);
    const bool bf16 = cfg->dtype == LDIT_BF16, fp8 = cfg->dtype == LDIT_FP8;
    auto put = [&](size_t off, const void *src, size_t n, const char *what) -> int {      // fp32 copy
        if (!src) return fail(LDIT_EINVAL, "weights: %s is null", what);
        LDIT_HIP_CHECK(hipMemcpyAsync(P + off, src, n * sizeof(float), hipMemcpyDeviceToDevice, stream));
        return LDIT_OK;
    };
    // matrix [rows, cols] at element offset elt_off of the block at `off`: fp32 copy, -> bf16, or -> fp8 codes with one
    // scale per row written to the fp32 vector at sw_off (+ row0)
    // `mul` (bf16 / fp8 builds only): a factor folded into the matrix as it is packed (bf16: into the values before the one
    // rounding; fp8: into the per-row scales).  Used for W_q: q' = (scale log2 e) q, so that the attention kernel's scores
    // are exp2-domain exponents (attention_bf16.hip, PRE) - the same single rounding of q as before, no extra error.
    auto put_mat = [&](size_t off, size_t elt_off, const void *src, size_t rows, size_t cols, size_t sw_off, size_t row0,
                       const char *what, float mul = 1.0f) -> int {
        if (!src) return fail(LDIT_EINVAL, "weights: %s is null", what);
        if (const int Sp = split_planes_of(cfg->dtype)) {
            // split-fp32 builds: the matrix as Sp bf16 planes side by side per row; `elt_off` counts fp32 elements of whole rows
            if (!aligned16(src)) return fail(LDIT_EINVAL, "weights: %s must be 16-byte aligned", what);
            return launch_split_planes(static_cast<const float *>(src), (int)cols, P + off + elt_off * 2 * Sp, (int)rows, (int)cols, Sp, stream, mul);
        }
        if (!bf16 && !fp8) return put(off + elt_off * 4, src, rows * cols, what);
        if (!aligned16(src)) return fail(LDIT_EINVAL, "weights: %s must be 16-byte aligned", what);
        if (fp8)
            return launch_quant_rows_fp8(static_cast<const float *>(src), P + off + elt_off,
                                         reinterpret_cast<float *>(P + sw_off) + row0, (int)rows, (int)cols, stream, mul);
        return launch_cvt_bf16(static_cast<const float *>(src), P + off + elt_off * 2, rows * cols, stream, mul);
    };
    // (the split-fp32 builds too: their attention runs on bf16-plane operands with exp2-domain scores, attention_planes.hip)
    const float qfold = (bf16 || fp8 || split_planes_of(cfg->dtype)) ? (1.0f / sqrtf((float)g.D)) * 1.44269504088896340736f : 1.0f;
    const size_t C = g.C, F = g.F;
    LDIT_TRY(put(pm.patch_w, w->patch_w, C * g.Kp, "patch_w"));
    if (bf16 || fp8) {
        if (!aligned16(w->patch_w)) return fail(LDIT_EINVAL, "weights: patch_w must be 16-byte aligned");
        LDIT_TRY(launch_cvt_bf16(static_cast<const float *>(w->patch_w), P + pm.patch_w16, C * g.Kp, stream));
    } else if (const int Sp = split_planes_of(cfg->dtype)) {
        if (!aligned16(w->patch_w)) return fail(LDIT_EINVAL, "weights: patch_w must be 16-byte aligned");
        LDIT_TRY(launch_split_planes(static_cast<const float *>(w->patch_w), g.Kp, P + pm.patch_w16, (int)C, g.Kp, Sp, stream));
    }
    LDIT_TRY(put(pm.patch_b, w->patch_b, C, "patch_b"));
    LDIT_TRY(put(pm.cls, w->cls, C, "cls"));
    LDIT_TRY(put(pm.pos, w->pos, (size_t)g.T * C, "pos"));
    for (int l = 0; l < g.L; ++l) {
        const ldit_layer_weights &s = w->layer[l];
        const PackedLayer &pl = pm.layer[l];
        LDIT_TRY(put(pl.ln1_w, s.ln1_w, C, "ln1_w"));
        LDIT_TRY(put(pl.ln1_b, s.ln1_b, C, "ln1_b"));
        if (fp8) LDIT_HIP_CHECK(hipMemsetAsync(P + pl.scales, 0, 8 * sizeof(float), stream));
        LDIT_TRY(put_mat(pl.wqkv, 0, s.wq, C, C, pl.sw_qkv, 0, "wq", qfold));
        LDIT_TRY(put_mat(pl.wqkv, C * C, s.wk, C, C, pl.sw_qkv, C, "wk"));
        LDIT_TRY(put_mat(pl.wqkv, 2 * C * C, s.wv, C, C, pl.sw_qkv, 2 * C, "wv"));
        if (!s.bq || !s.bv) return fail(LDIT_EINVAL, "weights: bq / bv is null");
        LDIT_TRY(launch_pack_qkv_bias(static_cast<const float *>(s.bq), static_cast<const float *>(s.bv),
                                      reinterpret_cast<float *>(P + pl.bqkv), g.C, stream, qfold));
        LDIT_TRY(put_mat(pl.wo, 0, s.wo, C, C, pl.sw_o, 0, "wo"));
        LDIT_TRY(put(pl.bo, s.bo, C, "bo"));
        LDIT_TRY(put(pl.lam1, s.lam1, C, "lam1"));
        LDIT_TRY(put(pl.ln2_w, s.ln2_w, C, "ln2_w"));
        LDIT_TRY(put(pl.ln2_b, s.ln2_b, C, "ln2_b"));
        LDIT_TRY(put_mat(pl.w1, 0, s.w1, F, C, pl.sw_1, 0, "w1"));
        LDIT_TRY(put(pl.b1, s.b1, F, "b1"));
        LDIT_TRY(put_mat(pl.w2, 0, s.w2, C, F, pl.sw_2, 0, "w2"));
        LDIT_TRY(put(pl.b2, s.b2, C, "b2"));
        LDIT_TRY(put(pl.lam2, s.lam2, C, "lam2"));
    }
    return LDIT_OK;
}

int ldit_set_fp8_act_scales(const ldit_cfg *cfg, void *packed, size_t packed_bytes, const float *act_scales, ldit_stream stream_)
{
    Geo g;
    LDIT_TRY(geometry(cfg, g));
    if (cfg->dtype != LDIT_FP8) return fail(LDIT_EINVAL, "set_fp8_act_scales: cfg.dtype is not LDIT_FP8");
    if (!packed || !act_scales) return fail(LDIT_EINVAL, "set_fp8_act_scales: null pointer");
    const PackedMap pm = packed_map(g, cfg->dtype);
    if (packed_bytes < pm.total) return fail(LDIT_EWORKSPACE, "packed buffer %zu bytes < required %zu", packed_bytes, pm.total);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    char *P = static_cast<char *>(packed);
    for (int l = 0; l < g.L; ++l)
        for (int a = 0; a < LDIT_FP8_A_COUNT; ++a) {
            const float v = act_scales[l * LDIT_FP8_A_COUNT + a];
            if (!(v > 0.0f) || !std::isfinite(v)) return fail(LDIT_EINVAL, "set_fp8_act_scales: layer %d scale %d = %g is not positive", l, a, (double)v);
            // slots 0, 2, 4, 6 of the layer's scale block; 4-byte pageable copies complete before the call returns
            LDIT_HIP_CHECK(hipMemcpyAsync(P + pm.layer[l].scales + 8 * a, &act_scales[l * LDIT_FP8_A_COUNT + a], sizeof(float),
                                          hipMemcpyHostToDevice, stream));
        }
    return LDIT_OK;
}

size_t ldit_workspace_bytes(const ldit_cfg *cfg, int32_t batch)
{
    Geo g;
    if (batch <= 0 || geometry(cfg, g) != LDIT_OK) return 0;
    return workspace_map(g, batch, cfg->dtype).total;
}

int ldit_vit_forward(const ldit_cfg *cfg, const void *packed, const void *x, int32_t batch, void *const *tap_out,
                     void *workspace, size_t workspace_bytes, ldit_stream stream)
{
    Probe probe;
    return forward(cfg, packed, x, batch, tap_out, workspace, workspace_bytes, static_cast<hipStream_t>(stream), probe);
}

int ldit_vit_forward_images(const ldit_cfg *cfg, const void *packed, const void *const *images, const int32_t *heights,
                            const int32_t *widths, int32_t half_in, float mean, float std, int32_t batch, void *const *tap_out,
                            void *workspace, size_t workspace_bytes, ldit_stream stream)
{
    Probe probe;
    const ImgSrc src{images, heights, widths, half_in != 0, mean, std};
    return forward(cfg, packed, nullptr, batch, tap_out, workspace, workspace_bytes, static_cast<hipStream_t>(stream), probe, &src);
}

int ldit_vit_forward_timed(const ldit_cfg *cfg, const void *packed, const void *x, int32_t batch, void *const *tap_out,
                           void *workspace, size_t workspace_bytes, ldit_stream stream, double *ms, int64_t *launches)
{
    if (!ms || !launches) return fail(LDIT_EINVAL, "ms / launches is null");
    Probe probe;
    probe.on = true;
    probe.stream = static_cast<hipStream_t>(stream);
    int rc = forward(cfg, packed, x, batch, tap_out, workspace, workspace_bytes, probe.stream, probe);
    int rc2 = probe.collect(ms, launches);
    return rc != LDIT_OK ? rc : rc2;
}

int ldit_linear_f32(const void *X, int64_t lda, const void *W, const void *bias, void *Y, int64_t ldy, int64_t M,
                    int64_t N, int64_t K, int32_t epilogue, const void *lam, const void *R, void *Y2, ldit_stream stream)
{
    if (M <= 0 || N <= 0 || K <= 0) return fail(LDIT_EINVAL, "linear: empty problem");
    if (M * (ldy > lda ? ldy : lda) >= (1ll << 31) || N * K >= (1ll << 31)) return fail(LDIT_EUNSUPPORTED, "linear: operand exceeds 2^31 elements");
    if (ldy < N || lda < K) return fail(LDIT_EINVAL, "linear: bad leading dimension");
    if (!Y || !aligned16(Y) || (Y2 && !aligned16(Y2))) return fail(LDIT_EINVAL, "linear: output null or misaligned");
    if (epilogue < LDIT_EPI_BIAS || epilogue > LDIT_EPI_SCALE_RESID) return fail(LDIT_EINVAL, "linear: unknown epilogue %d", epilogue);
    Probe probe;
    return linear(static_cast<const float *>(X), (int)lda, static_cast<const float *>(W), static_cast<const float *>(bias),
                  static_cast<float *>(Y), (int)ldy, (int)M, (int)N, (int)K, epilogue, static_cast<const float *>(lam),
                  static_cast<const float *>(R), static_cast<float *>(Y2), static_cast<hipStream_t>(stream), probe);
}

int ldit_layernorm_f32(const void *X, const void *gamma, const void *beta, void *Y, int64_t rows, int64_t C, float eps,
                       ldit_stream stream)
{
    if (C > (1 << 20)) return fail(LDIT_EINVAL, "layernorm: C out of range");
    return launch_layernorm(static_cast<const float *>(X), static_cast<const float *>(gamma), static_cast<const float *>(beta),
                            static_cast<float *>(Y), rows, (int)C, eps, static_cast<hipStream_t>(stream));
}

int ldit_attention_f32(const void *Q, const void *K, const void *V, void *O, int64_t B, int64_t N, int64_t H, int64_t D,
                       int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float scale, ldit_stream stream)
{
    if (B <= 0 || N <= 0 || H <= 0) return fail(LDIT_EINVAL, "attention: empty problem");
    const int64_t ldmax = ldq > ldk ? (ldq > ldv ? ldq : ldv) : (ldk > ldv ? ldk : ldv);
    if (B * N * (ldmax > ldo ? ldmax : ldo) >= (1ll << 31) || B * H * ((N + 127) / 128) >= (1ll << 31))
        return fail(LDIT_EUNSUPPORTED, "attention: operand exceeds 2^31 elements");
    if (ldq < H * D || ldk < H * D || ldv < H * D || ldo < H * D) return fail(LDIT_EINVAL, "attention: row stride smaller than H*D");
    return launch_attention(static_cast<const float *>(Q), static_cast<const float *>(K), static_cast<const float *>(V),
                            static_cast<float *>(O), (int)B, (int)N, (int)H, (int)D, (int)ldq, (int)ldk, (int)ldv, (int)ldo,
                            scale, static_cast<hipStream_t>(stream));
}

int ldit_embed_f32(const void *x, const void *patch_w, const void *patch_b, const void *cls, const void *pos, void *out,
                   int64_t B, int64_t in_ch, int64_t img_h, int64_t img_w, int64_t p, int64_t C, ldit_stream stream)
{
    if (B <= 0 || in_ch <= 0 || img_h <= 0 || img_w <= 0 || p <= 0 || C <= 0) return fail(LDIT_EINVAL, "embed: empty problem");
    if (img_h % p || img_w % p) return fail(LDIT_EINVAL, "embed: image %lldx%lld is not a multiple of patch %lld", (long long)img_h, (long long)img_w, (long long)p);
    if (!x || !patch_w || !patch_b || !cls || !pos || !out) return fail(LDIT_EINVAL, "embed: null operand");
    if (!aligned16(out) || !aligned16(pos) || (C & 3)) return fail(LDIT_EINVAL, "embed: out / pos must be 16-byte aligned, C a multiple of 4");
    Geo g{};
    g.C = (int)C; g.p = (int)p; g.in_ch = (int)in_ch; g.gh = (int)(img_h / p); g.gw = (int)(img_w / p);
    g.P = g.gh * g.gw; g.T = g.P + 1; g.Kp = (int)(in_ch * p * p);
    if (B * in_ch * img_h * img_w >= (1ll << 31) || B * g.T * C >= (1ll << 31)) return fail(LDIT_EUNSUPPORTED, "embed: operand exceeds 2^31 elements");
    Probe probe;
    return embed(g, static_cast<const float *>(x), static_cast<const float *>(patch_w), static_cast<const float *>(patch_b),
                 static_cast<const float *>(cls), static_cast<const float *>(pos), static_cast<float *>(out), (int)B,
                 (int)img_h, (int)img_w, static_cast<hipStream_t>(stream), probe);
}

int ldit_attention_planes(const void *Q, const void *K, const void *V, void *O, int64_t B, int64_t N, int64_t H, int64_t D,
                          int64_t ld_in, int64_t plane_in, int64_t ldo, int32_t planes, ldit_stream stream)
{
    if (planes != 2 && planes != 3) return fail(LDIT_EINVAL, "attention_planes: %d planes (2 or 3)", planes);
    if (B <= 0 || N <= 0 || H <= 0) return fail(LDIT_EINVAL, "attention_planes: empty problem");
    if (B * N * (ld_in > ldo ? ld_in : ldo) >= (1ll << 31) || B * H * ((N + 127) / 128) >= (1ll << 31))
        return fail(LDIT_EUNSUPPORTED, "attention_planes: operand exceeds 2^31 elements");
    return (planes == 2 ? launch_attention_planes2 : launch_attention_planes3)(Q, K, V, O, (int)B, (int)N, (int)H, (int)D, (int)ld_in,
                                                                              (int)plane_in, (int)ldo, static_cast<hipStream_t>(stream));
}

int ldit_split_f32_planes(const void *src, int64_t lds, void *dst, int64_t rows, int64_t cols, int32_t planes, ldit_stream stream)
{
    if (rows < 0 || cols < 0 || rows * (lds > planes * cols ? lds : planes * cols) >= (1ll << 31)) return fail(LDIT_EUNSUPPORTED, "split_planes: operand exceeds 2^31 elements");
    return launch_split_planes(static_cast<const float *>(src), (int)lds, dst, (int)rows, (int)cols, planes, static_cast<hipStream_t>(stream));
}

int ldit_layernorm_f32_planes(const void *X, const void *gamma, const void *beta, void *Y, int64_t rows, int64_t C, float eps,
                              int32_t planes, ldit_stream stream)
{
    if (C > (1 << 20)) return fail(LDIT_EINVAL, "layernorm: C out of range");
    return launch_layernorm_splitout(static_cast<const float *>(X), static_cast<const float *>(gamma), static_cast<const float *>(beta), Y,
                                     rows, (int)C, eps, planes, static_cast<hipStream_t>(stream));
}

int ldit_linear_planes(const void *Xp, int64_t lda, const void *Wp, const void *bias, void *Y, int64_t ldy, int64_t M, int64_t N,
                       int64_t K, int32_t epilogue, const void *lam, const void *R, void *Y2, int32_t planes, ldit_stream stream)
{
    if (M <= 0 || N <= 0 || K <= 0) return fail(LDIT_EINVAL, "linear_planes: empty problem");
    if (planes != 2 && planes != 3) return fail(LDIT_EINVAL, "linear_planes: %d planes (2 or 3)", planes);
    if (M * (ldy > lda ? ldy : lda) >= (1ll << 31) || N * K * planes >= (1ll << 31)) return fail(LDIT_EUNSUPPORTED, "linear_planes: operand exceeds 2^31 elements");
    if (lda < planes * K) return fail(LDIT_EINVAL, "linear_planes: bad leading dimension");
    if (!Y || !aligned16(Y) || (Y2 && !aligned16(Y2))) return fail(LDIT_EINVAL, "linear_planes: output null or misaligned");
    GemmExtra x{};
    if (planes == 2) { x.nseg = 3; x.seg_a = 0x001u; x.seg_w = 0x010u; }
    else { x.nseg = 6; x.seg_a = 0x001012u; x.seg_w = 0x010210u; }
    x.seg_inner = 1;
    int epi;
    if (epilogue == LDIT_EPI_BIAS) epi = EPI_F32;
    else if (epilogue == LDIT_EPI_SCALE_RESID) epi = EPI_SCALE_RESID;
    else if (epilogue == LDIT_EPI_BIAS_GELU) { epi = EPI_GELU_SPLIT; x.nsplit_out = planes; }
    else return fail(LDIT_EINVAL, "linear_planes: unknown epilogue %d", epilogue);
    if (epi != EPI_GELU_SPLIT && ldy < N) return fail(LDIT_EINVAL, "linear_planes: bad output stride");
    return launch_gemm_bf16_ex(Xp, (int)lda, Wp, static_cast<const float *>(bias), Y, (int)ldy, (int)M, (int)N, (int)K, epi,
                               static_cast<const float *>(lam), static_cast<const float *>(R), static_cast<float *>(Y2), x,
                               static_cast<hipStream_t>(stream));
}

int ldit_embed_bf16(const void *x, const void *patch_w_bf16, const void *patch_b, const void *cls, const void *pos, void *out,
                    void *scratch, int64_t B, int64_t in_ch, int64_t img_h, int64_t img_w, int64_t p, int64_t C, ldit_stream stream_)
{
    if (B <= 0 || in_ch <= 0 || img_h <= 0 || img_w <= 0 || p <= 0 || C <= 0) return fail(LDIT_EINVAL, "embed_bf16: empty problem");
    if (img_h % p || img_w % p) return fail(LDIT_EINVAL, "embed_bf16: image %lldx%lld is not a multiple of patch %lld", (long long)img_h, (long long)img_w, (long long)p);
    if (!x || !patch_w_bf16 || !patch_b || !cls || !pos || !out || !scratch) return fail(LDIT_EINVAL, "embed_bf16: null operand");
    if (!aligned16(out) || !aligned16(pos) || !aligned16(scratch) || !aligned16(patch_w_bf16) || (C & 3))
        return fail(LDIT_EINVAL, "embed_bf16: out / pos / scratch / patch_w must be 16-byte aligned, C a multiple of 4");
    const int64_t P = (img_h / p) * (img_w / p), Kp = in_ch * p * p;
    if (Kp % 64) return fail(LDIT_EUNSUPPORTED, "embed_bf16: in_ch*p*p = %lld must be a multiple of 64", (long long)Kp);
    if (B * in_ch * img_h * img_w >= (1ll << 31) || B * (P + 1) * C >= (1ll << 31) || B * P * Kp >= (1ll << 31))
        return fail(LDIT_EUNSUPPORTED, "embed_bf16: operand exceeds 2^31 elements");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    LDIT_TRY(launch_patches_rows(static_cast<const float *>(x), scratch, (int)B, (int)in_ch, (int)img_h, (int)img_w, (int)p, stream));
    GemmExtra xe{};
    xe.pos = static_cast<const float *>(pos); xe.patches = (int)P;
    LDIT_TRY(launch_gemm_bf16_ex(scratch, (int)Kp, patch_w_bf16, static_cast<const float *>(patch_b), out, (int)C, (int)(B * P), (int)C, (int)Kp,
                                 EPI_EMBED, nullptr, nullptr, nullptr, xe, stream));
    return launch_cls_rows(static_cast<const float *>(cls), static_cast<const float *>(pos), static_cast<float *>(out), (int)B, (int)(P + 1),
                           (int)C, stream);
}

int ldit_embed_bf16_images(const void *const *images, const int32_t *heights, const int32_t *widths, int32_t half_in, float mean,
                           float std, const void *patch_w_bf16, const void *patch_b, const void *cls, const void *pos, void *out,
                           void *scratch, int64_t B, int64_t in_ch, int64_t img_h, int64_t img_w, int64_t p, int64_t C, ldit_stream stream_)
{
    if (B <= 0 || B > 65535 || in_ch <= 0 || img_h <= 0 || img_w <= 0 || p <= 0 || C <= 0) return fail(LDIT_EINVAL, "embed_bf16_images: empty problem");
    if (img_h % p || img_w % p) return fail(LDIT_EINVAL, "embed_bf16_images: target %lldx%lld is not a multiple of patch %lld", (long long)img_h, (long long)img_w, (long long)p);
    if (!images || !heights || !widths || !patch_w_bf16 || !patch_b || !cls || !pos || !out || !scratch) return fail(LDIT_EINVAL, "embed_bf16_images: null operand");
    if (!(std > 0.0f)) return fail(LDIT_EINVAL, "embed_bf16_images: std must be positive");
    if (!aligned16(out) || !aligned16(pos) || !aligned16(scratch) || !aligned16(patch_w_bf16) || (C & 3))
        return fail(LDIT_EINVAL, "embed_bf16_images: out / pos / scratch / patch_w must be 16-byte aligned, C a multiple of 4");
    const int64_t P = (img_h / p) * (img_w / p), Kp = in_ch * p * p;
    if (Kp % 64) return fail(LDIT_EUNSUPPORTED, "embed_bf16_images: in_ch*p*p = %lld must be a multiple of 64", (long long)Kp);
    if (B * in_ch * img_h * img_w >= (1ll << 31) || B * (P + 1) * C >= (1ll << 31) || B * P * Kp >= (1ll << 31))
        return fail(LDIT_EUNSUPPORTED, "embed_bf16_images: operand exceeds 2^31 elements");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    LDIT_TRY(launch_patches_rows_images(images, half_in != 0, heights, widths, (int)B, (int)in_ch, mean, std, (int)img_h, (int)img_w, (int)p,
                                        scratch, stream));
    GemmExtra xe{};
    xe.pos = static_cast<const float *>(pos); xe.patches = (int)P;
    LDIT_TRY(launch_gemm_bf16_ex(scratch, (int)Kp, patch_w_bf16, static_cast<const float *>(patch_b), out, (int)C, (int)(B * P), (int)C, (int)Kp,
                                 EPI_EMBED, nullptr, nullptr, nullptr, xe, stream));
    return launch_cls_rows(static_cast<const float *>(cls), static_cast<const float *>(pos), static_cast<float *>(out), (int)B, (int)(P + 1),
                           (int)C, stream);
}

int ldit_tap_to_map_f32(const void *tap, void *out, int64_t B, int64_t Gh, int64_t Gw, int64_t C, float scale,
                        ldit_stream stream)
{
    if (!tap || !out || !aligned16(tap)) return fail(LDIT_EINVAL, "tap_to_map: null or misaligned operand");
    if (B * (Gh * Gw + 1) * C >= (1ll << 31)) return fail(LDIT_EUNSUPPORTED, "tap_to_map: operand exceeds 2^31 elements");
    return launch_tap_to_map(static_cast<const float *>(tap), static_cast<float *>(out), (int)B, (int)Gh, (int)Gw, (int)C,
                             scale, static_cast<hipStream_t>(stream));
}

int ldit_tap_to_map_bwd_f32(const void *dmap, void *dtap, int64_t B, int64_t Gh, int64_t Gw, int64_t C, float scale, ldit_stream stream)
{
    if (B * (Gh * Gw + 1) * C >= (1ll << 31) || (double)B * C * Gh * Gw * scale * scale >= 2147483648.0)
        return fail(LDIT_EUNSUPPORTED, "tap_to_map_bwd: operand exceeds 2^31 elements");
    return launch_tap_to_map_bwd(static_cast<const float *>(dmap), static_cast<float *>(dtap), (int)B, (int)Gh, (int)Gw, (int)C, scale,
                                 static_cast<hipStream_t>(stream));
}

int ldit_linear_bf16(const void *X, int64_t lda, const void *W, const void *bias, void *Y, int64_t ldy, int64_t M,
                     int64_t N, int64_t K, int32_t epilogue, const void *lam, const void *R, void *Y2, ldit_stream stream)
{
    if (M <= 0 || N <= 0 || K <= 0) return fail(LDIT_EINVAL, "linear_bf16: empty problem");
    if (M * (ldy > lda ? ldy : lda) >= (1ll << 31) || N * K >= (1ll << 31)) return fail(LDIT_EUNSUPPORTED, "linear_bf16: operand exceeds 2^31 elements");
    if (ldy < N || lda < K) return fail(LDIT_EINVAL, "linear_bf16: bad leading dimension");
    if (!Y || !aligned16(Y) || (Y2 && !aligned16(Y2))) return fail(LDIT_EINVAL, "linear_bf16: output null or misaligned");
    if (epilogue < LDIT_EPI_BIAS || epilogue > LDIT_EPI_SCALE_RESID) return fail(LDIT_EINVAL, "linear_bf16: unknown epilogue %d", epilogue);
    return launch_gemm_bf16(X, (int)lda, W, static_cast<const float *>(bias), Y, (int)ldy, (int)M, (int)N, (int)K, epilogue,
                            static_cast<const float *>(lam), static_cast<const float *>(R), static_cast<float *>(Y2),
                            static_cast<hipStream_t>(stream));
}

int ldit_attention_bf16(const void *Q, const void *K, const void *V, void *O, int64_t B, int64_t N, int64_t H, int64_t D,
                        int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float scale, ldit_stream stream)
{
    if (B <= 0 || N <= 0 || H <= 0) return fail(LDIT_EINVAL, "attention_bf16: empty problem");
    const int64_t ldmax = ldq > ldk ? (ldq > ldv ? ldq : ldv) : (ldk > ldv ? ldk : ldv);
    if (B * N * (ldmax > ldo ? ldmax : ldo) >= (1ll << 31) || B * H * ((N + 255) / 256) >= (1ll << 31))
        return fail(LDIT_EUNSUPPORTED, "attention_bf16: operand exceeds 2^31 elements");
    if (ldq < H * D || ldk < H * D || ldv < H * D || ldo < H * D) return fail(LDIT_EINVAL, "attention_bf16: row stride smaller than H*D");
    return launch_attention_bf16(Q, K, V, O, (int)B, (int)N, (int)H, (int)D, (int)ldq, (int)ldk, (int)ldv, (int)ldo, scale,
                                 static_cast<hipStream_t>(stream));
}

int ldit_cast_f32_bf16(const void *src, void *dst, int64_t n, ldit_stream stream)
{
    if (n < 0 || (n && (!src || !dst))) return fail(LDIT_EINVAL, "cast: null operand");
    if (!aligned16(src) || (reinterpret_cast<uintptr_t>(dst) & 7u)) return fail(LDIT_EINVAL, "cast: misaligned operand");
    return launch_cvt_bf16(static_cast<const float *>(src), dst, (size_t)n, static_cast<hipStream_t>(stream));
}

int ldit_linear_fp8(const void *X, int64_t lda, const void *W, const void *bias, void *Y, int64_t ldy, int64_t M, int64_t N,
                    int64_t K, int32_t epilogue, const void *lam, const void *R, void *Y2, float ab_scale, float out_inv_scale,
                    const void *w_scales, ldit_stream stream)
{
    if (M <= 0 || N <= 0 || K <= 0) return fail(LDIT_EINVAL, "linear_fp8: empty problem");
    if (w_scales && !aligned16(w_scales)) return fail(LDIT_EINVAL, "linear_fp8: w_scales must be 16-byte aligned");
    if (M * (ldy > lda ? ldy : lda) >= (1ll << 31) || N * K >= (1ll << 31)) return fail(LDIT_EUNSUPPORTED, "linear_fp8: operand exceeds 2^31 elements");
    if (ldy < N || lda < K) return fail(LDIT_EINVAL, "linear_fp8: bad leading dimension");
    if (!Y || (reinterpret_cast<uintptr_t>(Y) & 15u) || (Y2 && !aligned16(Y2))) return fail(LDIT_EINVAL, "linear_fp8: output null or misaligned");
    if (epilogue < LDIT_EPI_BIAS || epilogue > LDIT_EPI_SCALE_RESID) return fail(LDIT_EINVAL, "linear_fp8: unknown epilogue %d", epilogue);
    if (!(ab_scale > 0.0f) || (epilogue == LDIT_EPI_BIAS_GELU && !(out_inv_scale > 0.0f))) return fail(LDIT_EINVAL, "linear_fp8: scales must be positive");
    return launch_gemm_fp8(X, (int)lda, W, static_cast<const float *>(bias), Y, (int)ldy, (int)M, (int)N, (int)K, epilogue,
                           static_cast<const float *>(lam), static_cast<const float *>(R), static_cast<float *>(Y2), ab_scale,
                           out_inv_scale, nullptr, static_cast<const float *>(w_scales), nullptr, static_cast<hipStream_t>(stream));
}

int ldit_quant_rows_f32_fp8(const void *W, void *codes, void *scales, int64_t N, int64_t K, ldit_stream stream)
{
    if (N <= 0 || K <= 0 || N >= (1ll << 31) || K >= (1ll << 31)) return fail(LDIT_EINVAL, "quant_rows_fp8: bad shape");
    if (!codes || (reinterpret_cast<uintptr_t>(codes) & 3u)) return fail(LDIT_EINVAL, "quant_rows_fp8: codes null or misaligned");
    return launch_quant_rows_fp8(static_cast<const float *>(W), codes, static_cast<float *>(scales), (int)N, (int)K,
                                 static_cast<hipStream_t>(stream));
}

int ldit_quant_f32_fp8(const void *src, void *dst, int64_t n, float inv_scale, ldit_stream stream)
{
    if (n < 0 || (n && (!src || !dst))) return fail(LDIT_EINVAL, "quant_fp8: null operand");
    if (!aligned16(src) || (reinterpret_cast<uintptr_t>(dst) & 3u)) return fail(LDIT_EINVAL, "quant_fp8: misaligned operand");
    if (!(inv_scale > 0.0f)) return fail(LDIT_EINVAL, "quant_fp8: inv_scale must be positive");
    return launch_quant_fp8(static_cast<const float *>(src), dst, (size_t)n, inv_scale, nullptr, static_cast<hipStream_t>(stream));
}

int ldit_amax_f32(const void *src, int64_t n, void *out, ldit_stream stream)
{
    if (n < 0 || !out || (n && !src)) return fail(LDIT_EINVAL, "amax: null operand");
    return launch_amax_f32(static_cast<const float *>(src), (size_t)n, static_cast<float *>(out), false, static_cast<hipStream_t>(stream));
}

int ldit_preprocess_f32(const void *const *images, const int32_t *heights, const int32_t *widths, int32_t B, int32_t in_ch,
                        float mean, float std, int32_t out_h, int32_t out_w, void *out, ldit_stream stream)
{
    if (!images || !heights || !widths || !out) return fail(LDIT_EINVAL, "preprocess: null argument");
    if (B <= 0 || B > 65535 || in_ch <= 0 || out_h <= 0 || out_w <= 0) return fail(LDIT_EINVAL, "preprocess: bad geometry");
    if (!(std > 0.0f)) return fail(LDIT_EINVAL, "preprocess: std must be positive");
    if ((int64_t)B * in_ch * out_h * out_w >= (1ll << 31)) return fail(LDIT_EUNSUPPORTED, "preprocess: batch exceeds 2^31 elements");
    return launch_preprocess(images, false, heights, widths, B, in_ch, mean, std, out_h, out_w, static_cast<float *>(out),
                             static_cast<hipStream_t>(stream));
}

int ldit_preprocess_f16(const void *const *images, const int32_t *heights, const int32_t *widths, int32_t B, int32_t in_ch,
                        float mean, float std, int32_t out_h, int32_t out_w, void *out, ldit_stream stream)
{
    if (!images || !heights || !widths || !out) return fail(LDIT_EINVAL, "preprocess: null argument");
    if (B <= 0 || B > 65535 || in_ch <= 0 || out_h <= 0 || out_w <= 0) return fail(LDIT_EINVAL, "preprocess: bad geometry");
    if (!(std > 0.0f)) return fail(LDIT_EINVAL, "preprocess: std must be positive");
    if ((int64_t)B * in_ch * out_h * out_w >= (1ll << 31)) return fail(LDIT_EUNSUPPORTED, "preprocess: batch exceeds 2^31 elements");
    return launch_preprocess(images, true, heights, widths, B, in_ch, mean, std, out_h, out_w, static_cast<float *>(out),
                             static_cast<hipStream_t>(stream));
}

int ldit_fpn_merge_f32(const void *lat, const void *top, void *out, int64_t B, int64_t Gh, int64_t Gw, int64_t Ch, float scale,
                       int64_t top_h, int64_t top_w, ldit_stream stream)
{
    if (B * (Gh * Gw + 1) * Ch >= (1ll << 31)) return fail(LDIT_EUNSUPPORTED, "fpn_merge: operand exceeds 2^31 elements");
    return launch_fpn_merge(static_cast<const float *>(lat), static_cast<const float *>(top), static_cast<float *>(out), (int)B,
                            (int)Gh, (int)Gw, (int)Ch, scale, (int)top_h, (int)top_w, static_cast<hipStream_t>(stream));
}

int ldit_conv3x3_nhwc_f32(const void *x, const void *w, const void *bias, void *y, int64_t B, int64_t H, int64_t W, int64_t Cin,
                          int64_t Cout, const void *zeros, ldit_stream stream)
{
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return fail(LDIT_EINVAL, "conv3x3: empty problem");
    if (B * H * W * (Cin > Cout ? Cin : Cout) >= (1ll << 31) || Cout * 9 * Cin >= (1ll << 31)) return fail(LDIT_EUNSUPPORTED, "conv3x3: operand exceeds 2^31 elements");
    if (!x || !w || !y || !zeros || !aligned16(x) || !aligned16(w) || !aligned16(y) || !aligned16(zeros)) return fail(LDIT_EINVAL, "conv3x3: null or misaligned operand");
    GemmArgs a{};
    a.A = static_cast<const float *>(x); a.W = static_cast<const float *>(w); a.Y = static_cast<float *>(y);
    a.bias = static_cast<const float *>(bias);
    a.M = (int)(B * H * W); a.N = (int)Cout; a.K = (int)(9 * Cin); a.lda = (int)Cin; a.ldy = (int)Cout;
    a.conv_h = (int)H; a.conv_w = (int)W; a.conv_c = (int)Cin; a.zeros = static_cast<const float *>(zeros);
    return launch_gemm(a, EPI_BIAS, A_CONV3, static_cast<hipStream_t>(stream));
}

int ldit_fpn_merge_bwd_f32(const void *d_inner, void *d_lat, void *d_top, int64_t B, int64_t Gh, int64_t Gw, int64_t Ch, float scale,
                           int64_t top_h, int64_t top_w, ldit_stream stream)
{
    if (B * (Gh * Gw + 1) * Ch >= (1ll << 31)) return fail(LDIT_EUNSUPPORTED, "fpn_merge_bwd: operand exceeds 2^31 elements");
    return launch_fpn_merge_bwd(static_cast<const float *>(d_inner), static_cast<float *>(d_lat), static_cast<float *>(d_top), (int)B,
                                (int)Gh, (int)Gw, (int)Ch, scale, (int)top_h, (int)top_w, static_cast<hipStream_t>(stream));
}

int ldit_pad_nhwc_f32_bf16(const void *src, void *dst, int64_t B, int64_t H, int64_t W, int64_t C, ldit_stream stream)
{
    if (B * (H + 2) * (W + 2) * C >= (1ll << 33)) return fail(LDIT_EUNSUPPORTED, "pad_nhwc: operand too large");
    return launch_pad_nhwc_bf16(static_cast<const float *>(src), dst, (int)B, (int)H, (int)W, (int)C, static_cast<hipStream_t>(stream));
}

size_t ldit_colsum_scratch_bytes(int64_t M, int64_t N) { return (M > 0 && N > 0) ? colsum_scratch_bytes(M, N) : 0; }

int ldit_colsum_f32(const void *x, int64_t M, int64_t N, int64_t ldx, void *out, void *scratch, size_t scratch_bytes, ldit_stream stream)
{
    if (N >= (1ll << 31)) return fail(LDIT_EUNSUPPORTED, "colsum: too many columns");
    return launch_colsum_f32(static_cast<const float *>(x), M, (int)N, ldx, static_cast<float *>(out), static_cast<float *>(scratch),
                             scratch_bytes, false, static_cast<hipStream_t>(stream));
}

int ldit_colamax_f32(const void *x, int64_t M, int64_t N, int64_t ldx, void *out, void *scratch, size_t scratch_bytes, ldit_stream stream)
{
    if (N >= (1ll << 31)) return fail(LDIT_EUNSUPPORTED, "colamax: too many columns");
    return launch_colsum_f32(static_cast<const float *>(x), M, (int)N, ldx, static_cast<float *>(out), static_cast<float *>(scratch),
                             scratch_bytes, true, static_cast<hipStream_t>(stream));
}

int ldit_cast_f16_f32(const void *src, void *dst, int64_t n, ldit_stream stream)
{
    if (n < 0 || (n && (!src || !dst))) return fail(LDIT_EINVAL, "cast: null operand");
    if ((reinterpret_cast<uintptr_t>(src) & 7u) || !aligned16(dst)) return fail(LDIT_EINVAL, "cast: misaligned operand");
    return launch_cast_f16(src, dst, (size_t)n, true, static_cast<hipStream_t>(stream));
}

int ldit_cast_f32_f16(const void *src, void *dst, int64_t n, ldit_stream stream)
{
    if (n < 0 || (n && (!src || !dst))) return fail(LDIT_EINVAL, "cast: null operand");
    if (!aligned16(src) || (reinterpret_cast<uintptr_t>(dst) & 7u)) return fail(LDIT_EINVAL, "cast: misaligned operand");
    return launch_cast_f16(src, dst, (size_t)n, false, static_cast<hipStream_t>(stream));
}

}  // extern "C"
