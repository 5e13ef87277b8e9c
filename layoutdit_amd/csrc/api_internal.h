// Host-side internals shared by api.hip (inference entry points) and api_train.hip (train step): geometry checks, the
// byte layout of the packed parameter block and of the inference workspace, per-launch HIP-event bracketing.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "ldit_common.h"

namespace ldit {

inline size_t up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Geo {
    int C, L, H, F, D, p, in_ch, gh, gw, P, T, Kp;   // T tokens per image, Kp = in_ch*p*p
};

// bf16 planes per fp32 operand of the split-fp32 builds (0 = not a split build)
inline int split_planes_of(int dtype) { return dtype == LDIT_F32X3 ? 2 : dtype == LDIT_F32X6 ? 3 : 0; }

inline int geometry(const ldit_cfg *cfg, Geo &g)
{
    if (!cfg) return fail(LDIT_EINVAL, "cfg is null");
    if (cfg->dtype != LDIT_F32 && cfg->dtype != LDIT_BF16 && cfg->dtype != LDIT_FP8 && !split_planes_of(cfg->dtype))
        return fail(LDIT_EUNSUPPORTED, "dtype %d: only fp32 (0), bf16 (1), fp8 e4m3 (3) and split fp32 (4, 5) are implemented", cfg->dtype);
    g.C = cfg->hidden; g.L = cfg->layers; g.H = cfg->heads; g.F = cfg->mlp; g.p = cfg->patch; g.in_ch = cfg->in_ch;
    if (g.C <= 0 || g.L < 0 || g.H <= 0 || g.F <= 0 || g.p <= 0 || g.in_ch <= 0) return fail(LDIT_EINVAL, "cfg: non-positive dimension");
    if (g.C % g.H) return fail(LDIT_EINVAL, "cfg: hidden %d not divisible by heads %d", g.C, g.H);
    g.D = g.C / g.H;
    if (g.D != 64) return fail(LDIT_EUNSUPPORTED, "cfg: head_dim %d, only 64 is implemented", g.D);
    if (g.C % 32 || g.F % 32) return fail(LDIT_EUNSUPPORTED, "cfg: hidden and mlp must be multiples of 32");
    if ((cfg->dtype == LDIT_BF16 || split_planes_of(cfg->dtype)) && (g.C % 64 || g.F % 64))
        return fail(LDIT_EUNSUPPORTED, "cfg: bf16 needs hidden and mlp multiples of 64");
    if (cfg->dtype == LDIT_FP8 && (g.C % 128 || g.F % 128)) return fail(LDIT_EUNSUPPORTED, "cfg: fp8 needs hidden and mlp multiples of 128");
    if (cfg->img_h <= 0 || cfg->img_w <= 0 || cfg->img_h % g.p || cfg->img_w % g.p)
        return fail(LDIT_EINVAL, "cfg: image %dx%d is not a multiple of patch %d", cfg->img_h, cfg->img_w, g.p);
    g.gh = cfg->img_h / g.p; g.gw = cfg->img_w / g.p; g.P = g.gh * g.gw; g.T = g.P + 1;
    g.Kp = g.in_ch * g.p * g.p;
    if (g.Kp % 32 || g.p % 4) return fail(LDIT_EUNSUPPORTED, "cfg: in_ch*patch^2 must be a multiple of 32 and patch of 4");
    if (cfg->n_taps < 0 || cfg->n_taps > LDIT_MAX_TAPS) return fail(LDIT_EINVAL, "cfg: n_taps %d out of range", cfg->n_taps);
    for (int i = 0; i < cfg->n_taps; ++i)
        if (cfg->taps[i] < 0 || cfg->taps[i] > g.L) return fail(LDIT_EINVAL, "cfg: tap %d outside [0, %d]", cfg->taps[i], g.L);
    return LDIT_OK;
}

// Byte offsets into the packed parameter block; every offset is a multiple of 16 bytes.  In the bf16 build the four
// big matrices of a layer (fused q|k|v, o_proj, fc1, fc2) are stored as bf16, in the fp8 build as e4m3 codes; everything
// else stays fp32.  fp8 adds, per layer, one fp32 scale per output channel of each matrix (sw_*: measured and applied by
// ldit_pack_weights) and a block of 8 floats whose slots 0, 2, 4, 6 hold the activation scales a_ln1, a_attn, a_ln2,
// a_gelu (ldit_set_fp8_act_scales; the odd slots are spare).
struct PackedLayer { size_t ln1_w, ln1_b, wqkv, bqkv, wo, bo, lam1, ln2_w, ln2_b, w1, b1, w2, b2, lam2, scales, sw_qkv, sw_o, sw_1, sw_2; };
struct PackedMap {
    size_t patch_w, patch_b, cls, pos, total;
    size_t patch_w16;   // bf16 / fp8 builds: bf16 copy of the patch projection (the patch embedding runs on the bf16 GEMM there)
    std::vector<PackedLayer> layer;
};

inline PackedMap packed_map(const Geo &g, int dtype)
{
    PackedMap m;
    size_t o = 0;
    const size_t mat = dtype == LDIT_FP8 ? 1 : dtype == LDIT_BF16 ? 2 : split_planes_of(dtype) ? 2 * (size_t)split_planes_of(dtype) : 4;
    auto take = [&](size_t n, size_t elt) { size_t at = o; o += up(n * elt, 16); return at; };
    m.patch_w = take((size_t)g.C * g.Kp, 4);
    m.patch_b = take(g.C, 4);
    m.cls = take(g.C, 4);
    m.pos = take((size_t)g.T * g.C, 4);
    // bf16 copy of the patch projection (bf16 / fp8 builds) or its bf16 planes (split-fp32 builds): the patch embedding runs on the bf16 GEMM there
    m.patch_w16 = dtype != LDIT_F32 ? take((size_t)g.C * g.Kp, 2 * (size_t)(split_planes_of(dtype) ? split_planes_of(dtype) : 1)) : 0;
    m.layer.resize(g.L);
    for (int l = 0; l < g.L; ++l) {
        PackedLayer &pl = m.layer[l];
        pl.ln1_w = take(g.C, 4); pl.ln1_b = take(g.C, 4);
        pl.wqkv = take((size_t)3 * g.C * g.C, mat); pl.bqkv = take((size_t)3 * g.C, 4);
        pl.wo = take((size_t)g.C * g.C, mat); pl.bo = take(g.C, 4); pl.lam1 = take(g.C, 4);
        pl.ln2_w = take(g.C, 4); pl.ln2_b = take(g.C, 4);
        pl.w1 = take((size_t)g.F * g.C, mat); pl.b1 = take(g.F, 4);
        pl.w2 = take((size_t)g.C * g.F, mat); pl.b2 = take(g.C, 4); pl.lam2 = take(g.C, 4);
        pl.scales = dtype == LDIT_FP8 ? take(8, 4) : 0;
        pl.sw_qkv = dtype == LDIT_FP8 ? take((size_t)3 * g.C, 4) : 0;
        pl.sw_o = dtype == LDIT_FP8 ? take(g.C, 4) : 0;
        pl.sw_1 = dtype == LDIT_FP8 ? take(g.F, 4) : 0;
        pl.sw_2 = dtype == LDIT_FP8 ? take(g.C, 4) : 0;
    }
    m.total = o;
    return m;
}

struct Workspace { size_t h, y, big, total; };   // byte offsets

inline Workspace workspace_map(const Geo &g, int batch, int dtype)
{
    const size_t M = (size_t)batch * g.T;
    const size_t wide = (size_t)(3 * g.C > g.F ? 3 * g.C : g.F);
    const size_t act = dtype == LDIT_F32 ? 4 : 2;   // fp8 build: sized for its bf16 q|k|v; the fp8 buffers need less
    Workspace w;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += up(bytes, 256); return at; };
    w.h = take(M * g.C * 4);       // residual stream (always fp32)
    if (const size_t S = (size_t)split_planes_of(dtype)) {
        // split-fp32 builds: y = the S bf16 planes of the LayerNorm / attention output; big = the S bf16 planes of q|k|v
        // ([M, S * 3C], written by EPI_BIAS_SPLIT), later the S planes of the MLP hidden - each sized on its own: nothing
        // requires F >= 3C
        w.y = take(M * g.C * 2 * S);
        const size_t qkv = M * 3 * g.C * 2 * S, hid = M * g.F * 2 * S, patches = (size_t)batch * g.P * g.Kp * 2 * S;
        w.big = take(std::max(std::max(qkv, hid), patches));
        w.total = o;
        return w;
    }
    w.y = take(M * g.C * act);     // LayerNorm output, then attention output
    size_t big = M * wide * act;   // fused q|k|v, later the MLP hidden (never live together)
    if (dtype != LDIT_F32 && big < (size_t)batch * g.P * g.Kp * 2) big = (size_t)batch * g.P * g.Kp * 2;   // before layer 0: bf16 im2col of the batch
    // fp32 build fed by an image list (ldit_vit_forward_images): the transformed fp32 batch lives here until the embedding has read it
    // (only geometries narrower than 768 columns per token need the extra room)
    if (dtype == LDIT_F32 && big < (size_t)batch * g.P * g.Kp * 4) big = (size_t)batch * g.P * g.Kp * 4;
    w.big = take(big);
    w.total = o;
    return w;
}

// Optional per-launch HIP-event bracketing (ldit_vit_forward_timed).
struct Probe {
    bool on = false;
    hipStream_t stream = nullptr;
    std::vector<hipEvent_t> ev;
    std::vector<int> fam;
    int begin(int family)
    {
        if (!on) return LDIT_OK;
        hipEvent_t a, b;
        LDIT_HIP_CHECK(hipEventCreate(&a));
        LDIT_HIP_CHECK(hipEventCreate(&b));
        ev.push_back(a); ev.push_back(b); fam.push_back(family);
        LDIT_HIP_CHECK(hipEventRecord(a, stream));
        return LDIT_OK;
    }
    int end()
    {
        if (!on) return LDIT_OK;
        LDIT_HIP_CHECK(hipEventRecord(ev.back(), stream));
        return LDIT_OK;
    }
    int collect(double *ms, int64_t *launches)
    {
        if (!on) return LDIT_OK;
        LDIT_HIP_CHECK(hipStreamSynchronize(stream));
        for (size_t i = 0; i < fam.size(); ++i) {
            float t = 0.f;
            LDIT_HIP_CHECK(hipEventElapsedTime(&t, ev[2 * i], ev[2 * i + 1]));
            ms[fam[i]] += (double)t;
            launches[fam[i]] += 1;
        }
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
        ev.clear(); fam.clear();
        return LDIT_OK;
    }
};

#define LDIT_TRY(expr)                 \
    do {                               \
        int rc__ = (expr);             \
        if (rc__ != LDIT_OK) return rc__; \
    } while (0)

#define LDIT_RUN(probe, family, expr)  \
    do {                               \
        LDIT_TRY((probe).begin(family)); \
        LDIT_TRY(expr);                \
        LDIT_TRY((probe).end());       \
    } while (0)


}  // namespace ldit
