// C-ABI entry points of the train step (BASELINE.json configs[2]: ViT-B/16 bs=64 bf16 forward + backward + AdamW;
// SURVEY.md 8(f)-3; reference loop: ref src/layoutdit/training/trainer.py:148-187, optimizer :62-76).  Host-side sequencing
// only, same rules as api.hip: enqueue-only on the caller's stream, no allocation, no CPU fallback.
//
// What is differentiated: hs = self.dit(x).hidden_states (ref dit_backbone.py:47) with respect to every encoder parameter,
// for upstream gradients arriving at the tapped hidden states.  bf16 build only: GEMM / attention operands bf16, residual
// stream, LayerNorm statistics, softmax, all accumulation and every gradient that leaves the library fp32.
//
// Parameters, gradients and optimizer state are FLAT fp32 blocks in the layout of the fp32 packed block (packed_map(F32)):
// patch_w, patch_b, cls, pos, then per layer ln1_w, ln1_b, wqkv [3C,C], bqkv [3C] (the key third stays zero: BEiT has no
// key bias, TF:306), wo, bo, lam1, ln2_w, ln2_b, w1, b1, w2, b2, lam2.  ldit_flat_param_layout reports the offsets; the
// PyTorch mirror makes its nn.Parameters views of that block (layoutdit_amd/training.py).
//
// Backward of one layer = 8 MFMA GEMMs (2x the forward's FLOPs) on gemm_bf16_tr.hip, which reads reduction-major operands
// through transposing LDS reads - nothing is transposed in memory:
//   dgrad  dX[M,K] = dY[M,N] . W[N,K]      A = dY (K-contiguous), B = W exactly as nn.Linear stores it ([reduction N][K])
//   wgrad  dW[N,K] = dY^T . X              both operands [reduction = tokens][outputs], as the forward / the dgrad left them;
//                                          K = B*N tokens is split over workgroups, fp32 slabs summed in a fixed order
// plus attention backward (attention_bwd_bf16.hip), LayerScale / LayerNorm backward with two-stage column reductions
// (train_ops.hip).  Nothing uses atomics: gradients are bit-reproducible.
#include "api_internal.h"

namespace ldit {
namespace {

struct SavedLayer { size_t h_in, y1, qkv, lse, o, z1, h_mid, y2, a1, g, z2; };
struct SavedMap {
    std::vector<SavedLayer> layer;
    size_t rowscale, h_final, total;
    size_t patches;    // bf16 im2col of the batch [B P, 3 p p]: operand of the patch-embedding GEMM (forward) and of its wgrad
};

inline int pad64(int v) { return (v + 63) / 64 * 64; }

SavedMap saved_map(const Geo &g, int batch)
{
    SavedMap m;
    const size_t M = (size_t)batch * g.T, C = g.C, F = g.F;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += up(bytes, 256); return at; };
    m.layer.resize(g.L);
    for (int l = 0; l < g.L; ++l) {
        SavedLayer &s = m.layer[l];
        s.h_in = take(M * C * 4);
        s.y1 = take(M * C * 2);
        s.qkv = take(M * 3 * C * 2);
        s.lse = take((size_t)batch * g.H * g.T * 4);
        s.o = take(M * C * 2);
        s.z1 = take(M * C * 2);
        s.h_mid = take(M * C * 4);
        s.y2 = take(M * C * 2);
        s.a1 = take(M * F * 2);
        s.g = take(M * F * 2);
        s.z2 = take(M * C * 2);
    }
    m.rowscale = take((size_t)2 * (g.L > 0 ? g.L : 1) * M * 4);
    m.h_final = take(M * C * 4);
    m.patches = take((size_t)batch * g.P * g.Kp * 2);
    m.total = o;
    return m;
}

constexpr int MAX_SPLITS = 8;

// K-splits of a wgrad GEMM with an Mout x Nout output over `nk` k-tiles.  The output is a handful of tiles and K = all
// tokens of the batch, so K is cut until one round of the machine is full: 128 x 128 tiles run two per CU (512 slots).
// Every split costs one more fp32 slab to write and to sum, hence as few as fill that round, at most 8.
int pick_splits(int Mout, int Nout, int nk)
{
    // (Round 4, scripts/wgrad_tiles.py: timed alone, GEMM + slab sum, the eight-wave 256 x 256 tile with K cut six to eight ways
    // wins - W2 / W1 [768 x 3072] 81.8 -> 71.4 us.  Inside the train step it loses: twice the slabs are twice the fp32 traffic of
    // the reduction, which there runs from HBM - per layer wgrad 238 -> 223 us but slab sums 28 -> 53 us, whether they are
    // queued or issued right behind their GEMM (profiles/r04_wgrad_tiles.txt).  Few splits on the small tile stay.)
    const long tiles = (long)((Mout + 127) / 128) * ((Nout + 127) / 128);
    int s = (int)(512 / tiles);
    if (s > MAX_SPLITS) s = MAX_SPLITS;
    if (s > nk) s = nk;
    if (s < 1) s = 1;
    while (s > 1 && ((nk + s - 1) / s) * (s - 1) >= nk) --s;      // no empty slab
    return s;
}

struct TrainWs {
    size_t zeros, dh, dy, dz, da1, dob, dqkv, patches;
    size_t slab[4];                 // wgrad slabs: w2 / patch_w, w1, wo, wqkv
    size_t part[10];                // partial column sums, see backward()
    size_t part_rows_tile, part_rows_ln;
    size_t total;
};

TrainWs train_ws_map(const Geo &g, int batch)
{
    TrainWs w;
    const size_t M = (size_t)batch * g.T, C = g.C, F = g.F, Mp = pad64((int)M);
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += up(bytes, 256); return at; };
    w.zeros = take(256);                                  // zero page: reduction rows past the last token
    w.dh = take(M * C * 4);
    w.dy = take(M * C * 4);
    w.dz = take(M * C * 2);
    w.da1 = take(M * F * 2);
    w.dob = take(M * C * 2);
    w.dqkv = take(M * 3 * C * 2);
    w.patches = take((size_t)batch * g.P * g.Kp * 2);     // im2col of the batch, bf16 (patch-embedding wgrad)
    const size_t S = MAX_SPLITS;
    w.slab[0] = take(S * std::max(C * F, C * (size_t)g.Kp) * 4);
    w.slab[1] = take(S * F * C * 4);
    w.slab[2] = take(S * C * C * 4);
    w.slab[3] = take(S * 3 * C * C * 4);
    w.part_rows_tile = Mp / 64;
    w.part_rows_ln = (size_t)layernorm_bwd_blocks((int64_t)M);
    const size_t widths[10] = {C, C, F, C, C, C, C, 3 * C, C, C};
    const bool ln[10] = {false, false, false, true, true, false, false, false, true, true};
    (void)ln;
    const size_t part_rows = std::max(std::max(w.part_rows_ln, w.part_rows_tile), (size_t)layernorm_bwd_resid_blocks((int64_t)M));     // every buffer fits any producer
    for (int i = 0; i < 10; ++i) w.part[i] = take(part_rows * widths[i] * 4);
    w.total = o;
    return w;
}

int train_geometry(const ldit_cfg *cfg, Geo &g)
{
    LDIT_TRY(geometry(cfg, g));
    if (cfg->dtype != LDIT_BF16) return fail(LDIT_EUNSUPPORTED, "train step: only the bf16 build (cfg.dtype = LDIT_BF16) is implemented");
    return LDIT_OK;
}

int gemm(Probe &probe, const void *A, int lda, const void *W, const float *bias, void *Y, int ldy, int M, int N, int K, int epi,
         const float *lam, const float *R, float *Y2, const GemmExtra &x, hipStream_t stream)
{
    LDIT_RUN(probe, LDIT_K_GEMM, launch_gemm_bf16_ex(A, lda, W, bias, Y, ldy, M, N, K, epi, lam, R, Y2, x, stream));
    return LDIT_OK;
}

float *tap_of(const ldit_cfg *cfg, void *const *taps, int hidden_idx)
{
    if (!taps) return nullptr;
    for (int i = 0; i < cfg->n_taps; ++i)
        if (cfg->taps[i] == hidden_idx && taps[i]) return static_cast<float *>(taps[i]);
    return nullptr;
}

int forward_train(const ldit_cfg *cfg, const void *packed, const void *flat_params, const void *x, int32_t batch,
                  void *const *tap_out, const void *drop_scales, void *saved, size_t saved_bytes, hipStream_t stream, Probe &probe)
{
    Geo g;
    LDIT_TRY(train_geometry(cfg, g));
    if (batch <= 0) return fail(LDIT_EINVAL, "batch %d must be positive", batch);
    if (!packed || !flat_params || !x || !saved) return fail(LDIT_EINVAL, "null packed / flat_params / x / saved pointer");
    if (!aligned16(packed) || !aligned16(flat_params) || !aligned16(x) || !aligned16(saved)) return fail(LDIT_EINVAL, "pointers must be 16-byte aligned");
    if (cfg->n_taps && !tap_out) return fail(LDIT_EINVAL, "tap_out is null");
    for (int i = 0; i < cfg->n_taps; ++i)
        if (!tap_out[i] || !aligned16(tap_out[i])) return fail(LDIT_EINVAL, "tap_out[%d] is null or misaligned", i);
    if ((int64_t)batch * g.T * (int64_t)(g.F > 3 * g.C ? g.F : 3 * g.C) >= (1ll << 31))
        return fail(LDIT_EUNSUPPORTED, "batch %d: activation index space exceeds 2^31 elements, split the batch", batch);
    const SavedMap sm = saved_map(g, batch);
    if (saved_bytes < sm.total) return fail(LDIT_EWORKSPACE, "saved-activation block %zu bytes < required %zu", saved_bytes, sm.total);
    // `packed` is the bf16 MIRROR of the flat fp32 block (element i of the mirror = bf16 of element i of the master): a matrix
    // sits at half its fp32 byte offset.  Every fp32 vector (and the fp32 patch projection) is read from the master itself.
    const PackedMap pm = packed_map(g, LDIT_F32);
    const char *P = static_cast<const char *>(packed), *FP = static_cast<const char *>(flat_params);
    auto F32 = [&](size_t off) { return reinterpret_cast<const float *>(FP + off); };
    char *S = static_cast<char *>(saved);
    const int M = batch * g.T, C = g.C, F = g.F;
    const size_t act_bytes = (size_t)M * C * 4;
    float *rowscale = drop_scales ? reinterpret_cast<float *>(S + sm.rowscale) : nullptr;
    if (drop_scales && g.L)
        LDIT_RUN(probe, LDIT_K_OTHER, launch_expand_rowscale(static_cast<const float *>(drop_scales), rowscale, batch, g.T, 2 * g.L, stream));

    auto h_of = [&](int l) { return reinterpret_cast<float *>(S + (l < g.L ? sm.layer[l].h_in : sm.h_final)); };
    auto copy_taps = [&](int hidden_idx, const float *src, const float *already) -> int {
        for (int i = 0; i < cfg->n_taps; ++i)
            if (cfg->taps[i] == hidden_idx && tap_out[i] != already)
                LDIT_HIP_CHECK(hipMemcpyAsync(tap_out[i], src, act_bytes, hipMemcpyDeviceToDevice, stream));
        return LDIT_OK;
    };

    // embeddings (TF:153-176): the batch rounded to a bf16 im2col matrix (kept in the saved block: the wgrad's operand too), then
    // the bf16 MFMA GEMM on the mirror of the patch projection - mixed precision like every other GEMM of the step (the fp32
    // kernel took 161 us of a 13.4 ms ViT-B bs=64 step).  A patch length that is not a multiple of the bf16 k-tile stays fp32.
    if (g.Kp % 64 == 0) {
        char *patches = S + sm.patches;
        LDIT_RUN(probe, LDIT_K_OTHER, launch_patches_rows(static_cast<const float *>(x), patches, batch, g.in_ch, cfg->img_h, cfg->img_w,
                                                         g.p, stream));
        GemmExtra xe{};
        xe.pos = F32(pm.pos); xe.patches = g.P;
        LDIT_TRY(gemm(probe, patches, g.Kp, P + pm.patch_w / 2, F32(pm.patch_b), h_of(0), C, batch * g.P, C, g.Kp, EPI_EMBED, nullptr,
                      nullptr, nullptr, xe, stream));
        LDIT_RUN(probe, LDIT_K_OTHER, launch_cls_rows(F32(pm.cls), F32(pm.pos), h_of(0), batch, g.T, C, stream));
        LDIT_TRY(copy_taps(0, h_of(0), nullptr));
    } else {
        GemmArgs a{};
        a.A = static_cast<const float *>(x); a.W = F32(pm.patch_w); a.Y = h_of(0); a.bias = F32(pm.patch_b); a.pos = F32(pm.pos);
        a.M = batch * g.P; a.N = C; a.K = g.Kp; a.lda = g.in_ch * cfg->img_h * cfg->img_w; a.ldy = C;
        a.img_h = cfg->img_h; a.img_w = cfg->img_w; a.gw = g.gw; a.patches = g.P; a.patch = g.p; a.tokens = g.T;
        LDIT_RUN(probe, LDIT_K_GEMM, launch_gemm(a, EPI_EMBED, A_PATCH, stream));
        LDIT_RUN(probe, LDIT_K_OTHER, launch_cls_rows(F32(pm.cls), F32(pm.pos), h_of(0), batch, g.T, C, stream));
        LDIT_TRY(copy_taps(0, h_of(0), nullptr));
    }
    const float scale = 1.0f / sqrtf((float)g.D);
    for (int l = 0; l < g.L; ++l) {
        const PackedLayer &pl = pm.layer[l];
        const SavedLayer &sl = sm.layer[l];
        float *h_in = h_of(l), *h_mid = reinterpret_cast<float *>(S + sl.h_mid), *h_out = h_of(l + 1);
        char *y1 = S + sl.y1, *qkv = S + sl.qkv, *o = S + sl.o, *y2 = S + sl.y2, *gl = S + sl.g;
        float *tap = tap_of(cfg, tap_out, l + 1);
        GemmExtra none{}, x1{}, x2{}, x3{};
        x1.Ypre = S + sl.z1; x1.rowscale = rowscale ? rowscale + (size_t)(2 * l) * M : nullptr;
        x2.Ypre = S + sl.a1;                       // gelu'(pre-activation), the backward's factor
        x3.Ypre = S + sl.z2; x3.rowscale = rowscale ? rowscale + (size_t)(2 * l + 1) * M : nullptr;
        LDIT_RUN(probe, LDIT_K_LAYERNORM, launch_layernorm_bf16out(h_in, F32(pl.ln1_w), F32(pl.ln1_b), y1, M, C, cfg->ln_eps, stream));
        LDIT_TRY(gemm(probe, y1, C, P + pl.wqkv / 2, F32(pl.bqkv), qkv, 3 * C, M, 3 * C, C, EPI_BIAS, nullptr, nullptr, nullptr, none, stream));
        LDIT_RUN(probe, LDIT_K_ATTENTION,
                 launch_attention_bf16_lse(qkv, qkv + 2 * (size_t)C, qkv + 4 * (size_t)C, o, reinterpret_cast<float *>(S + sl.lse),
                                           batch, g.T, g.H, g.D, 3 * C, 3 * C, 3 * C, C, scale, stream));
        LDIT_TRY(gemm(probe, o, C, P + pl.wo / 2, F32(pl.bo), h_mid, C, M, C, C, EPI_SCALE_RESID, F32(pl.lam1), h_in, nullptr, x1, stream));
        LDIT_RUN(probe, LDIT_K_LAYERNORM, launch_layernorm_bf16out(h_mid, F32(pl.ln2_w), F32(pl.ln2_b), y2, M, C, cfg->ln_eps, stream));
        LDIT_TRY(gemm(probe, y2, C, P + pl.w1 / 2, F32(pl.b1), gl, F, M, F, C, EPI_BIAS_GELU, nullptr, nullptr, nullptr, x2, stream));
        LDIT_TRY(gemm(probe, gl, F, P + pl.w2 / 2, F32(pl.b2), h_out, C, M, C, F, EPI_SCALE_RESID, F32(pl.lam2), h_mid, tap, x3, stream));
        if (tap) LDIT_TRY(copy_taps(l + 1, h_out, tap));
    }
    return LDIT_OK;
}

// one wgrad:  out[Nout, Kout] = dY[tokens, Nout]^T . X[tokens, Kout], both operands as they lie in memory; split-K slabs
int wgrad(Probe &probe, ReduceJobs &jobs, const void *dY, int ld_dy, const void *X, int ld_x, float *out, float *slab, int Nout,
          int Kout, int tokens, const void *zeros, hipStream_t stream)
{
    GemmExtra x{};
    x.zeros = zeros;
    x.splits = pick_splits(Nout, Kout, (tokens + 63) / 64);
    float *dst = x.splits == 1 ? out : slab;
    LDIT_RUN(probe, LDIT_K_GEMM, launch_gemm_bf16_tr(dY, ld_dy, true, X, ld_x, nullptr, dst, Kout, Nout, Kout, tokens, EPI_F32, x, stream));
    if (x.splits > 1) {
        if (jobs.full()) LDIT_RUN(probe, LDIT_K_OTHER, launch_reduce_jobs(jobs, stream));
        jobs.add(slab, out, (int64_t)Nout * Kout, x.splits, (int64_t)Nout * Kout);
    }
    return LDIT_OK;
}

// one dgrad:  dX[M, Kout] = epi( dY[M, Nred] . W[Nred, Kout] ), W the bf16 copy of the nn.Linear weight as stored
int dgrad(Probe &probe, const void *dY, int Nred, const void *W, void *dX, int M, int Kout, int epi, const void *aux, const void *zeros,
          hipStream_t stream, float *colsum = nullptr)
{
    GemmExtra x{};
    x.zeros = zeros; x.aux = aux; x.ldaux = Kout; x.colsum = colsum;
    LDIT_RUN(probe, LDIT_K_GEMM, launch_gemm_bf16_tr(dY, Nred, false, W, Kout, nullptr, dX, Kout, M, Kout, Nred, epi, x, stream));
    return LDIT_OK;
}

int backward(const ldit_cfg *cfg, const void *flat_params, const void *packed, const void *x, int32_t batch, void *const *dtaps,
             const void *drop_scales, const void *saved, size_t saved_bytes, void *grads, size_t grads_bytes, void *workspace, size_t ws_bytes,
             int stage_hi, int stage_lo, hipStream_t stream, Probe &probe)
{
    Geo g;
    LDIT_TRY(train_geometry(cfg, g));
    if (batch <= 0) return fail(LDIT_EINVAL, "batch %d must be positive", batch);
    if (!flat_params || !packed || !x || !saved || !grads || !workspace) return fail(LDIT_EINVAL, "backward: null pointer");
    if (!aligned16(flat_params) || !aligned16(packed) || !aligned16(x) || !aligned16(saved) || !aligned16(grads) || !aligned16(workspace))
        return fail(LDIT_EINVAL, "backward: pointers must be 16-byte aligned");
    if (stage_hi > g.L || stage_lo < 0 || stage_lo > stage_hi) return fail(LDIT_EINVAL, "backward: stages [%d, %d] outside [0, %d]", stage_lo, stage_hi, g.L);
    const SavedMap sm = saved_map(g, batch);
    if (saved_bytes < sm.total) return fail(LDIT_EWORKSPACE, "saved-activation block %zu bytes < required %zu", saved_bytes, sm.total);
    const TrainWs wm = train_ws_map(g, batch);
    if (ws_bytes < wm.total) return fail(LDIT_EWORKSPACE, "workspace %zu bytes < required %zu", ws_bytes, wm.total);
    const PackedMap gm = packed_map(g, LDIT_F32), &pm = gm;      // vectors are read from the flat fp32 master
    if (grads_bytes < gm.total) return fail(LDIT_EWORKSPACE, "gradient block %zu bytes < required %zu", grads_bytes, gm.total);
    for (int i = 0; dtaps && i < cfg->n_taps; ++i)
        if (dtaps[i] && !aligned16(dtaps[i])) return fail(LDIT_EINVAL, "backward: dtaps[%d] misaligned", i);

    const char *P = static_cast<const char *>(flat_params), *W16 = static_cast<const char *>(packed), *S = static_cast<const char *>(saved);
    char *G = static_cast<char *>(grads), *ws = static_cast<char *>(workspace);
    auto F32 = [&](size_t off) { return reinterpret_cast<const float *>(P + off); };
    auto GR = [&](size_t off) { return reinterpret_cast<float *>(G + off); };
    auto PART = [&](int i) { return reinterpret_cast<float *>(ws + wm.part[i]); };
    const int M = batch * g.T, C = g.C, F = g.F, Mp = pad64(M);
    const size_t act = (size_t)M * C;
    float *dh = reinterpret_cast<float *>(ws + wm.dh), *dy = reinterpret_cast<float *>(ws + wm.dy);
    char *dz = ws + wm.dz, *da1 = ws + wm.da1, *dob = ws + wm.dob, *dqkv = ws + wm.dqkv;
    const char *zeros = ws + wm.zeros;
    LDIT_HIP_CHECK(hipMemsetAsync(ws + wm.zeros, 0, 256, stream));
    // stochastic depth: the forward left the per-row factors (expanded from drop_scales) in the saved block
    const float *rowscale = drop_scales ? reinterpret_cast<const float *>(S + sm.rowscale) : nullptr;
    const float scale = 1.0f / sqrtf((float)g.D);
    const int rows_tile = (int)wm.part_rows_tile, rows_ln = (int)wm.part_rows_ln;
    const int rows_lnr = C <= 1024 ? layernorm_bwd_resid_blocks((int64_t)M) : rows_ln;     // partial rows of the fused LayerNorm + LayerScale backward
    ReduceJobs jobs;

    // Upstream gradient arriving at hidden state `st` (dtaps).  For the last hidden state it IS dh's first value (a copy; anything
    // more is added); for every other state it rides in the LayerNorm backward that completes dh for that state - layer st's
    // layernorm_before, the last kernel of iteration st + 1, in this call or the previous stage's - so dh makes no extra round trip.
    // (Stages run L .. 0 in order on one workspace: dh itself already travels from call to call.)
    auto tap_grads = [&](int st, const float *(&found)[8]) {
        int n = 0;
        for (int i = 0; dtaps && i < cfg->n_taps && n < 8; ++i)
            if (cfg->taps[i] == st && dtaps[i]) found[n++] = static_cast<const float *>(dtaps[i]);
        return n;
    };
    if (stage_hi == g.L) {
        const float *t[8];
        const int n = tap_grads(g.L, t);
        if (n == 0) LDIT_HIP_CHECK(hipMemsetAsync(dh, 0, act * 4, stream));
        else LDIT_HIP_CHECK(hipMemcpyAsync(dh, t[0], act * 4, hipMemcpyDeviceToDevice, stream));
        for (int i = 1; i < n; ++i) LDIT_RUN(probe, LDIT_K_OTHER, launch_add_inplace(dh, t[i], act, stream));
    }
    for (int st = stage_hi; st >= stage_lo; --st) {
        if (st == 0) break;
        const int l = st - 1;
        const PackedLayer &pl = pm.layer[l], &gl = gm.layer[l];
        const SavedLayer &sl = sm.layer[l];
        const float *rs1 = rowscale ? rowscale + (size_t)(2 * l) * M : nullptr, *rs2 = rowscale ? rowscale + (size_t)(2 * l + 1) * M : nullptr;

        // ---- MLP branch:  h_out = h_mid + rs2 lam2 (.) (gelu(y2 W1^T + b1) W2^T + b2) --------------------------------
        LDIT_RUN(probe, LDIT_K_OTHER, launch_resid_bwd(dh, S + sl.z2, F32(pl.lam2), rs2, dz, nullptr, M, C, Mp, PART(0), PART(1), stream));
        LDIT_TRY(wgrad(probe, jobs, dz, C, S + sl.g, F, GR(gl.w2), reinterpret_cast<float *>(ws + wm.slab[0]), C, F, M, zeros, stream));
        // da1 = (dz W2) (.) gelu'; its column sums (the fc1 bias gradient) leave the same epilogue (F a multiple of 256), else a pass
        const bool b1_fused = F % 256 == 0;
        LDIT_TRY(dgrad(probe, dz, C, W16 + gl.w2 / 2, da1, M, F, EPI_GELU_BWD, S + sl.a1, zeros, stream, b1_fused ? PART(2) : nullptr));
        if (!b1_fused) LDIT_RUN(probe, LDIT_K_OTHER, launch_colsum_bf16(da1, M, F, F, PART(2), stream));
        LDIT_TRY(wgrad(probe, jobs, da1, F, S + sl.y2, C, GR(gl.w1), reinterpret_cast<float *>(ws + wm.slab[1]), F, C, M, zeros, stream));
        LDIT_TRY(dgrad(probe, da1, F, W16 + gl.w1 / 2, dy, M, C, EPI_F32, nullptr, zeros, stream));
        // ---- attention branch:  h_mid = h_in + rs1 lam1 (.) (attn(LN1(h_in)) Wo^T + bo) --------------------------------
        // its LayerScale / residual backward rides in the LayerNorm backward that produces dh (h_mid): one pass, dh read once
        LDIT_RUN(probe, LDIT_K_LAYERNORM, launch_layernorm_bwd_resid(dy, reinterpret_cast<const float *>(S + sl.h_mid), F32(pl.ln2_w), dh, M, C,
                                                                    cfg->ln_eps, PART(3), PART(4), S + sl.z1, F32(pl.lam1), rs1, dz, PART(5),
                                                                    PART(6), stream));
        LDIT_TRY(wgrad(probe, jobs, dz, C, S + sl.o, C, GR(gl.wo), reinterpret_cast<float *>(ws + wm.slab[2]), C, C, M, zeros, stream));
        LDIT_TRY(dgrad(probe, dz, C, W16 + gl.wo / 2, dob, M, C, EPI_BIAS, nullptr, zeros, stream));
        {
            const char *qkv = S + sl.qkv;
            LDIT_RUN(probe, LDIT_K_ATTENTION,
                     launch_attention_bwd_bf16(qkv, qkv + 2 * (size_t)C, qkv + 4 * (size_t)C, S + sl.o, dob,
                                               reinterpret_cast<const float *>(S + sl.lse), dqkv, dqkv + 2 * (size_t)C,
                                               dqkv + 4 * (size_t)C, batch, g.T, g.H, g.D, 3 * C, C, C, 3 * C, scale, stream));
        }
        LDIT_RUN(probe, LDIT_K_OTHER, launch_colsum_bf16(dqkv, M, 3 * C, 3 * C, PART(7), stream));
        LDIT_TRY(wgrad(probe, jobs, dqkv, 3 * C, S + sl.y1, C, GR(gl.wqkv), reinterpret_cast<float *>(ws + wm.slab[3]), 3 * C, C, M, zeros, stream));
        LDIT_TRY(dgrad(probe, dqkv, 3 * C, W16 + gl.wqkv / 2, dy, M, C, EPI_F32, nullptr, zeros, stream));
        {
            // dh becomes the gradient at hidden state st - 1: that state's tap gradient (if any) is summed in the same pass
            const float *t[8];
            const int n = tap_grads(st - 1, t);
            LDIT_RUN(probe, LDIT_K_LAYERNORM, launch_layernorm_bwd(dy, reinterpret_cast<const float *>(S + sl.h_in), F32(pl.ln1_w), dh, M, C,
                                                                  cfg->ln_eps, PART(8), PART(9), stream, n ? t[0] : nullptr));
            for (int i = 1; i < n; ++i) LDIT_RUN(probe, LDIT_K_OTHER, launch_add_inplace(dh, t[i], act, stream));
        }
        // ---- second stage of this layer's reductions: 4 wgrad slab sums (queued above) + 10 vectors, one launch ----------
        if (jobs.n + 12 > 16) LDIT_RUN(probe, LDIT_K_OTHER, launch_reduce_jobs(jobs, stream));
        jobs.add(PART(0), GR(gl.lam2), C, rows_tile, C);
        jobs.add(PART(1), GR(gl.b2), C, rows_tile, C);
        jobs.add(PART(2), GR(gl.b1), F, F % 256 == 0 ? gemm_bf16_tr_colsum_rows(M, F) : rows_tile, F);
        jobs.add(PART(3), GR(gl.ln2_w), C, rows_lnr, C);
        jobs.add(PART(4), GR(gl.ln2_b), C, rows_lnr, C);
        jobs.add(PART(5), GR(gl.lam1), C, rows_lnr, C);
        jobs.add(PART(6), GR(gl.bo), C, rows_lnr, C);
        // q | k | v bias gradient: BEiT has no key bias (TF:306) - its slot in the fused vector is written as a sum of ZERO partial rows
        jobs.add(PART(7), GR(gl.bqkv), C, rows_tile, 3 * C);
        jobs.add(PART(7), GR(gl.bqkv) + C, C, 0, 3 * C);
        jobs.add(PART(7) + 2 * C, GR(gl.bqkv) + 2 * C, C, rows_tile, 3 * C);
        jobs.add(PART(8), GR(gl.ln1_w), C, rows_ln, C);
        jobs.add(PART(9), GR(gl.ln1_b), C, rows_ln, C);
        LDIT_RUN(probe, LDIT_K_OTHER, launch_reduce_jobs(jobs, stream));
    }
    if (stage_lo == 0) {
        // ---- embeddings: h0[b, 0] = cls + pos[0];  h0[b, 1 + i] = patch_i . Wp^T + bp + pos[1 + i]  (TF:81-90, 168-172) ------
        const int Mq = batch * g.P;
        LDIT_RUN(probe, LDIT_K_OTHER, launch_embed_bwd_small(dh, GR(gm.pos), GR(gm.cls), GR(gm.patch_b), batch, g.T, C, stream));
        LDIT_RUN(probe, LDIT_K_OTHER, launch_rows_to_bf16(dh, dz, Mq, C, g.P, stream));                  // dE: patch rows of dh0, bf16
        // the bf16 im2col of the batch: the forward left it in the saved block when it ran the bf16 patch embedding
        const char *patches = S + sm.patches;
        if (g.Kp % 64 != 0) {
            LDIT_RUN(probe, LDIT_K_OTHER, launch_patches_rows(static_cast<const float *>(x), ws + wm.patches, batch, g.in_ch, cfg->img_h,
                                                             cfg->img_w, g.p, stream));
            patches = ws + wm.patches;
        }
        LDIT_TRY(wgrad(probe, jobs, dz, C, patches, g.Kp, GR(gm.patch_w), reinterpret_cast<float *>(ws + wm.slab[0]), C, g.Kp, Mq,
                       zeros, stream));
        LDIT_RUN(probe, LDIT_K_OTHER, launch_reduce_jobs(jobs, stream));
    }
    return LDIT_OK;
}

}  // namespace
}  // namespace ldit

using namespace ldit;

extern "C" {

size_t ldit_flat_param_bytes(const ldit_cfg *cfg)
{
    Geo g;
    if (geometry(cfg, g) != LDIT_OK) return 0;
    return packed_map(g, LDIT_F32).total;
}

int ldit_flat_param_layout(const ldit_cfg *cfg, int64_t *offsets, int32_t n)
{
    Geo g;
    LDIT_TRY(geometry(cfg, g));
    const int need = 4 + 14 * g.L + 1;
    if (!offsets || n < need) return fail(LDIT_EINVAL, "flat_param_layout: need room for %d offsets", need);
    const PackedMap m = packed_map(g, LDIT_F32);
    int k = 0;
    offsets[k++] = m.patch_w / 4; offsets[k++] = m.patch_b / 4; offsets[k++] = m.cls / 4; offsets[k++] = m.pos / 4;
    for (int l = 0; l < g.L; ++l) {
        const PackedLayer &p = m.layer[l];
        const size_t o[14] = {p.ln1_w, p.ln1_b, p.wqkv, p.bqkv, p.wo, p.bo, p.lam1, p.ln2_w, p.ln2_b, p.w1, p.b1, p.w2, p.b2, p.lam2};
        for (size_t v : o) offsets[k++] = (int64_t)(v / 4);
    }
    offsets[k++] = (int64_t)(m.total / 4);
    return LDIT_OK;
}

size_t ldit_train_saved_bytes(const ldit_cfg *cfg, int32_t batch)
{
    Geo g;
    if (batch <= 0 || train_geometry(cfg, g) != LDIT_OK) return 0;
    return saved_map(g, batch).total;
}

size_t ldit_train_workspace_bytes(const ldit_cfg *cfg, int32_t batch)
{
    Geo g;
    if (batch <= 0 || train_geometry(cfg, g) != LDIT_OK) return 0;
    return train_ws_map(g, batch).total;
}

size_t ldit_train_mirror_bytes(const ldit_cfg *cfg)
{
    Geo g;
    if (train_geometry(cfg, g) != LDIT_OK) return 0;
    return packed_map(g, LDIT_F32).total / 2;
}

int ldit_pack_train(const ldit_cfg *cfg, const void *flat_params, void *mirror, size_t mirror_bytes, ldit_stream stream)
{
    Geo g;
    LDIT_TRY(train_geometry(cfg, g));
    if (!flat_params || !mirror) return fail(LDIT_EINVAL, "pack_train: null pointer");
    if (!aligned16(flat_params) || !aligned16(mirror)) return fail(LDIT_EINVAL, "pack_train: pointers must be 16-byte aligned");
    const size_t n = packed_map(g, LDIT_F32).total / 4;
    if (mirror_bytes < 2 * n) return fail(LDIT_EWORKSPACE, "bf16 mirror %zu bytes < required %zu", mirror_bytes, 2 * n);
    // one pass: the whole flat block rounded to bf16, same element layout (matrices stay in nn.Linear's [out, in] layout: the
    // forward reads them K-contiguous, the dgrad reads the SAME copy reduction-major, the wgrad needs no weight)
    return launch_cvt_bf16(static_cast<const float *>(flat_params), mirror, n, static_cast<hipStream_t>(stream));
}

int ldit_vit_forward_train(const ldit_cfg *cfg, const void *packed, const void *flat_params, const void *x, int32_t batch,
                           void *const *tap_out, const void *drop_scales, void *saved, size_t saved_bytes, ldit_stream stream,
                           double *ms, int64_t *launches)
{
    Probe probe;
    probe.on = ms && launches;
    probe.stream = static_cast<hipStream_t>(stream);
    int rc = forward_train(cfg, packed, flat_params, x, batch, tap_out, drop_scales, saved, saved_bytes, probe.stream, probe);
    int rc2 = probe.collect(ms, launches);
    return rc != LDIT_OK ? rc : rc2;
}

int ldit_vit_backward(const ldit_cfg *cfg, const void *flat_params, const void *packed, const void *x, int32_t batch, void *const *dtaps,
                      const void *drop_scales, const void *saved, size_t saved_bytes, void *grads, size_t grads_bytes, void *workspace,
                      size_t workspace_bytes, int32_t stage_hi, int32_t stage_lo, ldit_stream stream, double *ms, int64_t *launches)
{
    Probe probe;
    probe.on = ms && launches;
    probe.stream = static_cast<hipStream_t>(stream);
    int rc = backward(cfg, flat_params, packed, x, batch, dtaps, drop_scales, saved, saved_bytes, grads, grads_bytes, workspace, workspace_bytes, stage_hi,
                      stage_lo, probe.stream, probe);
    int rc2 = probe.collect(ms, launches);
    return rc != LDIT_OK ? rc : rc2;
}

int ldit_adamw_step(void *params, const void *grads, void *exp_avg, void *exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int32_t step, float grad_scale, void *bf16_mirror, ldit_stream stream)
{
    if (n < 0) return fail(LDIT_EINVAL, "adamw: negative length");
    if (bf16_mirror && (reinterpret_cast<uintptr_t>(bf16_mirror) & 7u)) return fail(LDIT_EINVAL, "adamw: bf16 mirror misaligned");
    return launch_adamw(static_cast<float *>(params), static_cast<const float *>(grads), static_cast<float *>(exp_avg),
                        static_cast<float *>(exp_avg_sq), (size_t)n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, bf16_mirror,
                        static_cast<hipStream_t>(stream));
}

// ---- single kernels of the backward (unit parity tests) -------------------------------------------------------------------
int ldit_attention_fwd_lse_bf16(const void *Q, const void *K, const void *V, void *O, void *lse, int64_t B, int64_t N, int64_t H,
                                int64_t D, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float scale, ldit_stream stream)
{
    if (B <= 0 || N <= 0 || H <= 0) return fail(LDIT_EINVAL, "attention_fwd_lse: empty problem");
    if (B * N * 3 * H * D >= (1ll << 31)) return fail(LDIT_EUNSUPPORTED, "attention_fwd_lse: operand exceeds 2^31 elements");
    return launch_attention_bf16_lse(Q, K, V, O, static_cast<float *>(lse), (int)B, (int)N, (int)H, (int)D, (int)ldq, (int)ldk,
                                     (int)ldv, (int)ldo, scale, static_cast<hipStream_t>(stream));
}

int ldit_attention_bwd_bf16(const void *Q, const void *K, const void *V, const void *O, const void *dO, const void *lse, void *dQ,
                            void *dK, void *dV, int64_t B, int64_t N, int64_t H, int64_t D, int64_t ldqkv, int64_t ldo, int64_t lddo,
                            int64_t lddqkv, float scale, ldit_stream stream)
{
    if (B <= 0 || N <= 0 || H <= 0) return fail(LDIT_EINVAL, "attention_bwd: empty problem");
    if (B * N * (ldqkv > lddqkv ? ldqkv : lddqkv) >= (1ll << 31)) return fail(LDIT_EUNSUPPORTED, "attention_bwd: operand exceeds 2^31 elements");
    return launch_attention_bwd_bf16(Q, K, V, O, dO, static_cast<const float *>(lse), dQ, dK, dV, (int)B, (int)N, (int)H, (int)D,
                                     (int)ldqkv, (int)ldo, (int)lddo, (int)lddqkv, scale, static_cast<hipStream_t>(stream));
}

size_t ldit_layernorm_bwd_scratch_bytes(int64_t rows, int64_t C)
{
    if (rows <= 0 || C <= 0) return 0;
    return (size_t)2 * layernorm_bwd_blocks(rows) * (size_t)C * 4;
}

int ldit_layernorm_bwd_f32(const void *dy, const void *x, const void *gamma, void *dh, int64_t rows, int64_t C, float eps,
                           void *dgamma, void *dbeta, void *scratch, size_t scratch_bytes, ldit_stream stream_)
{
    if (rows <= 0 || C <= 0 || C > 4096) return fail(LDIT_EINVAL, "layernorm_bwd: bad shape");
    if (!dgamma || !dbeta || !scratch) return fail(LDIT_EINVAL, "layernorm_bwd: null operand");
    if (scratch_bytes < ldit_layernorm_bwd_scratch_bytes(rows, C)) return fail(LDIT_EWORKSPACE, "layernorm_bwd: scratch too small");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int blocks = layernorm_bwd_blocks(rows);
    float *pg = static_cast<float *>(scratch), *pb = pg + (size_t)blocks * C;
    LDIT_TRY(launch_layernorm_bwd(static_cast<const float *>(dy), static_cast<const float *>(x), static_cast<const float *>(gamma),
                                  static_cast<float *>(dh), rows, (int)C, eps, pg, pb, stream));
    ReduceJobs jobs;
    jobs.add(pg, static_cast<float *>(dgamma), C, blocks, C);
    jobs.add(pb, static_cast<float *>(dbeta), C, blocks, C);
    return launch_reduce_jobs(jobs, stream);
}

int ldit_linear_bf16_ex(const void *X, int64_t lda, const void *W, const void *bias, void *Y, int64_t ldy, int64_t M, int64_t N,
                        int64_t K, int32_t epilogue, const void *lam, const void *R, void *Y2, void *Ypre, const void *rowscale,
                        const void *aux, int64_t ldaux, int32_t splits, ldit_stream stream)
{
    if (M <= 0 || N <= 0 || K <= 0) return fail(LDIT_EINVAL, "linear_bf16_ex: empty problem");
    if (M * (ldy > lda ? ldy : lda) >= (1ll << 31) || N * K >= (1ll << 31)) return fail(LDIT_EUNSUPPORTED, "linear_bf16_ex: operand exceeds 2^31 elements");
    if (ldy < N || lda < K) return fail(LDIT_EINVAL, "linear_bf16_ex: bad leading dimension");
    if (!Y || !aligned16(Y) || (Y2 && !aligned16(Y2))) return fail(LDIT_EINVAL, "linear_bf16_ex: output null or misaligned");
    int epi;
    switch (epilogue) {
        case LDIT_EPI_BIAS: epi = EPI_BIAS; break;
        case LDIT_EPI_BIAS_GELU: epi = EPI_BIAS_GELU; break;
        case LDIT_EPI_SCALE_RESID: epi = EPI_SCALE_RESID; break;
        case LDIT_EPI_F32: epi = EPI_F32; break;
        case LDIT_EPI_GELU_BWD: epi = EPI_GELU_BWD; break;
        default: return fail(LDIT_EINVAL, "linear_bf16_ex: unknown epilogue %d", epilogue);
    }
    GemmExtra x{};
    x.Ypre = Ypre; x.rowscale = static_cast<const float *>(rowscale); x.aux = aux; x.ldaux = (int)ldaux; x.splits = splits < 1 ? 1 : splits;
    return launch_gemm_bf16_ex(X, (int)lda, W, static_cast<const float *>(bias), Y, (int)ldy, (int)M, (int)N, (int)K, epi,
                               static_cast<const float *>(lam), static_cast<const float *>(R), static_cast<float *>(Y2), x,
                               static_cast<hipStream_t>(stream));
}

int ldit_linear_bf16_tr(const void *A, int64_t lda, int32_t a_reduction_major, const void *W, int64_t ldw, void *Y, int64_t ldy,
                        int64_t M, int64_t N, int64_t K, int32_t epilogue, const void *aux, int32_t splits, const void *zeros,
                        ldit_stream stream)
{
    if (M <= 0 || N <= 0 || K <= 0) return fail(LDIT_EINVAL, "linear_bf16_tr: empty problem");
    if (M * ldy >= (1ll << 31) || K * (lda > ldw ? lda : ldw) >= (1ll << 31) || M * lda >= (1ll << 31))
        return fail(LDIT_EUNSUPPORTED, "linear_bf16_tr: operand exceeds 2^31 elements");
    if (ldy < N || ldw < N || lda < (a_reduction_major ? M : K)) return fail(LDIT_EINVAL, "linear_bf16_tr: bad leading dimension");
    if (!Y || !aligned16(Y)) return fail(LDIT_EINVAL, "linear_bf16_tr: output null or misaligned");
    int epi;
    switch (epilogue) {
        case LDIT_EPI_BIAS: epi = EPI_BIAS; break;
        case LDIT_EPI_F32: epi = EPI_F32; break;
        case LDIT_EPI_GELU_BWD: epi = EPI_GELU_BWD; break;
        default: return fail(LDIT_EINVAL, "linear_bf16_tr: epilogue %d not available", epilogue);
    }
    GemmExtra x{};
    x.aux = aux; x.ldaux = (int)N; x.splits = splits < 1 ? 1 : splits; x.zeros = zeros;
    return launch_gemm_bf16_tr(A, (int)lda, a_reduction_major != 0, W, (int)ldw, nullptr, Y, (int)ldy, (int)M, (int)N, (int)K, epi, x,
                               static_cast<hipStream_t>(stream));
}

int ldit_reduce_slabs_f32(const void *slabs, void *out, int64_t n, int32_t count, ldit_stream stream)
{
    if (!slabs || !out || n <= 0 || count <= 0) return fail(LDIT_EINVAL, "reduce_slabs: bad argument");
    ReduceJobs jobs;
    jobs.add(static_cast<const float *>(slabs), static_cast<float *>(out), n, count, n);
    return launch_reduce_jobs(jobs, static_cast<hipStream_t>(stream));
}

}  // extern "C"
