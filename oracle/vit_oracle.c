/*
 * vit_oracle.c - CPU restatement of the DiT / BEiT encoder forward.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the HIP path: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may build, load or call it.  Nothing under layoutdit_amd/ links or imports it.
 *
 * What it restates (the algorithm lives in a third-party dependency of the reference, HuggingFace
 * `transformers`, pinned 4.49.0 at ref uv.lock:1771-1772; call site ref
 * src/layoutdit/modeling/dit_backbone.py:47 `hs = self.dit(x).hidden_states`).  Line numbers below are the
 * installed transformers 5.15.0 copy ("TF:"), whose arithmetic is the same:
 *   TF:models/beit/modeling_beit.py:81,90      patch embedding = Conv2d(k=16,s=16) -> flatten -> transpose
 *   TF:models/beit/modeling_beit.py:168-172    [cls ; patches] + position_embeddings
 *   TF:models/beit/modeling_beit.py:305-307    q = Wq y + bq ; k = Wk y (NO bias) ; v = Wv y + bv
 *   TF:models/beit/modeling_beit.py:268-293    softmax(q k^T * D^-1/2) v  (softmax in fp32, no mask, no dropout)
 *   TF:models/beit/modeling_beit.py:426-442    pre-LN block with LayerScale lambda_1 / lambda_2 and residuals
 *   TF:models/beit/modeling_beit.py:352-357    fc1 -> GELU -> fc2 ; GELU = erf form, TF:activations.py:70-89
 *   TF:models/beit/modeling_beit.py:504-506    no final LayerNorm on the sequence (use_mean_pooling=True)
 *   TF:models/beit/configuration_beit.py:81    layer_norm_eps = 1e-12
 *   ref src/layoutdit/modeling/dit_backbone.py:33-34,50-61   taps d/3,d/2,2d/3,d ; CLS drop ; bilinear rescale
 *
 * Parity pin: tests/test_oracle_golden.py checks this file against the .npz vectors under tests/golden/, generated in the
 * build container by tests/golden/make_golden.py from transformers.BeitModel (CPU, fp32).
 *
 * Arithmetic: inputs/outputs fp32.  Default build accumulates every dot product, LayerNorm statistic and softmax
 * sum in double (a tighter reference than a fp32 CPU run).  -DORACLE_F32ACC accumulates in float: that build is
 * the "port" timed as bench.py's cpu_baseline.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef ORACLE_F32ACC
typedef float acc_t;
#define ACC_EXP expf
#define ACC_ERF erff
#define ACC_SQRT sqrtf
#else
typedef double acc_t;
#define ACC_EXP exp
#define ACC_ERF erf
#define ACC_SQRT sqrt
#endif

typedef struct {
    int32_t C, L, H, F, patch, in_ch, n_taps;
    int32_t taps[8];
    float ln_eps;
} oracle_cfg;

typedef struct {
    const float *ln1_w, *ln1_b, *wq, *bq, *wk, *wv, *bv, *wo, *bo, *lam1;
    const float *ln2_w, *ln2_b, *w1, *b1, *w2, *b2, *lam2;
} oracle_layer;

typedef struct {
    const float *patch_w, *patch_b, *cls, *pos; /* pos: [1+Gh*Gw, C], already resampled for the grid */
    const oracle_layer *layers;
} oracle_weights;

int oracle_abi_version(void) { return 1; }
int oracle_acc_bytes(void) { return (int)sizeof(acc_t); }

/* Y[M,N] = X[M,K] . W[N,K]^T (+ b[N]) ; W laid out as nn.Linear.weight. */
void oracle_linear(const float *X, const float *W, const float *b, int64_t M, int64_t K, int64_t N, float *Y)
{
    const int64_t MB = 4, NB = 4;
    const int64_t mblocks = (M + MB - 1) / MB, nblocks = (N + NB - 1) / NB;
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t mb = 0; mb < mblocks; ++mb) {
        for (int64_t nb = 0; nb < nblocks; ++nb) {
            const int64_t m0 = mb * MB, n0 = nb * NB;
            if (m0 + MB <= M && n0 + NB <= N) {
                const float *x0 = X + (m0 + 0) * K, *x1 = X + (m0 + 1) * K, *x2 = X + (m0 + 2) * K, *x3 = X + (m0 + 3) * K;
                const float *w0 = W + (n0 + 0) * K, *w1 = W + (n0 + 1) * K, *w2 = W + (n0 + 2) * K, *w3 = W + (n0 + 3) * K;
                acc_t a00 = 0, a01 = 0, a02 = 0, a03 = 0, a10 = 0, a11 = 0, a12 = 0, a13 = 0;
                acc_t a20 = 0, a21 = 0, a22 = 0, a23 = 0, a30 = 0, a31 = 0, a32 = 0, a33 = 0;
#pragma omp simd reduction(+ : a00, a01, a02, a03, a10, a11, a12, a13, a20, a21, a22, a23, a30, a31, a32, a33)
                for (int64_t k = 0; k < K; ++k) {
                    const acc_t xa = x0[k], xb = x1[k], xc = x2[k], xd = x3[k];
                    const acc_t wa = w0[k], wb = w1[k], wc = w2[k], wd = w3[k];
                    a00 += xa * wa; a01 += xa * wb; a02 += xa * wc; a03 += xa * wd;
                    a10 += xb * wa; a11 += xb * wb; a12 += xb * wc; a13 += xb * wd;
                    a20 += xc * wa; a21 += xc * wb; a22 += xc * wc; a23 += xc * wd;
                    a30 += xd * wa; a31 += xd * wb; a32 += xd * wc; a33 += xd * wd;
                }
                const acc_t acc[4][4] = {{a00, a01, a02, a03}, {a10, a11, a12, a13}, {a20, a21, a22, a23}, {a30, a31, a32, a33}};
                for (int i = 0; i < 4; ++i)
                    for (int j = 0; j < 4; ++j)
                        Y[(m0 + i) * N + n0 + j] = (float)(acc[i][j] + (b ? (acc_t)b[n0 + j] : (acc_t)0));
            } else {
                for (int64_t m = m0; m < m0 + MB && m < M; ++m)
                    for (int64_t n = n0; n < n0 + NB && n < N; ++n) {
                        acc_t a = 0;
                        for (int64_t k = 0; k < K; ++k) a += (acc_t)X[m * K + k] * (acc_t)W[n * K + k];
                        Y[m * N + n] = (float)(a + (b ? (acc_t)b[n] : (acc_t)0));
                    }
            }
        }
    }
}

/* Row LayerNorm, biased variance about the mean (two pass), y = (x-mu)*rsqrt(var+eps)*g + b. */
void oracle_layernorm(const float *X, const float *g, const float *b, int64_t rows, int64_t C, float eps, float *Y)
{
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r) {
        const float *x = X + r * C;
        acc_t mu = 0;
        for (int64_t c = 0; c < C; ++c) mu += x[c];
        mu /= (acc_t)C;
        acc_t var = 0;
        for (int64_t c = 0; c < C; ++c) { const acc_t d = (acc_t)x[c] - mu; var += d * d; }
        var /= (acc_t)C;
        const acc_t rstd = (acc_t)1 / ACC_SQRT(var + (acc_t)eps);
        for (int64_t c = 0; c < C; ++c) Y[r * C + c] = (float)(((acc_t)x[c] - mu) * rstd * (acc_t)g[c] + (acc_t)b[c]);
    }
}

/* erf GELU: 0.5 x (1 + erf(x / sqrt 2)). */
void oracle_gelu(const float *X, int64_t n, float *Y)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const acc_t x = X[i];
        Y[i] = (float)((acc_t)0.5 * x * ((acc_t)1 + ACC_ERF(x * (acc_t)0.70710678118654752440)));
    }
}

/* Multi-head attention on token-major tensors: Q,K,V,O are [B, N, H*D] with row stride ld (floats). */
void oracle_attention(const float *Q, const float *K, const float *V, int64_t B, int64_t N, int64_t H, int64_t D,
                      int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float scale, float *O)
{
#pragma omp parallel
    {
        acc_t *p = (acc_t *)malloc(sizeof(acc_t) * (size_t)N);
        acc_t *o = (acc_t *)malloc(sizeof(acc_t) * (size_t)D);
#pragma omp for collapse(2) schedule(static)
        for (int64_t b = 0; b < B; ++b) {
            for (int64_t h = 0; h < H; ++h) {
                for (int64_t i = 0; i < N; ++i) {
                    const float *q = Q + (b * N + i) * ldq + h * D;
                    acc_t mx = -INFINITY;
                    for (int64_t j = 0; j < N; ++j) {
                        const float *k = K + (b * N + j) * ldk + h * D;
                        acc_t s = 0;
                        for (int64_t d = 0; d < D; ++d) s += (acc_t)q[d] * (acc_t)k[d];
                        s *= (acc_t)scale;
                        p[j] = s;
                        if (s > mx) mx = s;
                    }
                    acc_t sum = 0;
                    for (int64_t j = 0; j < N; ++j) { p[j] = ACC_EXP(p[j] - mx); sum += p[j]; }
                    for (int64_t d = 0; d < D; ++d) o[d] = 0;
                    for (int64_t j = 0; j < N; ++j) {
                        const float *v = V + (b * N + j) * ldv + h * D;
                        const acc_t pj = p[j];
                        for (int64_t d = 0; d < D; ++d) o[d] += pj * (acc_t)v[d];
                    }
                    float *out = O + (b * N + i) * ldo + h * D;
                    for (int64_t d = 0; d < D; ++d) out[d] = (float)(o[d] / sum);
                }
            }
        }
        free(p);
        free(o);
    }
}

/* Patch embedding + [cls ; patches] + pos.  x: [B,in_ch,Himg,Wimg] NCHW ; Wp: [C,in_ch,p,p] ; out: [B,1+P,C]. */
void oracle_embed(const float *x, const float *Wp, const float *bp, const float *cls, const float *pos, int64_t B,
                  int64_t in_ch, int64_t Himg, int64_t Wimg, int64_t p, int64_t C, float *out)
{
    const int64_t Gh = Himg / p, Gw = Wimg / p, P = Gh * Gw, N = P + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        for (int64_t t = 0; t < N; ++t) {
            float *o = out + (b * N + t) * C;
            if (t == 0) {
                for (int64_t c = 0; c < C; ++c) o[c] = (float)((acc_t)cls[c] + (acc_t)pos[c]);
                continue;
            }
            const int64_t gy = (t - 1) / Gw, gx = (t - 1) % Gw;
            for (int64_t c = 0; c < C; ++c) {
                acc_t a = 0;
                for (int64_t ch = 0; ch < in_ch; ++ch)
                    for (int64_t dy = 0; dy < p; ++dy) {
                        const float *xr = x + ((b * in_ch + ch) * Himg + gy * p + dy) * Wimg + gx * p;
                        const float *wr = Wp + ((c * in_ch + ch) * p + dy) * p;
                        for (int64_t dx = 0; dx < p; ++dx) a += (acc_t)xr[dx] * (acc_t)wr[dx];
                    }
                o[c] = (float)((a + (acc_t)bp[c]) + (acc_t)pos[t * C + c]);
            }
        }
    }
}

/* h <- h + lam (.) y   (TF:modeling_beit.py:432-434, 440-442; drop_path is the identity in eval). */
static void scale_residual(float *h, const float *y, const float *lam, int64_t rows, int64_t C)
{
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t c = 0; c < C; ++c) h[r * C + c] = (float)((acc_t)lam[c] * (acc_t)y[r * C + c] + (acc_t)h[r * C + c]);
}

/*
 * Whole encoder.  hidden_all (optional): (L+1) x [B,N,C] = BeitModel(...).hidden_states.  tap_out[i] (optional):
 * [B,N,C] copy of hidden state cfg->taps[i].  Returns 0, or -1 on a bad argument / allocation failure.
 */
int oracle_vit_forward(const oracle_cfg *cfg, const oracle_weights *w, const float *x, int64_t B, int64_t Himg,
                       int64_t Wimg, float *hidden_all, float *const *tap_out)
{
    const int64_t C = cfg->C, F = cfg->F, H = cfg->H, p = cfg->patch;
    if (C % H || Himg % p || Wimg % p || cfg->n_taps > 8) return -1;
    const int64_t D = C / H, N = (Himg / p) * (Wimg / p) + 1, M = B * N;
    float *h = (float *)malloc(sizeof(float) * (size_t)(M * C));
    float *y = (float *)malloc(sizeof(float) * (size_t)(M * C));
    float *qkv = (float *)malloc(sizeof(float) * (size_t)(M * 3 * C));
    float *a = (float *)malloc(sizeof(float) * (size_t)(M * C));
    float *f = (float *)malloc(sizeof(float) * (size_t)(M * F));
    if (!h || !y || !qkv || !a || !f) { free(h); free(y); free(qkv); free(a); free(f); return -1; }

    oracle_embed(x, w->patch_w, w->patch_b, w->cls, w->pos, B, cfg->in_ch, Himg, Wimg, p, C, h);
    if (hidden_all) memcpy(hidden_all, h, sizeof(float) * (size_t)(M * C));
    for (int t = 0; t < cfg->n_taps; ++t)
        if (tap_out && tap_out[t] && cfg->taps[t] == 0) memcpy(tap_out[t], h, sizeof(float) * (size_t)(M * C));

    const float scale = (float)(1.0 / sqrt((double)D));
    for (int l = 0; l < cfg->L; ++l) {
        const oracle_layer *lw = &w->layers[l];
        oracle_layernorm(h, lw->ln1_w, lw->ln1_b, M, C, cfg->ln_eps, y);
        oracle_linear(y, lw->wq, lw->bq, M, C, C, qkv);
        oracle_linear(y, lw->wk, NULL, M, C, C, qkv + M * C);
        oracle_linear(y, lw->wv, lw->bv, M, C, C, qkv + 2 * M * C);
        oracle_attention(qkv, qkv + M * C, qkv + 2 * M * C, B, N, H, D, C, C, C, C, scale, a);
        oracle_linear(a, lw->wo, lw->bo, M, C, C, y);
        scale_residual(h, y, lw->lam1, M, C);
        oracle_layernorm(h, lw->ln2_w, lw->ln2_b, M, C, cfg->ln_eps, y);
        oracle_linear(y, lw->w1, lw->b1, M, C, F, f);
        oracle_gelu(f, M * F, f);
        oracle_linear(f, lw->w2, lw->b2, M, F, C, y);
        scale_residual(h, y, lw->lam2, M, C);
        if (hidden_all) memcpy(hidden_all + (size_t)(l + 1) * (size_t)(M * C), h, sizeof(float) * (size_t)(M * C));
        for (int t = 0; t < cfg->n_taps; ++t)
            if (tap_out && tap_out[t] && cfg->taps[t] == l + 1) memcpy(tap_out[t], h, sizeof(float) * (size_t)(M * C));
    }
    free(h); free(y); free(qkv); free(a); free(f);
    return 0;
}

/*
 * Tap post-processing of DiTBackbone.forward (ref src/layoutdit/modeling/dit_backbone.py:50-61):
 * drop CLS, view tokens as a [C, Gh, Gw] map, bilinear rescale by `scale` (align_corners=False; the
 * F.interpolate(scale_factor=s) rule: src = (dst + 0.5) / s - 0.5 clamped at 0, neighbour clamped at size-1).
 * tap: [B, 1+Gh*Gw, C] ; out: [B, C, Oh, Ow] NCHW contiguous with Oh = floor(Gh*scale), Ow = floor(Gw*scale).
 */
void oracle_tap_to_map(const float *tap, int64_t B, int64_t Gh, int64_t Gw, int64_t C, double scale, float *out)
{
    const int64_t Oh = (int64_t)floor((double)Gh * scale), Ow = (int64_t)floor((double)Gw * scale), N = Gh * Gw + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        for (int64_t c = 0; c < C; ++c) {
            for (int64_t oy = 0; oy < Oh; ++oy) {
                double sy = ((double)oy + 0.5) / scale - 0.5;
                if (sy < 0) sy = 0;
                int64_t iy0 = (int64_t)sy;
                if (iy0 > Gh - 1) iy0 = Gh - 1;
                const int64_t iy1 = iy0 + (iy0 < Gh - 1 ? 1 : 0);
                const double ly = sy - (double)iy0, hy = 1.0 - ly;
                for (int64_t ox = 0; ox < Ow; ++ox) {
                    double sx = ((double)ox + 0.5) / scale - 0.5;
                    if (sx < 0) sx = 0;
                    int64_t ix0 = (int64_t)sx;
                    if (ix0 > Gw - 1) ix0 = Gw - 1;
                    const int64_t ix1 = ix0 + (ix0 < Gw - 1 ? 1 : 0);
                    const double lx = sx - (double)ix0, hx = 1.0 - lx;
#define TAPV(yy, xx) ((double)tap[(b * N + 1 + (yy) * Gw + (xx)) * C + c])
                    const double v = hy * (hx * TAPV(iy0, ix0) + lx * TAPV(iy0, ix1)) + ly * (hx * TAPV(iy1, ix0) + lx * TAPV(iy1, ix1));
#undef TAPV
                    out[((b * C + c) * Oh + oy) * Ow + ox] = (float)v;
                }
            }
        }
    }
}

/*
 * Detector input transform (ref src/layoutdit/modeling/model.py:50-54; torchvision GeneralizedRCNNTransform, pinned
 * 0.19.0, source not in /root/reference -> restated from F.interpolate(size=..., mode="bilinear",
 * align_corners=False) semantics: src = (dst + 0.5) * in/out - 0.5 clamped at 0).  img [in_ch,h,w] -> out [in_ch,oh,ow].
 */
void oracle_preprocess(const float *img, int64_t in_ch, int64_t h, int64_t w, double mean, double std, int64_t oh,
                       int64_t ow, float *out)
{
    const double sch = (double)h / (double)oh, scw = (double)w / (double)ow;
    for (int64_t ch = 0; ch < in_ch; ++ch)
        for (int64_t oy = 0; oy < oh; ++oy) {
            double sy = ((double)oy + 0.5) * sch - 0.5;
            if (sy < 0) sy = 0;
            int64_t iy0 = (int64_t)sy;
            if (iy0 > h - 1) iy0 = h - 1;
            const int64_t iy1 = iy0 + (iy0 < h - 1 ? 1 : 0);
            const double ly = sy - (double)iy0, hy = 1.0 - ly;
            for (int64_t ox = 0; ox < ow; ++ox) {
                double sx = ((double)ox + 0.5) * scw - 0.5;
                if (sx < 0) sx = 0;
                int64_t ix0 = (int64_t)sx;
                if (ix0 > w - 1) ix0 = w - 1;
                const int64_t ix1 = ix0 + (ix0 < w - 1 ? 1 : 0);
                const double lx = sx - (double)ix0, hx = 1.0 - lx;
                const float *p = img + ch * h * w;
                const double v = hy * (hx * p[iy0 * w + ix0] + lx * p[iy0 * w + ix1]) + ly * (hx * p[iy1 * w + ix0] + lx * p[iy1 * w + ix1]);
                out[(ch * oh + oy) * ow + ox] = (float)((v - mean) / std);
            }
        }
}
