// Backward of the fused self-attention (train step, BASELINE configs[2]):  given Q, K, V, O (bf16, as the forward left
// them), dO (bf16) and the forward's log2-domain log-sum-exp L2[q] = log2 sum_k exp2(c s_qk), c = scale log2 e, produce
// dQ, dK, dV (bf16) per (image, head).  TF:models/beit/modeling_beit.py:268-293 differentiated:
//     P = softmax(scale Q K^T)      dV = P^T dO      dP = dO V^T      dS = P (.) (dP - delta) scale,  delta_q = sum_d dO O
//     dQ = dS K                     dK = dS^T Q
// Probabilities are recomputed from L2 (p = exp2(c s - L2), no row maximum needed), never stored.
//
// One 512-thread workgroup per (image, head) for N <= 256 tokens (the detector runs at 224 x 224 -> 197; longer sequences take
// the blocked form at the end of this file), so Q, K, V and dO
// of the head live in LDS for the whole kernel (row-major 128-B rows, 16-B chunk XOR-swizzled with (row>>1)&7: b128 row
// reads for the operands that are consumed row-wise, ds_read_b64_tr_b16 for the ones consumed column-wise - one image
// serves both).  Rows past N are ZERO in all four images, which makes every padded contribution vanish without masks:
// a padded key has K = V = 0 (no term in dQ, and its own dK / dV rows are never stored), a padded query has Q = dO = 0,
// delta = 0, L2 = 0 (p = 1, finite) and adds nothing to dK / dV.
//
// Two phases, every product on v_mfma_f32_32x32x16_bf16 with the accumulator of the first product reused as the B operand
// of the second one (guide section 3, "An accumulator tile as the next MFMA's operand"; same trick as the forward):
//   phase 1, wave = one block of 32 keys, loop over query blocks:   S[q, key] and dP[q, key] with the key on the lane,
//            dV^T[d, key] += dO^T P,  dK^T[d, key] += Q^T dS     (A operands = transposing reads of the dO / Q images)
//   phase 2, wave = one block of 32 queries, loop over key blocks:  S^T[key, q], dP^T[key, q] with the query on the lane,
//            dQ^T[d, q] += K^T dS^T                              (A operand = transposing reads of the K image)
// S and dP are therefore computed twice (28 MFMAs per (query block, key block) pair instead of 20) - the price for
// keeping every sum inside one wave: no atomics, no cross-wave reduction, bit-reproducible.  Attention backward is 4 % of
// the train step's FLOPs.
#include "ldit_common.h"

namespace ldit {

namespace {

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int ROWB = 128;      // bytes per image row (64 bf16)

__device__ __forceinline__ int img_off(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }

// 8 consecutive elements (one 16-B chunk) of image row `row`
__device__ __forceinline__ bf16x8 row_frag(const char *img, int row, int chunk)
{
    return *reinterpret_cast<const bf16x8 *>(img + img_off(row, chunk));
}

// A operand of a product that sums over the image's ROW index: lane (c = lane&31, h) gets, for element jj,
// img[row0 + 8 (jj>>2) + 4 h + (jj&3)][col0 + c] - two transposing reads (4 rows x 16 columns per 16-lane group)
__device__ __forceinline__ bf16x8 col_frag(const char *img, int row0, int col0, int lane)
{
    const int grp = lane >> 4, i16 = lane & 15;
    const int col = col0 + 16 * (grp & 1) + 4 * (i16 & 3);
    const int r = row0 + 4 * (grp >> 1) + (i16 >> 2);
    union { s16x4 v[2]; bf16x8 f; } u;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int row = r + 8 * t;
        const char *a = img + row * ROWB + (((col >> 3) ^ ((row >> 1) & 7)) << 4) + ((col & 7) << 1);
        u.v[t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)a);
    }
    return u.f;
}

__device__ __forceinline__ bf16x8 pack8(const f32x16 &v, int s)
{
    bf16x8 o;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) o[jj] = (bf16_t)v[8 * s + jj];
    return o;
}

__global__ void __launch_bounds__(512, 2) attention_bwd_bf16(const bf16_t *__restrict__ Q, const bf16_t *__restrict__ K,
                                                             const bf16_t *__restrict__ V, const bf16_t *__restrict__ O,
                                                             const bf16_t *__restrict__ dO, const float *__restrict__ lse,
                                                             bf16_t *__restrict__ dQ, bf16_t *__restrict__ dK,
                                                             bf16_t *__restrict__ dV, int N, int H, int ldqkv, int ldo,
                                                             int lddo, int lddqkv, float scale)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nb = (N + 31) >> 5, NP = nb * 32;
    char *Qs = smem, *Ks = Qs + NP * ROWB, *Vs = Ks + NP * ROWB, *Gs = Vs + NP * ROWB;      // Gs = dO image
    float *L2s = reinterpret_cast<float *>(Gs + NP * ROWB), *dlt = L2s + NP;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c32 = lane & 31, h = lane >> 5;
    const int head = blockIdx.x % H, b = blockIdx.x / H;
    const size_t tok0 = (size_t)b * N;

    // ---- stage the head: 16-B chunk (row, ch) per thread and pass; delta = rowsum(dO . O) on the way.  All passes' loads
    //      (up to 4 x 5 chunks) are issued before the first LDS store: one memory round trip for the head, not one per pass.
    {
        constexpr int PASSES = 4;                       // N <= 256 rows, 64 rows per pass
        bf16x8 q[PASSES], k[PASSES], v[PASSES], g[PASSES], o[PASSES];
        const int ch = tid & 7;
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int row = 64 * ps + (tid >> 3);
            q[ps] = bf16x8{}; k[ps] = bf16x8{}; v[ps] = bf16x8{}; g[ps] = bf16x8{}; o[ps] = bf16x8{};
            if (row < N) {
                const size_t t = tok0 + row;
                q[ps] = *reinterpret_cast<const bf16x8 *>(Q + t * ldqkv + head * 64 + 8 * ch);
                k[ps] = *reinterpret_cast<const bf16x8 *>(K + t * ldqkv + head * 64 + 8 * ch);
                v[ps] = *reinterpret_cast<const bf16x8 *>(V + t * ldqkv + head * 64 + 8 * ch);
                g[ps] = *reinterpret_cast<const bf16x8 *>(dO + t * lddo + head * 64 + 8 * ch);
                o[ps] = *reinterpret_cast<const bf16x8 *>(O + t * ldo + head * 64 + 8 * ch);
            }
        }
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int row = 64 * ps + (tid >> 3);
            float d = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) d = __builtin_fmaf((float)g[ps][e], (float)o[ps][e], d);
            d += __shfl_xor(d, 1, 64);
            d += __shfl_xor(d, 2, 64);
            d += __shfl_xor(d, 4, 64);
            if (row < NP) {
                const int off = img_off(row, ch);
                *reinterpret_cast<bf16x8 *>(Qs + off) = q[ps];
                *reinterpret_cast<bf16x8 *>(Ks + off) = k[ps];
                *reinterpret_cast<bf16x8 *>(Vs + off) = v[ps];
                *reinterpret_cast<bf16x8 *>(Gs + off) = g[ps];
                if (ch == 0) {
                    dlt[row] = d;
                    L2s[row] = row < N ? lse[((size_t)b * H + head) * N + row] : 0.0f;
                }
            }
        }
    }
    __syncthreads();
    if (wave >= nb) return;            // wave-uniform: the transposing reads below need every lane of a wave active

    const float c = scale * 1.44269504088896340736f;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // ================= phase 1: this wave's 32 keys; dK^T, dV^T [d = register row, key = lane] ========================
    {
        const int key0 = wave * 32;
        bf16x8 kf[4], vf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kf[s] = row_frag(Ks, key0 + c32, 2 * s + h);
            vf[s] = row_frag(Vs, key0 + c32, 2 * s + h);
        }
        f32x16 dk[2] = {zero16, zero16}, dv[2] = {zero16, zero16};
        for (int i = 0; i < nb; ++i) {
            const int q0 = i * 32;
            f32x16 sc = zero16, dp = zero16;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Qs, q0 + c32, 2 * s + h), kf[s], sc, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Gs, q0 + c32, 2 * s + h), vf[s], dp, 0, 0, 0);
            }
            // register r <-> query q0 + (r&3) + 8 (r>>2) + 4 h
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 l2 = *reinterpret_cast<const f32x4 *>(L2s + q0 + 8 * g + 4 * h);
                const f32x4 de = *reinterpret_cast<const f32x4 *>(dlt + q0 + 8 * g + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[4 * g + e], c, -l2[e]));
                    sc[4 * g + e] = p;
                    dp[4 * g + e] = p * (dp[4 * g + e] - de[e]) * scale;
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 pb = pack8(sc, s), db = pack8(dp, s);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(col_frag(Gs, q0 + 16 * s, 32 * dt, lane), pb, dv[dt], 0, 0, 0);
                    dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(col_frag(Qs, q0 + 16 * s, 32 * dt, lane), db, dk[dt], 0, 0, 0);
                }
            }
        }
        const int key = key0 + c32;
        if (key < N) {
            bf16_t *pk = dK + (tok0 + key) * lddqkv + head * 64 + 4 * h;
            bf16_t *pv = dV + (tok0 + key) * lddqkv + head * 64 + 4 * h;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const bf16x4 a = {(bf16_t)dk[dt][4 * g + 0], (bf16_t)dk[dt][4 * g + 1], (bf16_t)dk[dt][4 * g + 2], (bf16_t)dk[dt][4 * g + 3]};
                    const bf16x4 w = {(bf16_t)dv[dt][4 * g + 0], (bf16_t)dv[dt][4 * g + 1], (bf16_t)dv[dt][4 * g + 2], (bf16_t)dv[dt][4 * g + 3]};
                    *reinterpret_cast<bf16x4 *>(pk + 32 * dt + 8 * g) = a;
                    *reinterpret_cast<bf16x4 *>(pv + 32 * dt + 8 * g) = w;
                }
        }
    }

    // ================= phase 2: this wave's 32 queries; dQ^T [d = register row, query = lane] ==========================
    {
        const int q0 = wave * 32;
        bf16x8 qf[4], gf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qf[s] = row_frag(Qs, q0 + c32, 2 * s + h);
            gf[s] = row_frag(Gs, q0 + c32, 2 * s + h);
        }
        const float l2 = L2s[q0 + c32], de = dlt[q0 + c32];
        f32x16 dq[2] = {zero16, zero16};
        for (int j = 0; j < nb; ++j) {
            const int key0 = j * 32;
            f32x16 sc = zero16, dp = zero16;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Ks, key0 + c32, 2 * s + h), qf[s], sc, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Vs, key0 + c32, 2 * s + h), gf[s], dp, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[r], c, -l2));
                dp[r] = p * (dp[r] - de) * scale;
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 db = pack8(dp, s);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(col_frag(Ks, key0 + 16 * s, 32 * dt, lane), db, dq[dt], 0, 0, 0);
            }
        }
        const int q = q0 + c32;
        if (q < N) {
            bf16_t *pq = dQ + (tok0 + q) * lddqkv + head * 64 + 4 * h;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const bf16x4 a = {(bf16_t)dq[dt][4 * g + 0], (bf16_t)dq[dt][4 * g + 1], (bf16_t)dq[dt][4 * g + 2], (bf16_t)dq[dt][4 * g + 3]};
                    *reinterpret_cast<bf16x4 *>(pq + 32 * dt + 8 * g) = a;
                }
        }
    }
}

// ---- N > 256 tokens (round 3: ViT-L/16 at 512 x 512 = 1025 tokens trains too) ---------------------------------------------
// The same two phases, BLOCKED: the sequence is cut into blocks of 256 rows and the four LDS images of the kernel above (Q, K,
// V, dO, 32 KB each, + L2 and delta) hold ONE resident block and ONE streamed block:
//   ROLE 0 (one workgroup per (image, head, KEY block)):   K, V of the block resident; every QUERY block is streamed through
//           the Q / dO / L2 / delta images and phase 1 runs on it; dK^T, dV^T stay in the waves' registers across ALL query
//           blocks and are stored once - no sum crosses a workgroup;
//   ROLE 1 (one workgroup per (image, head, QUERY block)):  Q, dO, L2, delta resident; every KEY block is streamed through the
//           K / V images and phase 2 runs on it; dQ^T stays in registers across all key blocks.
// S and dP are recomputed per role exactly as in the one-block kernel (same products, same zero-padding argument per block),
// so every output element is a fixed-order sum inside one wave: no atomics, bit-reproducible.  Staging is synchronous
// (load, barrier, multiply, barrier): this path exists so that long sequences CAN train; BASELINE configs[2] (197 tokens)
// runs the one-block kernel.
template <int ROLE>
__global__ void __launch_bounds__(512, 2) attention_bwd_bf16_blk(const bf16_t *__restrict__ Q, const bf16_t *__restrict__ K,
                                                                 const bf16_t *__restrict__ V, const bf16_t *__restrict__ O,
                                                                 const bf16_t *__restrict__ dO, const float *__restrict__ lse,
                                                                 bf16_t *__restrict__ dQ, bf16_t *__restrict__ dK,
                                                                 bf16_t *__restrict__ dV, int N, int H, int nblk, int ldqkv, int ldo,
                                                                 int lddo, int lddqkv, float scale)
{
    constexpr int BR = 256;                               // rows per block
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *Qs = smem, *Ks = Qs + BR * ROWB, *Vs = Ks + BR * ROWB, *Gs = Vs + BR * ROWB;      // Gs = dO image
    float *L2s = reinterpret_cast<float *>(Gs + BR * ROWB), *dlt = L2s + BR;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c32 = lane & 31, h = lane >> 5;
    const int blk = blockIdx.x % nblk, bh = blockIdx.x / nblk, head = bh % H, b = bh / H;
    const size_t tok0 = (size_t)b * N;
    const int ch = tid & 7;

    // stage the key side (K, V images) / the query side (Q, dO images, L2, delta) of rows [r0, r0 + 256); rows past N are zero
    auto stage_keys = [&](int r0) {
        bf16x8 k[4], v[4];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int row = r0 + 64 * ps + (tid >> 3);
            k[ps] = bf16x8{}; v[ps] = bf16x8{};
            if (row < N) {
                const size_t t = tok0 + row;
                k[ps] = *reinterpret_cast<const bf16x8 *>(K + t * ldqkv + head * 64 + 8 * ch);
                v[ps] = *reinterpret_cast<const bf16x8 *>(V + t * ldqkv + head * 64 + 8 * ch);
            }
        }
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int off = img_off(64 * ps + (tid >> 3), ch);
            *reinterpret_cast<bf16x8 *>(Ks + off) = k[ps];
            *reinterpret_cast<bf16x8 *>(Vs + off) = v[ps];
        }
    };
    auto stage_queries = [&](int r0) {
        bf16x8 q[4], g[4], o[4];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int row = r0 + 64 * ps + (tid >> 3);
            q[ps] = bf16x8{}; g[ps] = bf16x8{}; o[ps] = bf16x8{};
            if (row < N) {
                const size_t t = tok0 + row;
                q[ps] = *reinterpret_cast<const bf16x8 *>(Q + t * ldqkv + head * 64 + 8 * ch);
                g[ps] = *reinterpret_cast<const bf16x8 *>(dO + t * lddo + head * 64 + 8 * ch);
                o[ps] = *reinterpret_cast<const bf16x8 *>(O + t * ldo + head * 64 + 8 * ch);
            }
        }
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int lrow = 64 * ps + (tid >> 3), row = r0 + lrow;
            float d = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) d = __builtin_fmaf((float)g[ps][e], (float)o[ps][e], d);
            d += __shfl_xor(d, 1, 64);
            d += __shfl_xor(d, 2, 64);
            d += __shfl_xor(d, 4, 64);
            const int off = img_off(lrow, ch);
            *reinterpret_cast<bf16x8 *>(Qs + off) = q[ps];
            *reinterpret_cast<bf16x8 *>(Gs + off) = g[ps];
            if (ch == 0) {
                dlt[lrow] = d;
                L2s[lrow] = row < N ? lse[((size_t)b * H + head) * N + row] : 0.0f;
            }
        }
    };

    const float c = scale * 1.44269504088896340736f;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int r_res = blk * BR;                                  // first row of the resident block
    const bool live = r_res + wave * 32 < N;                     // wave-uniform: this wave's 32 resident rows hold a token

    if (ROLE == 0) {
        stage_keys(r_res);
        f32x16 dk[2] = {zero16, zero16}, dv[2] = {zero16, zero16};
        const int key0 = wave * 32;
        for (int qb = 0; qb < nblk; ++qb) {
            if (qb) __syncthreads();                             // the previous query block is consumed
            stage_queries(qb * BR);
            __syncthreads();
            if (!live) continue;                                 // (every wave still reaches both barriers)
            bf16x8 kf[4], vf[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                kf[s] = row_frag(Ks, key0 + c32, 2 * s + h);
                vf[s] = row_frag(Vs, key0 + c32, 2 * s + h);
            }
            const int nq = (N - qb * BR) < BR ? (N - qb * BR + 31) >> 5 : BR / 32;      // 32-row sub-blocks holding a query
            for (int i = 0; i < nq; ++i) {
                const int q0 = i * 32;
                f32x16 sc = zero16, dp = zero16;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Qs, q0 + c32, 2 * s + h), kf[s], sc, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Gs, q0 + c32, 2 * s + h), vf[s], dp, 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 l2 = *reinterpret_cast<const f32x4 *>(L2s + q0 + 8 * g + 4 * h);
                    const f32x4 de = *reinterpret_cast<const f32x4 *>(dlt + q0 + 8 * g + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[4 * g + e], c, -l2[e]));
                        sc[4 * g + e] = p;
                        dp[4 * g + e] = p * (dp[4 * g + e] - de[e]) * scale;
                    }
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 pb = pack8(sc, s), db = pack8(dp, s);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(col_frag(Gs, q0 + 16 * s, 32 * dt, lane), pb, dv[dt], 0, 0, 0);
                        dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(col_frag(Qs, q0 + 16 * s, 32 * dt, lane), db, dk[dt], 0, 0, 0);
                    }
                }
            }
        }
        const int key = r_res + key0 + c32;
        if (live && key < N) {
            bf16_t *pk = dK + (tok0 + key) * lddqkv + head * 64 + 4 * h;
            bf16_t *pv = dV + (tok0 + key) * lddqkv + head * 64 + 4 * h;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const bf16x4 a = {(bf16_t)dk[dt][4 * g + 0], (bf16_t)dk[dt][4 * g + 1], (bf16_t)dk[dt][4 * g + 2], (bf16_t)dk[dt][4 * g + 3]};
                    const bf16x4 w = {(bf16_t)dv[dt][4 * g + 0], (bf16_t)dv[dt][4 * g + 1], (bf16_t)dv[dt][4 * g + 2], (bf16_t)dv[dt][4 * g + 3]};
                    *reinterpret_cast<bf16x4 *>(pk + 32 * dt + 8 * g) = a;
                    *reinterpret_cast<bf16x4 *>(pv + 32 * dt + 8 * g) = w;
                }
        }
    } else {
        stage_queries(r_res);
        f32x16 dq[2] = {zero16, zero16};
        const int q0 = wave * 32;
        for (int kb = 0; kb < nblk; ++kb) {
            if (kb) __syncthreads();
            stage_keys(kb * BR);
            __syncthreads();
            if (!live) continue;
            bf16x8 qf[4], gf[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                qf[s] = row_frag(Qs, q0 + c32, 2 * s + h);
                gf[s] = row_frag(Gs, q0 + c32, 2 * s + h);
            }
            const float l2 = L2s[q0 + c32], de = dlt[q0 + c32];
            const int nk = (N - kb * BR) < BR ? (N - kb * BR + 31) >> 5 : BR / 32;
            for (int j = 0; j < nk; ++j) {
                const int key0 = j * 32;
                f32x16 sc = zero16, dp = zero16;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Ks, key0 + c32, 2 * s + h), qf[s], sc, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Vs, key0 + c32, 2 * s + h), gf[s], dp, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[r], c, -l2));
                    dp[r] = p * (dp[r] - de) * scale;
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 db = pack8(dp, s);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
                        dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(col_frag(Ks, key0 + 16 * s, 32 * dt, lane), db, dq[dt], 0, 0, 0);
                }
            }
        }
        const int q = r_res + q0 + c32;
        if (live && q < N) {
            bf16_t *pq = dQ + (tok0 + q) * lddqkv + head * 64 + 4 * h;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const bf16x4 a = {(bf16_t)dq[dt][4 * g + 0], (bf16_t)dq[dt][4 * g + 1], (bf16_t)dq[dt][4 * g + 2], (bf16_t)dq[dt][4 * g + 3]};
                    *reinterpret_cast<bf16x4 *>(pq + 32 * dt + 8 * g) = a;
                }
        }
    }
}

}  // namespace

int launch_attention_bwd_bf16(const void *Q, const void *K, const void *V, const void *O, const void *dO, const float *lse,
                              void *dQ, void *dK, void *dV, int B, int N, int H, int D, int ldqkv, int ldo, int lddo, int lddqkv,
                              float scale, hipStream_t stream)
{
    if (B <= 0 || N <= 0 || H <= 0) return fail(LDIT_EINVAL, "attention_bwd: empty problem");
    if (D != 64) return fail(LDIT_EUNSUPPORTED, "attention_bwd: head_dim=%d, only 64 is implemented", D);
    if (!Q || !K || !V || !O || !dO || !lse || !dQ || !dK || !dV) return fail(LDIT_EINVAL, "attention_bwd: null operand");
    if ((ldqkv | ldo | lddo) & 7 || (lddqkv & 3)) return fail(LDIT_EINVAL, "attention_bwd: row strides must be multiples of 8 (in) / 4 (out)");
    if (!aligned16(Q) || !aligned16(K) || !aligned16(V) || !aligned16(O) || !aligned16(dO) ||
        ((reinterpret_cast<uintptr_t>(dQ) | reinterpret_cast<uintptr_t>(dK) | reinterpret_cast<uintptr_t>(dV)) & 7u))
        return fail(LDIT_EINVAL, "attention_bwd: operands must be 16-byte (inputs) / 8-byte (outputs) aligned");
    if (N > 256) {
        // blocked form: one launch per role, a workgroup per (image, head, 256-row block)
        const int nblk = (N + 255) / 256, ldsb = 4 * 256 * ROWB + 2 * 256 * 4;
        LDIT_DYN_LDS(attention_bwd_bf16_blk<0>, ldsb);
        LDIT_DYN_LDS(attention_bwd_bf16_blk<1>, ldsb);
        hipLaunchKernelGGL(attention_bwd_bf16_blk<0>, dim3((unsigned)(B * H * nblk)), dim3(512), ldsb, stream, static_cast<const bf16_t *>(Q),
                           static_cast<const bf16_t *>(K), static_cast<const bf16_t *>(V), static_cast<const bf16_t *>(O),
                           static_cast<const bf16_t *>(dO), lse, static_cast<bf16_t *>(dQ), static_cast<bf16_t *>(dK),
                           static_cast<bf16_t *>(dV), N, H, nblk, ldqkv, ldo, lddo, lddqkv, scale);
        hipLaunchKernelGGL(attention_bwd_bf16_blk<1>, dim3((unsigned)(B * H * nblk)), dim3(512), ldsb, stream, static_cast<const bf16_t *>(Q),
                           static_cast<const bf16_t *>(K), static_cast<const bf16_t *>(V), static_cast<const bf16_t *>(O),
                           static_cast<const bf16_t *>(dO), lse, static_cast<bf16_t *>(dQ), static_cast<bf16_t *>(dK),
                           static_cast<bf16_t *>(dV), N, H, nblk, ldqkv, ldo, lddo, lddqkv, scale);
        LDIT_HIP_CHECK(hipGetLastError());
        return LDIT_OK;
    }
    const int NP = ((N + 31) / 32) * 32, lds = 4 * NP * ROWB + 2 * NP * 4;
    LDIT_DYN_LDS(attention_bwd_bf16, 4 * 256 * ROWB + 2 * 256 * 4);
    hipLaunchKernelGGL(attention_bwd_bf16, dim3((unsigned)(B * H)), dim3(512), lds, stream, static_cast<const bf16_t *>(Q),
                       static_cast<const bf16_t *>(K), static_cast<const bf16_t *>(V), static_cast<const bf16_t *>(O),
                       static_cast<const bf16_t *>(dO), lse, static_cast<bf16_t *>(dQ), static_cast<bf16_t *>(dK),
                       static_cast<bf16_t *>(dV), N, H, ldqkv, ldo, lddo, lddqkv, scale);
    LDIT_HIP_CHECK(hipGetLastError());
    return LDIT_OK;
}

}  // namespace ldit
