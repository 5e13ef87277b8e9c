// Fused multi-head self-attention, bf16 MFMA, fp32 softmax:  O = softmax(Q K^T * scale) V  per (image, head), no mask.
// Q, K, V: bf16 column slices of the fused [B*N, 3C] projection; O: bf16 [B*N, C].  D = 64, any N (flash-style chunks of
// 256 keys with the online-softmax rescale; N = 1025 for ViT-L at 512x512 is 5 chunks).
//
// Same transposed formulation as attention_f32.hip, on v_mfma_f32_32x32x16_bf16:
//   S^T[key][query] = K . Q^T     A = K fragment (LDS b128: 8 consecutive d of one key), B = Q^T fragment (registers)
//   O^T[d][query]   = V^T . P^T   A = V^T fragment (LDS b128: 8 keys of one d),        B = P^T = the S^T accumulator
// A lane (query j, half h) of an S^T tile holds, in registers 8s..8s+7, the keys 16s + 8(jj>>2) + 4h + (jj&3) - packed
// pairwise to bf16 they ARE the B-operand fragment of k-step s of the second product, provided the A operand uses the
// same key order.  V^T is therefore parked in LDS with the keys of each 32-key tile permuted to
// position 16s + 8h + 4(jj>>2) + (jj&3): one b128 read per lane and step, no shuffles, P never touches LDS.
// Softmax (scale, running max, exp, running sum) is fp32 and lane-local apart from one lane<->lane+32 exchange.
#include "ldit_common.h"

namespace ldit {

namespace {

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int KROWB = 128;            // bytes per K row in LDS (64 bf16), chunk-swizzled like the GEMM tiles
constexpr int VPAD = 8;               // bf16 of padding per V^T row: 528-B stride -> conflict-free b128 reads over d

// OUT_FP8: O is written as fp8 e4m3 codes of o / *qscale (operand of the fp8 o_proj GEMM), ldo in elements.
template <int KT, int NW, bool OUT_FP8>
__global__ void __launch_bounds__(NW * 64, NW / 4) attention_bf16(const bf16_t *__restrict__ Q, const bf16_t *__restrict__ K,
                                                                  const bf16_t *__restrict__ V, void *__restrict__ Ov,
                                                                  int N, int H, int ldq, int ldk, int ldv, int ldo,
                                                                  float scale, int nqg, const float *__restrict__ qscale)
{
    constexpr int KC = KT * 32, VSTR = KC + VPAD;     // keys per chunk; V^T row stride in bf16
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *Ks = smem;                                   // [KC][64] bf16, swizzled 16-B chunks
    bf16_t *Vt = reinterpret_cast<bf16_t *>(smem + KC * KROWB);   // [64][VSTR] bf16, keys permuted per 32-tile

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c32 = lane & 31, h = lane >> 5;
    const int bid = blockIdx.x;
    const int qg = bid % nqg, bh = bid / nqg, head = bh % H, b = bh / H;
    const int qt = qg * NW + wave;
    const bool active = qt * 32 < N;                   // wave-uniform
    const size_t tok0 = (size_t)b * N;

    // Q^T fragments: lane (query c32, half h), k-step s holds Q[query][16s + 8h .. +7]
    bf16x8 qf[4];
    {
        int qrow = qt * 32 + c32;
        qrow = qrow < N ? qrow : N - 1;
        const bf16_t *qp = Q + (tok0 + qrow) * ldq + head * 64 + 8 * h;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8 *>(qp + 16 * s);
    }

    f32x16 o[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { o[0][e] = 0.0f; o[1][e] = 0.0f; }
    float m_run = -INFINITY, l_run = 0.0f;
    const int sw = (c32 >> 1) & 7;

    for (int c0 = 0; c0 < N; c0 += KC) {
        const int nkeys = (N - c0) < KC ? (N - c0) : KC;
        const int ktiles = (nkeys + 31) >> 5;
        if (c0) __syncthreads();
        // ---- stage K (row-major, swizzled) and V^T (transposed, keys permuted); rows past N are zeros -------------------
        for (int u = tid; u < ktiles * 32 * 8; u += NW * 64) {
            const int row = u >> 3, c = u & 7;
            bf16x8 kv, vv;
#pragma unroll
            for (int e = 0; e < 8; ++e) { kv[e] = (bf16_t)0.0f; vv[e] = (bf16_t)0.0f; }
            if (row < nkeys) {
                kv = *reinterpret_cast<const bf16x8 *>(K + (tok0 + c0 + row) * ldk + head * 64 + 8 * c);
                vv = *reinterpret_cast<const bf16x8 *>(V + (tok0 + c0 + row) * ldv + head * 64 + 8 * c);
            }
            *reinterpret_cast<bf16x8 *>(Ks + row * KROWB + ((c ^ ((row >> 1) & 7)) * 16)) = kv;
            // key = 32 kt + 16 s + 8 a + 4 hh + bb  ->  position 32 kt + 16 s + 8 hh + 4 a + bb
            const int k5 = row & 31, pos = (row & ~31) | (k5 & 16) | ((k5 & 4) << 1) | ((k5 & 8) >> 1) | (k5 & 3);
#pragma unroll
            for (int e = 0; e < 8; ++e) Vt[(8 * c + e) * VSTR + pos] = vv[e];
        }
        __syncthreads();
        if (!active) continue;

        // ---- S^T = K . Q^T ---------------------------------------------------------------------------------------------
        f32x16 s[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s[kt][e] = 0.0f;
            if (kt < ktiles) {
                const char *kr = Ks + (kt * 32 + c32) * KROWB;
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(kr + (((2 * st + h) ^ sw) * 16));
                    s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[st], s[kt], 0, 0, 0);
                }
            }
        }
        // ---- scale, mask padded keys, running max --------------------------------------------------------------------------
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            if (kt < ktiles) {
                const bool partial = (kt + 1) * 32 > nkeys;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = s[kt][r] * scale;
                    if (partial && (kt * 32 + 8 * (r >> 2) + 4 * h + (r & 3)) >= nkeys) v = -INFINITY;
                    s[kt][r] = v;
                    mx = fmaxf(mx, v);
                }
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);
        m_run = m_new;
        float lsum = 0.0f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            if (kt < ktiles) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = __expf(s[kt][r] - m_new);
                    s[kt][r] = pv;
                    lsum += pv;
                }
            }
        }
        l_run = l_run * alpha + lsum;
        if (c0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) { o[0][e] *= alpha; o[1][e] *= alpha; }
        }
        // ---- O^T += V^T . P^T -------------------------------------------------------------------------------------------------
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            if (kt < ktiles) {
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    bf16x8 pf;
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) pf[jj] = (bf16_t)s[kt][8 * st + jj];
                    const bf16_t *vr = Vt + c32 * VSTR + kt * 32 + 16 * st + 8 * h;
                    const bf16x8 v0 = *reinterpret_cast<const bf16x8 *>(vr);
                    const bf16x8 v1 = *reinterpret_cast<const bf16x8 *>(vr + 32 * VSTR);
                    o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pf, o[0], 0, 0, 0);
                    o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pf, o[1], 0, 0, 0);
                }
            }
        }
    }

    if (!active) return;
    const float l = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = OUT_FP8 ? 1.0f / (l * qscale[0]) : 1.0f / l;
    const int qrow = qt * 32 + c32;
    if (OUT_FP8) {
        if (qrow < N) {
            unsigned char *op = static_cast<unsigned char *>(Ov) + (tok0 + qrow) * ldo + head * 64 + 4 * h;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<unsigned *>(op + dt * 32 + 8 * g) =
                        pack_fp8x4(o[dt][4 * g + 0] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
        }
    } else if (qrow < N) {
        bf16_t *op = static_cast<bf16_t *>(Ov) + (tok0 + qrow) * ldo + head * 64 + 4 * h;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const bf16x4 t = {(bf16_t)(o[dt][4 * g + 0] * inv), (bf16_t)(o[dt][4 * g + 1] * inv),
                                  (bf16_t)(o[dt][4 * g + 2] * inv), (bf16_t)(o[dt][4 * g + 3] * inv)};
                *reinterpret_cast<bf16x4 *>(op + dt * 32 + 8 * g) = t;
            }
    }
}

}  // namespace

template <bool OUT_FP8>
static int launch_attn(const void *Q, const void *K, const void *V, void *O, int B, int N, int H, int D, int ldq, int ldk, int ldv,
                       int ldo, float scale, const float *qscale, hipStream_t stream)
{
    if (B <= 0 || N <= 0 || H <= 0) return fail(LDIT_EINVAL, "attention_bf16: empty problem");
    if (D != 64) return fail(LDIT_EUNSUPPORTED, "attention_bf16: head_dim=%d, only 64 is implemented", D);
    if (!Q || !K || !V || !O) return fail(LDIT_EINVAL, "attention_bf16: null operand");
    if ((ldq | ldk | ldv) & 7 || (ldo & 3)) return fail(LDIT_EINVAL, "attention_bf16: row strides must be multiples of 8 (in) / 4 (out)");
    if (!aligned16(Q) || !aligned16(K) || !aligned16(V) || (reinterpret_cast<uintptr_t>(O) & (OUT_FP8 ? 3u : 7u)))
        return fail(LDIT_EINVAL, "attention_bf16: operands must be 16-byte aligned");
    constexpr int KT = 8, NW = 8;
    constexpr int lds = KT * 32 * KROWB + 64 * (KT * 32 + VPAD) * 2;
    const int nqt = (N + 31) / 32, nqg = (nqt + NW - 1) / NW;
    auto kern = attention_bf16<KT, NW, OUT_FP8>;
    static bool attr_set = false;
    if (!attr_set) {
        LDIT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(B * H * nqg)), dim3(NW * 64), lds, stream, static_cast<const bf16_t *>(Q),
                       static_cast<const bf16_t *>(K), static_cast<const bf16_t *>(V), O, N, H, ldq, ldk, ldv, ldo, scale, nqg,
                       qscale);
    LDIT_HIP_CHECK(hipGetLastError());
    return LDIT_OK;
}

int launch_attention_bf16(const void *Q, const void *K, const void *V, void *O, int B, int N, int H, int D, int ldq, int ldk,
                          int ldv, int ldo, float scale, hipStream_t stream)
{
    return launch_attn<false>(Q, K, V, O, B, N, H, D, ldq, ldk, ldv, ldo, scale, nullptr, stream);
}

int launch_attention_bf16_fp8out(const void *Q, const void *K, const void *V, void *O, int B, int N, int H, int D, int ldq,
                                 int ldk, int ldv, int ldo, float scale, const float *qscale, hipStream_t stream)
{
    if (!qscale) return fail(LDIT_EINVAL, "attention_bf16: fp8 output needs a scale");
    return launch_attn<true>(Q, K, V, O, B, N, H, D, ldq, ldk, ldv, ldo, scale, qscale, stream);
}

}  // namespace ldit
