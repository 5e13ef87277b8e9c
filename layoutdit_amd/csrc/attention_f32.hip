// Fused multi-head self-attention, fp32 MFMA:  O = softmax(Q K^T * scale) V  per (image, head), no mask.
// (TF:models/beit/modeling_beit.py:268-293 eager definition, :323-338 SDPA call; softmax in fp32.)
//
// gfx950 design (D = 64):
//   * one 512-thread workgroup = one (image, head, group of 8 query tiles); wave w owns 32 queries.  All 7 query tiles
//     of a 197-token image share ONE staging of the head's K and V, and the two waves per SIMD cover each other's
//     softmax (VALU) with MFMAs.  (First version: 4-wave groups, K/V staged twice per head, 150 us per layer.)
//   * K and V of the head are staged in LDS in chunks of KT key tiles (KT = 7 -> all 197 (+pad) keys of a 224x224
//     image in ONE chunk: 59.5 KiB K (rows padded to 272 B so ds_read_b128 is conflict-free) + 56 KiB V);
//     longer sequences (N = 1025 at 512x512) stream chunks with the usual online-softmax rescale.
//   * products are computed TRANSPOSED with v_mfma_f32_32x32x2_f32 so softmax never leaves the lane:
//       S^T[key][query] = K . Q^T      A = K fragment (LDS, b128), B = Q^T (registers, loaded once)
//       O^T[d][query]   = V^T . P^T    A = V^T fragment (LDS, b32, conflict-free), B = the S^T accumulator register itself
//     An S^T accumulator register r of lane (query j, half h) holds key 8(r>>2)+4h+(r&3) of the tile - exactly the
//     B-operand slot (k = h) of a 32x32x2 step over keys {8(r>>2)+(r&3), +4}.  So P feeds the second product
//     straight from the accumulator: no LDS round trip, no cross-lane shuffles.  The row max / row sum are lane-local
//     over the registers plus ONE lane<->lane+32 exchange.
//   * O^T leaves each lane with 4 consecutive d per register quad -> 16-B stores.
#include "ldit_common.h"

#ifdef LDIT_GEMM_STAMPS
// diagnostic build only (make dbg; scripts/attn_f32_stamps.py): per-workgroup phase cycles of wave 0, written to a buffer
// registered by ldit_dbg_set_attn32_stamps - 6 x int64 per workgroup: Q load + staging, S = K Q^T, softmax, P V, store, total
__device__ unsigned long long *g_attn32_stamps = nullptr;
#define A32_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define A32_STAMP(var)
#endif

namespace ldit {

namespace {

constexpr int KSTR = 68;   // floats per K row in LDS (64 + 4 pad: 272-B stride -> 16 lanes hit 16 distinct 16-B slots)
constexpr int VSTR = 64;

// (NW = 4 is held to the 256 registers of NW = 8: given 512, hipcc parks every V fragment read right in front of its MFMA
// and the P V phase ran 2.7x longer - 42.8k vs 15.9k cycles per wave.)
template <int KT, int NW>
__global__ void __launch_bounds__(NW * 64, NW == 4 ? 2 : NW / 4) attention_f32(const float *__restrict__ Q, const float *__restrict__ K,
                                                     const float *__restrict__ V, float *__restrict__ O, int N, int H,
                                                     int ldq, int ldk, int ldv, int ldo, float scale, int nqg)
{
    constexpr int KROWS = KT * 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *Ks = reinterpret_cast<float *>(smem);
    float *Vs = Ks + KROWS * KSTR;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int bid = blockIdx.x;
    const int qg = bid % nqg, bh = bid / nqg, head = bh % H, b = bh / H;
    const int qt = qg * NW + wave;
    const bool active = qt * 32 < N;          // wave-uniform
    const size_t tok0 = (size_t)b * N;
#ifdef LDIT_GEMM_STAMPS
    unsigned long long ph[4] = {0, 0, 0, 0};
#endif
    A32_STAMP(t_begin);

    // Q^T operand: lane (query li, half lh) keeps Q[query][32*lh .. 32*lh+31]; MFMA step s pairs d = {s, 32+s}
    float q[32];
    {
        int qrow = qt * 32 + li;
        qrow = qrow < N ? qrow : N - 1;
        const f32x4 *qp = reinterpret_cast<const f32x4 *>(Q + (tok0 + qrow) * ldq + head * 64 + 32 * lh);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const f32x4 t = qp[u];
            q[4 * u + 0] = t[0]; q[4 * u + 1] = t[1]; q[4 * u + 2] = t[2]; q[4 * u + 3] = t[3];
        }
    }

    f32x16 o[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { o[0][e] = 0.0f; o[1][e] = 0.0f; }
    float m_run = -INFINITY, l_run = 0.0f;

    for (int c0 = 0; c0 < N; c0 += KROWS) {
        const int nkeys = (N - c0) < KROWS ? (N - c0) : KROWS;
        const int ktiles = (nkeys + 31) >> 5;                    // wave- and block-uniform
        A32_STAMP(t0);
        if (c0) __syncthreads();                                 // previous chunk fully consumed
        // ---- stage K, V chunk (zero-fill keys >= N inside the last tile: 0 * garbage must not make NaN) ---------
        // A batch of a thread's loads is issued before its first LDS store (two to four memory round trips per chunk, not one
        // per element: the rolled loop paid ~0.8 us for each of its 7-14 trips, a third of the kernel at small batches).
        {
            constexpr int PER = (32 * 16) / (NW * 64) > 0 ? (32 * 16) / (NW * 64) : 1;     // float4 per thread, key tile and operand
            static_assert((32 * 16) % (NW * 64) == 0, "a key tile must split evenly over the threads");
            constexpr int BATCH = 4;                   // key tiles per pass: 4 + 4 loads in flight (32 registers)
#pragma unroll
            for (int e = 0; e < PER; ++e)
#pragma unroll
                for (int k0 = 0; k0 < KT; k0 += BATCH) {
                    f32x4 kv[BATCH], vv[BATCH];
#pragma unroll
                    for (int i = 0; i < BATCH; ++i) {
                        const int kt = k0 + i, u = kt * 512 + e * NW * 64 + tid, row = u >> 4, c4 = (u & 15) * 4;
                        kv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                        vv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (kt < KT && kt < ktiles && row < nkeys) {
                            kv[i] = *reinterpret_cast<const f32x4 *>(K + (tok0 + c0 + row) * ldk + head * 64 + c4);
                            vv[i] = *reinterpret_cast<const f32x4 *>(V + (tok0 + c0 + row) * ldv + head * 64 + c4);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < BATCH; ++i) {
                        const int kt = k0 + i, u = kt * 512 + e * NW * 64 + tid, row = u >> 4, c4 = (u & 15) * 4;
                        if (kt < KT && kt < ktiles) {
                            *reinterpret_cast<f32x4 *>(Ks + row * KSTR + c4) = kv[i];
                            *reinterpret_cast<f32x4 *>(Vs + row * VSTR + c4) = vv[i];
                        }
                    }
                }
        }
        __syncthreads();
        if (!active) continue;
        A32_STAMP(t1);

        // ---- S^T = K . Q^T --------------------------------------------------------------------------------------
        f32x16 s[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s[kt][e] = 0.0f;
            if (kt < ktiles) {
                const float *kr = Ks + (kt * 32 + li) * KSTR + 32 * lh;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const f32x4 kf = *reinterpret_cast<const f32x4 *>(kr + 4 * u);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], q[4 * u + e], s[kt], 0, 0, 0);
                }
            }
        }
#ifdef LDIT_GEMM_STAMPS
        asm volatile("s_nop 0" ::"v"(s[0][0]), "v"(s[KT - 1][0]));
#endif
        A32_STAMP(t2);
        // ---- scale, mask the padded keys, chunk max ---------------------------------------------------------------
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            if (kt < ktiles) {
                const bool partial = (kt + 1) * 32 > nkeys;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = s[kt][r] * scale;
                    if (partial && (kt * 32 + 8 * (r >> 2) + 4 * lh + (r & 3)) >= nkeys) v = -INFINITY;
                    s[kt][r] = v;
                    mx = fmaxf(mx, v);
                }
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);               // first chunk: exp(-inf) = 0
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
        l_run = l_run * alpha + lsum;                            // per-half partial; halves are added at the end
        if (c0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) { o[0][e] *= alpha; o[1][e] *= alpha; }
        }
        A32_STAMP(t3);
        // ---- O^T += V^T . P^T ---------------------------------------------------------------------------------------
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            if (kt < ktiles) {
                const int left = nkeys - kt * 32;          // valid keys in this tile (uniform); < 32 only in the last one
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // step r multiplies keys 8(r>>2)+(r&3) and +4: skip it when both lie in the zero padding
                    // (197 tokens: the 7th key tile holds 5 keys -> 12 of its 16 steps, 5 % of all MFMAs, vanish)
                    if (8 * (r >> 2) + (r & 3) >= left) continue;
                    const float *vr = Vs + (kt * 32 + 8 * (r >> 2) + 4 * lh + (r & 3)) * VSTR + li;
                    o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[0], s[kt][r], o[0], 0, 0, 0);
                    o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[32], s[kt][r], o[1], 0, 0, 0);
                }
            }
        }
#ifdef LDIT_GEMM_STAMPS
        asm volatile("s_nop 0" ::"v"(o[0][0]), "v"(o[1][0]));
        ph[0] += t1 - t0 + (c0 ? 0 : t0 - t_begin); ph[1] += t2 - t1; ph[2] += t3 - t2; ph[3] += __builtin_amdgcn_s_memtime() - t3;
#endif
    }

    if (!active) return;
#ifdef LDIT_GEMM_STAMPS
    if (g_attn32_stamps && wave == 0 && lane == 0) {
        unsigned long long *d = g_attn32_stamps + (size_t)blockIdx.x * 6;
        d[0] = ph[0]; d[1] = ph[1]; d[2] = ph[2]; d[3] = ph[3]; d[5] = __builtin_amdgcn_s_memtime() - t_begin;
    }
#endif
    const float l = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l;
    const int qrow = qt * 32 + li;
    if (qrow < N) {
        float *op = O + (tok0 + qrow) * ldo + head * 64 + 4 * lh;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 t = {o[dt][4 * g + 0] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv};
                *reinterpret_cast<f32x4 *>(op + dt * 32 + 8 * g) = t;
            }
    }
}

}  // namespace

int launch_attention(const float *Q, const float *K, const float *V, float *O, int B, int N, int H, int D, int ldq,
                     int ldk, int ldv, int ldo, float scale, hipStream_t stream)
{
    if (B <= 0 || N <= 0 || H <= 0) return fail(LDIT_EINVAL, "attention: empty problem");
    if (D != 64) return fail(LDIT_EUNSUPPORTED, "attention: head_dim=%d, only 64 is implemented", D);
    if (!Q || !K || !V || !O) return fail(LDIT_EINVAL, "attention: null operand");
    if ((ldq | ldk | ldv | ldo) & 3) return fail(LDIT_EINVAL, "attention: row strides must be multiples of 4 floats");
    if (!aligned16(Q) || !aligned16(K) || !aligned16(V) || !aligned16(O)) return fail(LDIT_EINVAL, "attention: operands must be 16-byte aligned");
    constexpr int KT = 7;
    constexpr int lds = KT * 32 * (KSTR + VSTR) * 4;
    const int nqt = (N + 31) / 32;
    auto go = [&](auto kern, int nw, std::atomic<unsigned long long> &attr_done) -> int {
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, attr_done)) return rc;
        const int nqg = (nqt + nw - 1) / nw;
        hipLaunchKernelGGL(kern, dim3((unsigned)(B * H * nqg)), dim3(nw * 64), lds, stream, Q, K, V, O, N, H, ldq, ldk, ldv, ldo,
                           scale, nqg);
        LDIT_HIP_CHECK(hipGetLastError());
        return LDIT_OK;
    };
    // Eight query tiles per workgroup share one staging of the head's K and V - right when B*H workgroups fill the chip.
    // At serving sizes (one image = 12 workgroups, each a CU with two waves per SIMD queueing on one matrix pipe) the
    // query tiles are spread over twice the workgroups, one wave per SIMD: 34 -> 18 us at B = 1.  A query tile's arithmetic
    // does not depend on which workgroup computes it, so the results are bit-identical either way.
    static std::atomic<unsigned long long> set8{0}, set4{0};      // per-device bookkeeping (ensure_dynamic_lds)
    const long wgs8 = (long)B * H * ((nqt + 7) / 8);
    if (wgs8 <= 128 && nqt > 4) { if (int rc = go(attention_f32<KT, 4>, 4, set4)) return rc; }
    else if (int rc = go(attention_f32<KT, 8>, 8, set8)) return rc;
    LDIT_HIP_CHECK(hipGetLastError());
    return LDIT_OK;
}

}  // namespace ldit

#ifdef LDIT_GEMM_STAMPS
extern "C" int ldit_dbg_set_attn32_stamps(void *buf)
{
    unsigned long long *p = static_cast<unsigned long long *>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_attn32_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -3;
}
#endif
