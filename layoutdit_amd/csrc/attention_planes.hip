// Fused multi-head self-attention of the split-fp32 builds (include/ldit.h, LDIT_F32X3 / LDIT_F32X6):  O = softmax(Q K^T * scale) V per
// (image, head), every operand of both products held as TWO (THREE) bf16 planes x ~= p0 + p1 (+ p2) and every product formed from
// three (six) plane products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; softmax fp32.  D = 64, any N.  Described for two
// planes below; three planes = the same kernel with six products and 96 KB of LDS.
//
// Why: with the GEMMs of that build on the bf16 matrix pipe, the fp32-MFMA attention kernel (attention_f32.hip, 1/16 of the bf16
// matrix rate, 110 us per ViT-B layer at bs=64) had become 16 % of the step.  Same flash structure as attention_bf16.hip - 64-key
// chunks of K and V by LDS-DMA into a double buffer, transposed products S^T = K . Q^T and O^T = V^T . P^T so that the score
// accumulator IS the B operand of the second product, exp2-domain softmax on pre-scaled queries (ldit_pack_weights folds
// scale * log2 e into W_q / b_q), running maximum subtracted by one extra MFMA step, deferred rescale - with
//   * K and V chunks staged per plane (four 8-KB images per stage, 64 KB per workgroup: two workgroups per CU),
//   * three MFMAs per (key tile, 16-deep step): k1.q0 + k0.q1 + k0.q0, smallest first,
//   * P split in registers: p0 = bf16(p), p1 = bf16(p - p0) - two conversions and one subtraction per score,
//   * O written as the two bf16 planes of its fp32 value (operand of the o_proj GEMM).
// Layouts: Q, K, V = plane-major column slices of the q|k|v GEMM's output [B*N, 2 * 3C] (plane s of a row at column s * plane_in);
// O = [B*N, 2 * C] (plane s at column s * H * 64).
#include <cstdlib>
#include <type_traits>

#include "ldit_common.h"

namespace ldit {

namespace {

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int KROWB = 128;            // bytes per K row in LDS (64 bf16)
constexpr int KC = 64, HALF = KC * KROWB;   // keys per chunk; one plane image of K or V
constexpr int PIECES = KC / 8;

// The plane products of one fp32 product, smallest first (ldit.h): two planes -> x1 y0 + x0 y1 + x0 y0; three planes ->
// x2 y0 + x1 y1 + x0 y2 + x1 y0 + x0 y1 + x0 y0.
constexpr int n_products(int planes) { return planes == 2 ? 3 : 6; }
constexpr int prod_x(int planes, int g) { return planes == 2 ? (g == 0 ? 1 : 0) : (g == 0 ? 2 : g == 1 ? 1 : g == 3 ? 1 : 0); }
constexpr int prod_y(int planes, int g) { return planes == 2 ? (g == 1 ? 1 : 0) : (g == 1 ? 1 : g == 2 ? 2 : g == 4 ? 1 : 0); }

__device__ __forceinline__ void glds16p(const void *gsrc, char *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// LDS-DMA piece from a wave-uniform 64-bit base + per-lane 32-bit byte offset (see attention_bf16.hip)
__device__ __forceinline__ void glds16p_sbase(const void *ubase, unsigned voff_bytes, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff_bytes), "s"(ubase), "s"(lds_dst)
                 : "memory");
}

// PLANES = 2: 64 KB of LDS (two stages of K0 | K1 | V0 | V1); PLANES = 3 (the attention of LDIT_F32X6): 96 KB.  Either way ONE
// workgroup of eight waves (256 queries) per CU: two waves per SIMD (169 / 217 registers), and one staging of a chunk serves all seven
// query tiles of a 197-token image.  (Four-wave workgroups, two per CU: 3 % slower with two planes; with three planes the LDS
// allows only one of them per CU and the kernel ran no faster than the fp32-MFMA attention: 1.42 vs 1.06 ms per ViT-B forward.)
template <int PLANES, int NW>
__global__ void __launch_bounds__(NW * 64, 2) attention_planes(const bf16_t *__restrict__ Q, const bf16_t *__restrict__ K,
                                                                const bf16_t *__restrict__ V, bf16_t *__restrict__ O, int N, int H,
                                                                int ld_in, int plane_in, int ldo, int nqg)
{
    constexpr int STAGE = 2 * PLANES * HALF, NPROD = n_products(PLANES), PPW = PIECES / NW;
    static_assert(PIECES % NW == 0 && (8 * NW) % 16 == 0, "DMA pieces split evenly over the waves, one swizzle phase per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c32 = lane & 31, h = lane >> 5;
    const int bid = blockIdx.x;
    const int qg = bid % nqg, bh = bid / nqg, head = bh % H, b = bh / H;
    const int qt = qg * NW + wave;
    const bool active = qt * 32 < N;                   // wave-uniform
    const size_t tok0 = (size_t)b * N;

    // Q^T fragments per plane: lane (query c32, half h), k-step s holds Q[query][16s + 8h .. +7]
    bf16x8 qf[PLANES][4];
    {
        int qrow = qt * 32 + c32;
        qrow = qrow < N ? qrow : N - 1;
        const bf16_t *qp = Q + (tok0 + qrow) * ld_in + head * 64 + 8 * h;
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
            for (int s = 0; s < 4; ++s) qf[pl][s] = *reinterpret_cast<const bf16x8 *>(qp + pl * plane_in + 16 * s);
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
            for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(qf[pl][s]));      // retire the loads here (see attention_bf16.hip)
    }

    const int kkey = lane >> 3, vkey = 4 * (lane >> 5) + ((lane & 15) >> 2);
    const int vd = 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
    const bf16_t *Kh = K + tok0 * ld_in + head * 64, *Vh = V + tok0 * ld_in + head * 64;
    const unsigned koff = (unsigned)(8 * wave + kkey) * (unsigned)ld_in + 8u * ((lane & 7) ^ (((8 * wave + kkey) >> 1) & 7));   // elements
    const unsigned voff = (unsigned)(8 * wave + vkey) * (unsigned)ld_in + (unsigned)vd;
    auto issue = [&](int stage, int c0) {
        char *sb = smem + stage * STAGE;
        if (c0 + KC <= N) {                              // wave-uniform
            const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)((__attribute__((address_space(3))) char *)sb));
#pragma unroll
            for (int u = 0; u < PPW; ++u) {
                const int piece = wave + NW * u;
#pragma unroll
                for (int pl = 0; pl < PLANES; ++pl) {
                    const bf16_t *kbase = Kh + (size_t)(c0 + 8 * NW * u) * ld_in + pl * plane_in;
                    const bf16_t *vbase = Vh + (size_t)(c0 + 8 * NW * u) * ld_in + pl * plane_in;
                    glds16p_sbase(kbase, 2u * koff, dst + pl * HALF + piece * 1024);
                    glds16p_sbase(vbase, 2u * voff, dst + (PLANES + pl) * HALF + piece * 1024);
                }
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < PPW; ++u) {
            const int piece = wave + NW * u;
            const int krow = 8 * piece + kkey;
            int key = c0 + krow;
            key = key < N ? key : N - 1;
            int vk = c0 + 8 * piece + vkey;
            vk = vk < N ? vk : N - 1;
#pragma unroll
            for (int pl = 0; pl < PLANES; ++pl) {
                glds16p(Kh + (size_t)key * ld_in + pl * plane_in + 8 * ((lane & 7) ^ ((krow >> 1) & 7)), sb + pl * HALF + piece * 1024);
                glds16p(Vh + (size_t)vk * ld_in + pl * plane_in + vd, sb + (PLANES + pl) * HALF + piece * 1024);
            }
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { o[0][e] = 0.0f; o[1][e] = 0.0f; }
    float m_run = 0.0f, l_run = 0.0f;
    const float lazy = 8.0f;                             // deferred-rescale threshold (2^8 in the exp2 domain)
    bf16x4 ones4 = {(bf16_t)1.0f, (bf16_t)1.0f, (bf16_t)1.0f, (bf16_t)1.0f};
    bf16x4 mfrag = {(bf16_t)0.0f, (bf16_t)0.0f, (bf16_t)0.0f, (bf16_t)0.0f};
    const int sw = (c32 >> 1) & 7;
    const int vlane = 64 * ((lane & 15) >> 2) + 32 * ((lane >> 4) & 1) + 8 * (lane & 3) + 512 * h;

    const int nchunks = (N + KC - 1) / KC;
    issue(0, 0);
    auto step = [&](const int ci, auto nkt_c) {
        constexpr int NKT = decltype(nkt_c)::value;
        const int c0 = ci * KC;
        const int nkeys = (N - c0) < KC ? (N - c0) : KC;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (ci + 1 < nchunks) issue((ci + 1) & 1, c0 + KC);
        if (!active) return;
        const char *Ks = smem + (ci & 1) * STAGE;
        const char *Vs = Ks + PLANES * HALF;
        // ---- S^T = K . Q^T over the plane products, smallest first ------------------------------------------------------------
        f32x16 s[NKT];
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            const char *kr = Ks + (kt * 32 + c32) * KROWB;
            {
                union { bf16x4 f; s16x4 v; } ua, ub;
                ua.f = ones4; ub.f = mfrag;
                s[kt] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(ua.v, ub.v, zero16, 0, 0, 0);      // = -m_run
            }
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                bf16x8 kf[PLANES];
#pragma unroll
                for (int pl = 0; pl < PLANES; ++pl) kf[pl] = *reinterpret_cast<const bf16x8 *>(kr + pl * HALF + (((2 * st + h) ^ sw) * 16));
#pragma unroll
                for (int g = 0; g < NPROD; ++g)
                    s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[prod_x(PLANES, g)], qf[prod_y(PLANES, g)][st], s[kt], 0, 0, 0);
            }
        }
        if (nkeys < NKT * 32) {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((kt * 32 + 8 * (r >> 2) + 4 * h + (r & 3)) >= nkeys) s[kt][r] = -INFINITY;
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // s holds S' = score - m_run (exp2 domain).  The maximum moves on the first chunk and when a chunk beats it by 2^8; then
        // O, the row sum, this chunk's S' and the MFMA operand that carries m are moved exactly once (attention_bf16.hip, PRE).
        float lsum = 0.0f;
        const bool grow = ci == 0 || mx > lazy;
        if (__builtin_amdgcn_ballot_w64(grow)) {
            const float want = grow ? m_run + mx : m_run;
            const bf16_t hi = (bf16_t)(-want);
            const bf16_t lo = (bf16_t)(-want - (float)hi);
            const float m_new = -((float)hi + (float)lo);
            const float d = m_new - m_run;
            const float alpha = ci == 0 ? 1.0f : __builtin_amdgcn_exp2f(-d);
            m_run = m_new;
            if (h == 0) { mfrag[0] = hi; mfrag[1] = lo; }
            l_run *= alpha;
#pragma unroll
            for (int e = 0; e < 16; ++e) { o[0][e] *= alpha; o[1][e] *= alpha; }
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kt][r] -= d;
        }
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(s[kt][r]);
                s[kt][r] = pv;
                lsum += pv;
            }
        l_run += lsum;
        // ---- O^T += V^T . P^T over the plane products ----------------------------------------------------------------------------
        const unsigned vaddr = (unsigned)(uintptr_t)((__attribute__((address_space(3))) const char *)(Vs + vlane));
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                s16x4 vr[PLANES][2][2];             // [plane][dt][u]
#pragma unroll
                for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                        for (int u = 0; u < 2; ++u)
                            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2"
                                         : "=v"(vr[pl][dt][u])
                                         : "v"(vaddr), "n"(pl * HALF + (16 * kt + 8 * st + 4 * u + dt) * 256));
                bf16x8 pp[PLANES];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    float pv = s[kt][8 * st + jj];
#pragma unroll
                    for (int pl = 0; pl < PLANES; ++pl) {
                        pp[pl][jj] = (bf16_t)pv;
                        pv -= (float)pp[pl][jj];
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    union { s16x4 v[2]; bf16x8 f; } vf[PLANES];
#pragma unroll
                    for (int pl = 0; pl < PLANES; ++pl) { vf[pl].v[0] = vr[pl][dt][0]; vf[pl].v[1] = vr[pl][dt][1]; }
#pragma unroll
                    for (int g = 0; g < NPROD; ++g)
                        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[prod_x(PLANES, g)].f, pp[prod_y(PLANES, g)], o[dt], 0, 0, 0);
                }
            }
        }
    };
    const int last_tiles = (N - (nchunks - 1) * KC + 31) >> 5;
    for (int ci = 0; ci + 1 < nchunks; ++ci) step(ci, std::integral_constant<int, 2>{});
    if (last_tiles == 2) step(nchunks - 1, std::integral_constant<int, 2>{});
    else step(nchunks - 1, std::integral_constant<int, 1>{});

    if (!active) return;
    const float l = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l;
    const int qrow = qt * 32 + c32;
    if (qrow < N) {
        bf16_t *op = O + (tok0 + qrow) * ldo + head * 64 + 4 * h;
        const int plane = H * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 t = {o[dt][4 * g + 0] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv};
#pragma unroll
                for (int sp = 0; sp < PLANES; ++sp) {
                    const bf16x4 pk = {(bf16_t)t[0], (bf16_t)t[1], (bf16_t)t[2], (bf16_t)t[3]};
                    *reinterpret_cast<bf16x4 *>(op + sp * plane + dt * 32 + 8 * g) = pk;
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] -= (float)pk[e];
                }
            }
    }
}

}  // namespace

// Q, K, V: plane 0 of the operand's column slice; plane s lies `plane_in` elements further in the same row (row stride ld_in).
// The queries must be pre-multiplied by scale * log2(e).  O: bf16 [B*N, ldo], plane s at column s * H * 64.
template <int PLANES>
static int launch_planes_t(const void *Q, const void *K, const void *V, void *O, int B, int N, int H, int D, int ld_in, int plane_in,
                           int ldo, hipStream_t stream)
{
    if (B <= 0 || N <= 0 || H <= 0) return fail(LDIT_EINVAL, "attention_planes: empty problem");
    if (D != 64) return fail(LDIT_EUNSUPPORTED, "attention_planes: head_dim=%d, only 64 is implemented", D);
    if (!Q || !K || !V || !O) return fail(LDIT_EINVAL, "attention_planes: null operand");
    if ((ld_in | plane_in) & 7 || (ldo & 3) || ldo < PLANES * H * 64 || ld_in < (PLANES - 1) * plane_in + H * 64)
        return fail(LDIT_EINVAL, "attention_planes: bad strides (multiples of 8 in / 4 out, rows wide enough for the planes)");
    if (!aligned16(Q) || !aligned16(K) || !aligned16(V) || (reinterpret_cast<uintptr_t>(O) & 7u))
        return fail(LDIT_EINVAL, "attention_planes: operands must be 16-byte aligned");
    static std::atomic<unsigned long long> attr{0};
    constexpr int lds = 2 * 2 * PLANES * HALF, NW = 8;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(attention_planes<PLANES, NW>), lds, attr)) return rc;
    const int nqt = (N + 31) / 32, nqg = (nqt + NW - 1) / NW;
    hipLaunchKernelGGL((attention_planes<PLANES, NW>), dim3((unsigned)(B * H * nqg)), dim3(NW * 64), lds, stream, static_cast<const bf16_t *>(Q),
                       static_cast<const bf16_t *>(K), static_cast<const bf16_t *>(V), static_cast<bf16_t *>(O), N, H, ld_in, plane_in, ldo,
                       nqg);
    LDIT_HIP_CHECK(hipGetLastError());
    return LDIT_OK;
}

int launch_attention_planes2(const void *Q, const void *K, const void *V, void *O, int B, int N, int H, int D, int ld_in, int plane_in,
                             int ldo, hipStream_t stream)
{
    return launch_planes_t<2>(Q, K, V, O, B, N, H, D, ld_in, plane_in, ldo, stream);
}

int launch_attention_planes3(const void *Q, const void *K, const void *V, void *O, int B, int N, int H, int D, int ld_in, int plane_in,
                             int ldo, hipStream_t stream)
{
    return launch_planes_t<3>(Q, K, V, O, B, N, H, D, ld_in, plane_in, ldo, stream);
}

}  // namespace ldit
