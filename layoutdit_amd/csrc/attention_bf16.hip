// Fused multi-head self-attention, bf16 MFMA, fp32 softmax:  O = softmax(Q K^T * scale) V  per (image, head), no mask.
// Q, K, V: bf16 column slices of the fused [B*N, 3C] projection; O: bf16 [B*N, C] (or fp8 codes).  D = 64, any N
// (flash-style chunks of 64 keys with the online-softmax rescale; N = 1025 for ViT-L at 512x512 is 17 chunks).
//
// Same transposed formulation as attention_f32.hip, on v_mfma_f32_32x32x16_bf16:
//   S^T[key][query] = K . Q^T     A = K fragment (LDS b128: 8 consecutive d of one key), B = Q^T fragment (registers)
//   O^T[d][query]   = V^T . P^T   A = V^T fragment (two ds_read_b64_tr_b16: 4 + 4 keys of one d), B = P^T = the S^T accumulator
// A lane (query j, half h) of an S^T tile holds, in registers 8s..8s+7, the keys 16s + 8(jj>>2) + 4h + (jj&3) - packed
// pairwise to bf16 they ARE the B-operand fragment of k-step s of the second product, provided the A operand uses the
// same key order: the transposing LDS read delivers exactly that (rows = 4 consecutive keys starting at 16s + 8u + 4h,
// columns = the 16 d of the lane group), so V stays key-major in LDS, P never touches LDS and nothing is shuffled.
// K and V chunks are staged by LDS-DMA (global_load_lds, no registers, no scatter writes) into a double buffer: chunk
// c+1 flies while chunk c is multiplied, with a single workgroup barrier per chunk.  K image: 128-B rows, 16-B chunk XOR-swizzled with (row>>1)&7 (conflict-free
// b128 reads); V image: 256-B blocks [4 keys][32 d] - one transposing read of a 32-lane half is exactly one block, i.e.
// one full sweep of the 64 banks.  Rows past N are clamped duplicates of the last key and masked to -inf.
// Softmax runs on raw scores in the exp2 domain (p = exp2(s c - m c), c = scale log2 e: one FMA + one v_exp_f32 per
// element), fp32, lane-local apart from one lane<->lane+32 exchange.  Four waves (128 queries) per workgroup, 32 KB of
// LDS and 128 VGPRs: four workgroups per CU, so on every SIMD one wave's softmax VALU work overlaps another's MFMAs.
#include <cstdlib>
#include <type_traits>

#include "ldit_common.h"

#define LDIT_TRY_RC(expr) do { int rc__ = (expr); if (rc__ != LDIT_OK) return rc__; } while (0)

#ifdef LDIT_GEMM_STAMPS
// diagnostic build only (make dbg; scripts/attn_stamps.py): per-workgroup phase cycles of wave 0, written to a buffer
// registered by ldit_dbg_set_attn_stamps - 6 x int64 per workgroup: wait+barrier, DMA issue, S = K Q^T issue,
// softmax (incl. the wait for S), P V, kernel total
__device__ unsigned long long *g_attn_stamps = nullptr;
#define ATT_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define ATT_STAMP(var)
#endif

namespace ldit {

namespace {

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int KROWB = 128;            // bytes per K row in LDS (64 bf16)

__device__ __forceinline__ void glds16a(const void *gsrc, char *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// LDS-DMA piece with a wave-uniform 64-bit base in SGPRs and a per-lane 32-bit BYTE offset: `global_load_lds_dwordx4 voff, s[base]`
// (the builtin only takes a per-lane 64-bit pointer: two VALU adds per piece and a live VGPR pair).  M0 (the LDS destination)
// is compiler-reserved: saved and restored inside the statement (cdna guide 5.7).
__device__ __forceinline__ void glds16_sbase(const void *ubase, unsigned voff_bytes, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff_bytes), "s"(ubase), "s"(lds_dst)
                 : "memory");
}

// OUT_FP8: O is written as fp8 e4m3 codes of o / *qscale (operand of the fp8 o_proj GEMM), ldo in elements.
// PRE (round 3): Q arrives already multiplied by scale * log2(e) (the packed inference path folds that factor into W_q / b_q at
// pack time: api.hip, ldit_pack_weights), so a raw score IS its exp2-domain exponent, and the running maximum is subtracted
// BY THE MATRIX PIPE: one extra MFMA step per key tile with an all-ones A column and B = (-m_hi, -m_lo) (the maximum split into
// two bf16 so that 2^-17 |m| is all it loses) initialises the S^T accumulators with -m_run.  p = exp2(S') then needs no
// per-element FMA: 32 of the ~180 VALU instructions of a chunk, on a kernel whose four waves per SIMD saturate instruction
// issue (PMC: SQ_ACTIVE_INST_ANY x 4 waves = 1.07 of the SIMD's cycles; MFMA pipe 32 % busy).
template <int KT, int NW, bool OUT_FP8, bool PRE>
__global__ void __launch_bounds__(NW * 64, KT == 2 ? 4 : 2) attention_bf16(const bf16_t *__restrict__ Q, const bf16_t *__restrict__ K,
                                                             const bf16_t *__restrict__ V, void *__restrict__ Ov,
                                                             int N, int H, int ldq, int ldk, int ldv, int ldo,
                                                             float scale, int nqg, const float *__restrict__ qscale,
                                                             float *__restrict__ lse)
{
    constexpr int KC = KT * 32, HALF = KC * KROWB, STAGE = 2 * HALF;     // keys per chunk; K image, then V image
    constexpr int PIECES = KC / 8, PPW = PIECES / NW;                    // 1-KB DMA pieces (8 keys) per operand, per wave
    static_assert(PIECES % NW == 0, "DMA pieces must split evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];

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
        // retire these loads HERE: left pending, hipcc guards their first use inside the chunk loop with s_waitcnt vmcnt(0),
        // which would also drain the DMA of the next chunk on every iteration
#pragma unroll
        for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(qf[s]));
    }

    // DMA source geometry of this lane.  K piece: 8 keys x 128 B, lane -> key lane>>3, source chunk (lane&7) ^ swizzle.
    // V piece: 8 keys x 128 B as four 256-B blocks [4 keys][32 d]: lane -> block lane>>4 (key group (lane>>5), d half
    // (lane>>4)&1), key (lane&15)>>2 of the group, 16-B piece lane&3 of the 64-B half row.
    const int kkey = lane >> 3, vkey = 4 * (lane >> 5) + ((lane & 15) >> 2);
    const int vd = 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
    const bf16_t *Kh = K + tok0 * ldk + head * 64, *Vh = V + tok0 * ldv + head * 64;
    // Per-lane source offsets inside a chunk are constants of the kernel; the chunk's base address is wave-uniform and advances
    // in scalar registers, so issuing a chunk costs no vector address arithmetic (round 2 recomputed the clamp, the row
    // product and a 64-bit add per piece: ~30 VALU instructions per chunk on a kernel that is VALU-issue bound).  Only a chunk
    // that reaches past key N-1 (the short last one) takes the clamped path (rows past N = duplicates of the last key).
    // (piece u of a wave is 8 NW keys further down: the XOR swizzle term (krow >> 1) & 7 does not change, so ONE offset per
    // operand serves every piece - the step goes into the scalar base)
    static_assert((8 * NW) % 16 == 0, "pieces of one wave must share the swizzle phase");
    const unsigned koff = (unsigned)(8 * wave + kkey) * (unsigned)ldk + 8u * ((lane & 7) ^ (((8 * wave + kkey) >> 1) & 7));   // elements
    const unsigned voff = (unsigned)(8 * wave + vkey) * (unsigned)ldv + (unsigned)vd;
    auto issue = [&](int stage, int c0) {
        char *kb = smem + stage * STAGE, *vb = kb + HALF;
        if (c0 + KC <= N) {                              // wave-uniform
            const unsigned kdst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)((__attribute__((address_space(3))) char *)kb));
#pragma unroll
            for (int u = 0; u < PPW; ++u) {
                const int piece = wave + NW * u;
                const bf16_t *kbase = Kh + (size_t)(c0 + 8 * NW * u) * ldk, *vbase = Vh + (size_t)(c0 + 8 * NW * u) * ldv;
                glds16_sbase(kbase, 2u * koff, kdst + piece * 1024);
                glds16_sbase(vbase, 2u * voff, kdst + HALF + piece * 1024);
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < PPW; ++u) {
            const int piece = wave + NW * u;
            const int krow = 8 * piece + kkey;
            int key = c0 + krow;
            key = key < N ? key : N - 1;
            glds16a(Kh + (size_t)key * ldk + 8 * ((lane & 7) ^ ((krow >> 1) & 7)), kb + piece * 1024);
            int vk = c0 + 8 * piece + vkey;
            vk = vk < N ? vk : N - 1;
            glds16a(Vh + (size_t)vk * ldv + vd, vb + piece * 1024);
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { o[0][e] = 0.0f; o[1][e] = 0.0f; }
    float m_run = PRE ? 0.0f : -INFINITY, l_run = 0.0f;
    const float c = scale * 1.44269504088896340736f;     // softmax in the exp2 domain (PRE: already inside q)
    const float lazy = PRE ? 8.0f : 8.0f / c;            // deferred-rescale threshold in score units (2^8 in the exp2 domain)
    // PRE: operands of the extra MFMA step.  A = 1 at k = 0, 1 for every key row; B = (-m_hi, -m_lo) at k = 0, 1 of the lane's
    // query (k = 8 h + j: only the h = 0 half carries them).
    // (the step runs on v_mfma_f32_32x32x8_bf16 - lane (row, h) supplies k = 4 h + j, j < 4 - so its operands are two registers
    // each: A = all ones, B = (-m_hi, -m_lo, 0, 0) in the h = 0 half, zeros in the other)
    // (Row sums on the matrix pipe - an all-ones A operand per 16-key step - were built and measured: 112.5 us against 106.2 us
    // at N = 1025; four more MFMAs per chunk cost more than the 16 v_pk_add_f32 they replace.  Not kept.)
    bf16x4 ones4 = {(bf16_t)1.0f, (bf16_t)1.0f, (bf16_t)1.0f, (bf16_t)1.0f};
    bf16x4 mfrag = {(bf16_t)0.0f, (bf16_t)0.0f, (bf16_t)0.0f, (bf16_t)0.0f};
    const int sw = (c32 >> 1) & 7;
    // transposing read: lane 4q+p of a 16-lane group addresses key q, d 4p..4p+3 of the group's 16 columns
    const int vlane = 64 * ((lane & 15) >> 2) + 32 * ((lane >> 4) & 1) + 8 * (lane & 3) + 512 * h;

    const int nchunks = (N + KC - 1) / KC;
#ifdef LDIT_GEMM_STAMPS
    unsigned long long tw = 0, ti = 0, ts = 0, tx = 0, tp = 0;
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
    issue(0, 0);
    // One chunk: wait for its K/V images, start the next chunk's DMA, multiply.  Instantiated per number of live 32-key tiles
    // so that the score registers are straight-line values inside it (per-tile `if (kt < ktiles)` guards made every tile a
    // phi and cost ~40 v_mov per chunk): the loop below runs the full chunks, the short last chunk is peeled behind it (a
    // last chunk of N = 197 skips its dead tile).  Two instantiations INSIDE the loop spill at 128 registers; peeled they do not.
    auto step = [&](const int ci, auto nkt_c) {
        constexpr int NKT = decltype(nkt_c)::value;
        ATT_STAMP(t0);
        const int c0 = ci * KC;
        const int nkeys = (N - c0) < KC ? (N - c0) : KC;
        // ONE barrier per chunk: behind it every wave's pieces of chunk ci have landed AND every wave has finished
        // multiplying chunk ci-1 (program order), so the other stage is free - chunk ci+1 is issued right here and has the
        // whole of chunk ci's arithmetic to land.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        ATT_STAMP(t1);
        if (ci + 1 < nchunks) issue((ci + 1) & 1, c0 + KC);
        ATT_STAMP(t2);
#ifdef LDIT_GEMM_STAMPS
        tw += t1 - t0; ti += t2 - t1;
#endif
        if (active) {
            const char *Ks = smem + (ci & 1) * STAGE;
            const char *Vs = Ks + HALF;
            {
                // ---- S^T = K . Q^T ---------------------------------------------------------------------------------
                f32x16 s[NKT];
                const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) {
                    const char *kr = Ks + (kt * 32 + c32) * KROWB;
                    if (PRE) {
                        union { bf16x4 f; s16x4 v; } ua, ub;
                        ua.f = ones4; ub.f = mfrag;
                        s[kt] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(ua.v, ub.v, zero16, 0, 0, 0);      // = -m_run
                    }
#pragma unroll
                    for (int st = 0; st < 4; ++st) {
                        const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(kr + (((2 * st + h) ^ sw) * 16));
                        s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[st], (PRE || st) ? s[kt] : zero16, 0, 0, 0);
                    }
                }
                ATT_STAMP(t3);
                // ---- mask padded keys (last chunk only: a real branch, not 2 VALU ops on every score), running max on
                //      the raw scores (scale > 0 commutes with max) ------------------------------------------------------
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
                // Deferred rescale: the running maximum only moves when a chunk beats it by more than 2^8 in the exp2
                // domain (always on the first chunk, m_run = -inf).  Until then p = exp2(c s - c m_run) may exceed 1 by up
                // to 2^8 - harmless in fp32 sums and in bf16 P (a power-of-two scale of the same mantissas) - and the 32
                // accumulator multiplies, the alpha exp2 and the l rescale are skipped behind a wave-uniform branch.
                float lsum = 0.0f;
                if constexpr (PRE) {
                    // s holds S' = score - m_run.  The maximum moves on the first chunk (m_run = 0 there: whatever the scores
                    // are, they are re-based on their own maximum) and when a chunk beats it by 2^8.  Everything still at the
                    // old maximum is moved exactly once: O, the row sum, THIS chunk's S' and the MFMA operand that carries m.
                    const bool grow = ci == 0 || mx > lazy;
                    if (__builtin_amdgcn_ballot_w64(grow)) {
                        // the new maximum as the sum of two bf16 (what the extra MFMA step can carry); d = the exact shift applied
                        const float want = grow ? m_run + mx : m_run;
                        const bf16_t hi = (bf16_t)(-want);
                        const bf16_t lo = (bf16_t)(-want - (float)hi);
                        const float m_new = -((float)hi + (float)lo);
                        const float d = m_new - m_run;                   // exactly 0 for lanes that keep m_run (same hi, lo)
                        // first chunk: O and the row sum are zero, and exp2(-d) may overflow for very negative scores
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
                } else {
                const bool grow = mx > m_run + lazy;
                if (__builtin_amdgcn_ballot_w64(grow)) {
                    const float m_new = grow ? mx : m_run;
                    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);      // 1 for lanes that keep m_run
                    m_run = m_new;
                    l_run *= alpha;
#pragma unroll
                    for (int e = 0; e < 16; ++e) { o[0][e] *= alpha; o[1][e] *= alpha; }
                }
                const float mc = m_run * c;
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], c, -mc));
                        s[kt][r] = pv;
                        lsum += pv;
                    }
                }
                l_run += lsum;
                ATT_STAMP(t4);
                // ---- O^T += V^T . P^T -----------------------------------------------------------------------------------
                // The transposing reads are inline asm: through the builtin, hipcc cannot tell that they do not alias the
                // LDS-DMA writes of the NEXT chunk and guards each group with s_waitcnt vmcnt(0), which would serialise
                // the double buffer.  Asm reads are invisible to the compiler's waitcnt pass, hence the explicit
                // lgkmcnt(0) and the sched_barrier that keeps the MFMAs below it.
                const unsigned vaddr = (unsigned)(uintptr_t)((__attribute__((address_space(3))) const char *)(Vs + vlane));
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) {
                    // block (key group, d half) = ((32 kt + 16 st + 8 u + 4 h) / 4) * 2 + dt; h is in vlane
                    s16x4 vr[2][2][2];                 // [st][dt][u]
#pragma unroll
                    for (int st = 0; st < 2; ++st)
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                            for (int u = 0; u < 2; ++u)
                                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2"
                                             : "=v"(vr[st][dt][u])
                                             : "v"(vaddr), "n"((16 * kt + 8 * st + 4 * u + dt) * 256));
                    bf16x8 pf[2];
#pragma unroll
                    for (int st = 0; st < 2; ++st)
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) pf[st][jj] = (bf16_t)s[kt][8 * st + jj];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int st = 0; st < 2; ++st)
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt) {
                            union { s16x4 v[2]; bf16x8 f; } vf;
                            vf.v[0] = vr[st][dt][0]; vf.v[1] = vr[st][dt][1];
                            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.f, pf[st], o[dt], 0, 0, 0);
                        }
                }
#ifdef LDIT_GEMM_STAMPS
                asm volatile("s_nop 0" :: "v"(o[0][0]), "v"(o[1][0]));
                const unsigned long long t5 = __builtin_amdgcn_s_memtime();
                ts += t3 - t2; tx += t4 - t3; tp += t5 - t4;
#endif
            }
        }
    };
    const int last_tiles = (N - (nchunks - 1) * KC + 31) >> 5;
    for (int ci = 0; ci + 1 < nchunks; ++ci) step(ci, std::integral_constant<int, KT>{});
    if (last_tiles == KT) step(nchunks - 1, std::integral_constant<int, KT>{});
    else if (KT == 4 && last_tiles == 3) step(nchunks - 1, std::integral_constant<int, KT == 4 ? 3 : 1>{});
    else if (KT == 4 && last_tiles == 2) step(nchunks - 1, std::integral_constant<int, KT == 4 ? 2 : 1>{});
    else step(nchunks - 1, std::integral_constant<int, 1>{});

#ifdef LDIT_GEMM_STAMPS
    if (g_attn_stamps && wave == 0 && lane == 0) {
        unsigned long long *d = g_attn_stamps + (size_t)blockIdx.x * 6;
        d[0] = tw; d[1] = ti; d[2] = ts; d[3] = tx; d[4] = tp; d[5] = __builtin_amdgcn_s_memtime() - t_begin;
    }
#endif
    if (!active) return;
    const float l = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = OUT_FP8 ? 1.0f / (l * qscale[0]) : 1.0f / l;
    const int qrow = qt * 32 + c32;
    // train step: log2-domain log-sum-exp of the scaled scores, L2 = log2 sum_k exp2(c s_k) = m c + log2 l, so that the
    // backward recomputes p = exp2(c s - L2) with no row maximum (layout [B, H, N])
    if (lse && h == 0 && qrow < N) lse[((size_t)b * H + head) * N + qrow] = (PRE ? m_run : m_run * c) + __builtin_amdgcn_logf(l);
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
                       int ldo, float scale, const float *qscale, float *lse, hipStream_t stream)
{
    if (B <= 0 || N <= 0 || H <= 0) return fail(LDIT_EINVAL, "attention_bf16: empty problem");
    if (D != 64) return fail(LDIT_EUNSUPPORTED, "attention_bf16: head_dim=%d, only 64 is implemented", D);
    if (!Q || !K || !V || !O) return fail(LDIT_EINVAL, "attention_bf16: null operand");
    if ((ldq | ldk | ldv) & 7 || (ldo & 3)) return fail(LDIT_EINVAL, "attention_bf16: row strides must be multiples of 8 (in) / 4 (out)");
    if (!aligned16(Q) || !aligned16(K) || !aligned16(V) || (reinterpret_cast<uintptr_t>(O) & (OUT_FP8 ? 3u : 7u)))
        return fail(LDIT_EINVAL, "attention_bf16: operands must be 16-byte aligned");
    const bool kt4 = diag().attn_bf16_kt4;
    constexpr int NW = 4;
    const int nqt = (N + 31) / 32;
    auto go_nw = [&](auto kern, int nw, int lds, std::atomic<unsigned long long> &attr_done) -> int {
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, attr_done)) return rc;
        const int nqg = (nqt + nw - 1) / nw;
        hipLaunchKernelGGL(kern, dim3((unsigned)(B * H * nqg)), dim3(nw * 64), lds, stream, static_cast<const bf16_t *>(Q),
                           static_cast<const bf16_t *>(K), static_cast<const bf16_t *>(V), O, N, H, ldq, ldk, ldv, ldo, scale, nqg,
                           qscale, lse);
        return LDIT_OK;
    };
    auto go = [&](auto kern, int lds, std::atomic<unsigned long long> &attr_done) -> int { return go_nw(kern, NW, lds, attr_done); };
    // LDS = two stages of (K image + V image).  64-key chunks: 32 KB and 128 VGPRs -> four workgroups per CU (four waves per
    // SIMD), measured 8-10 % faster than 128-key chunks at two per CU (LDIT_ATTN_BF16_KT=4) on N = 197 and N = 1025.
    // scale == 0: Q is pre-multiplied by scale * log2(e) (PRE, the packed inference path); otherwise the factor is applied here.
    static std::atomic<unsigned long long> set2{0}, set4{0}, set2p{0};      // per-device bookkeeping (ensure_dynamic_lds)
    // (eight query tiles per workgroup - half the DMA pieces and K/V traffic per tile - measured equal: 108.6 vs 107.8 us at N = 1025)
    // Round 3: five to eight query tiles (N = 129 .. 256: the 197 tokens of a 224 x 224 image) run as ONE eight-wave workgroup per
    // (image, head) - one staging of every chunk instead of two, no half-empty second workgroup: 27.3 -> 23.5 us at bs=64, 16.2 -> 14.7 us
    // at bs=32 (scripts/attn_bf16_bench.py, interleaved); long sequences stay on four-wave workgroups (N = 1025: 102.6 vs 105 us).  A
    // query tile's arithmetic does not depend on its workgroup: bit-identical either way (LDIT_ATTN_BF16_NW = 4 / 8 forces one).
    static std::atomic<unsigned long long> set2p8{0}, set28{0};
    const int force_nw = diag().attn_bf16_nw;
    const bool wide = force_nw == 8 || (force_nw != 4 && nqt > 4 && nqt <= 8);
    if (scale == 0.0f && wide) LDIT_TRY_RC(go_nw(attention_bf16<2, 8, OUT_FP8, true>, 8, 2 * 2 * 2 * 32 * KROWB, set2p8));
    else if (scale == 0.0f) LDIT_TRY_RC(go(attention_bf16<2, NW, OUT_FP8, true>, 2 * 2 * 2 * 32 * KROWB, set2p));
    else if (wide && !kt4) LDIT_TRY_RC(go_nw(attention_bf16<2, 8, OUT_FP8, false>, 8, 2 * 2 * 2 * 32 * KROWB, set28));
    else if (kt4) LDIT_TRY_RC(go(attention_bf16<4, NW, OUT_FP8, false>, 2 * 2 * 4 * 32 * KROWB, set4));
    else LDIT_TRY_RC(go(attention_bf16<2, NW, OUT_FP8, false>, 2 * 2 * 2 * 32 * KROWB, set2));
    LDIT_HIP_CHECK(hipGetLastError());
    return LDIT_OK;
}

int launch_attention_bf16(const void *Q, const void *K, const void *V, void *O, int B, int N, int H, int D, int ldq, int ldk,
                          int ldv, int ldo, float scale, hipStream_t stream)
{
    return launch_attn<false>(Q, K, V, O, B, N, H, D, ldq, ldk, ldv, ldo, scale, nullptr, nullptr, stream);
}

// train step: also writes the log2-domain log-sum-exp per (image, head, query) for attention_bwd_bf16
int launch_attention_bf16_lse(const void *Q, const void *K, const void *V, void *O, float *lse, int B, int N, int H, int D, int ldq,
                              int ldk, int ldv, int ldo, float scale, hipStream_t stream)
{
    if (!lse) return fail(LDIT_EINVAL, "attention_bf16: lse is null");
    return launch_attn<false>(Q, K, V, O, B, N, H, D, ldq, ldk, ldv, ldo, scale, nullptr, lse, stream);
}

int launch_attention_bf16_fp8out(const void *Q, const void *K, const void *V, void *O, int B, int N, int H, int D, int ldq,
                                 int ldk, int ldv, int ldo, float scale, const float *qscale, hipStream_t stream)
{
    if (!qscale) return fail(LDIT_EINVAL, "attention_bf16: fp8 output needs a scale");
    return launch_attn<true>(Q, K, V, O, B, N, H, D, ldq, ldk, ldv, ldo, scale, qscale, nullptr, stream);
}

}  // namespace ldit

#ifdef LDIT_GEMM_STAMPS
extern "C" int ldit_dbg_set_attn_stamps(void *buf)
{
    unsigned long long *p = static_cast<unsigned long long *>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_attn_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -3;
}
#endif
