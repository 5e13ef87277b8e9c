// bf16 MFMA GEMM whose operands are stored REDUCTION-MAJOR - the two GEMMs of a backward pass, without transposed copies:
//
//   dgrad   dX[M, Kout] = dY[M, Nred] . W[Nred, Kout]        A = dY  K-contiguous (as the forward's operands),
//                                                            B = W as nn.Linear stores it: [reduction rows][output cols]
//   wgrad   dW[Nout, Kout] = dY[tokens, Nout]^T . X[tokens, Kout]     BOTH operands [reduction rows = tokens][output cols]
//
// gemm_bf16.hip wants 8 consecutive reduction elements of one output row / column per lane (one ds_read_b128 of a
// K-contiguous LDS row).  Here a reduction-major operand is staged as it lies in memory - LDS image [64 reduction rows]
// [BT output columns], filled by LDS-DMA in whole 256 / 512-byte rows - and the MFMA fragment is gathered by
// ds_read_b64_tr_b16, gfx950's transposing LDS read (4 reduction rows x 16 columns per 16-lane group, two reads per
// 8-deep fragment, natural k order so it pairs with a K-contiguous operand on the other side).  The 64-byte granule of a
// row is XOR-swizzled with (reduction row & 3) - on the DMA source address and on the read - so the four rows a group
// touches fall on four different quarters of the 256-byte bank window (conflict-free; unswizzled they alias 4-way because
// the row stride is a multiple of 256 bytes).
// The matrix instruction is v_mfma_f32_32x32x16_bf16.  A v_mfma_f32_16x16x32_bf16 build of the main loop (the shape that bought
// gemm_bf16.hip 2 % inside the models; same sums bit for bit; lane group q reads reduction rows 8 q .. 8 q + 7 of a 32-deep step for
// one 16-column fragment, the two 32-byte halves of a granule swapped on rows with bit 3 set so the two groups of a 32-lane half -
// rows 8 apart, same 16 columns - stay conflict-free) passed the same tests and measured 0.5 - 1.2 % SLOWER in the train step
// (profiles/r04_tr_mfma16_ab.txt); it lived in this file as -DLDIT_TR_MFMA16 from commit 9cb824f to d318110.
// Reduction rows past the end (tokens are not a multiple of 64) are fetched from a page of zeros: LDS-DMA has no predication.
//
// Main loop (round 4: the pipeline of gemm_bf16.hip; rounds 2-3 ran a simple issue-all / read / multiply / barrier loop that left the
// matrix pipe idle through every DMA burst, every hand-over and the first fragment reads behind it - the dgrad of fc2 took 115 us
// where the forward GEMM of the same shape takes 72): two LDS stages, ONE barrier per 64-deep k-tile placed in front of the last
// 16-deep step, whose MFMAs run while the first fragments of the next tile are read and the DMA of the tile after it is issued
// into the stage the barrier just freed; inside a step the order reads | MFMAs | wait is pinned by hand (below); on the eight-wave tiles one wave per SIMD issues the pieces (its partner's MFMAs cover
// the issue stalls).  Same swapped-operand accumulator layout and epilogues as gemm_bf16.hip (fp32 / split-K slabs, bf16, dgrad x
// saved GELU derivative, column sums).
#include <cstdlib>
#include <type_traits>

#include "gemm_bf16_common.h"

namespace ldit {

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));

// TA: A operand reduction-major (wgrad) or K-contiguous (dgrad).  The W operand is always reduction-major.
template <int WM, int WN, int TM, int TN, int EPI, bool TA>
__global__ void __launch_bounds__(64 * WM * WN, (WM * WN) / 4) gemm_bf16_tr(const GemmArgsH p0)
{
    GemmArgsH p = p0;
    constexpr int NWAVES = WM * WN;
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int A_BYTES = BM * ROWB, STAGE = (BM + BN) * ROWB;
    static_assert((!TA || BM == 128 || BM == 256) && (BN == 128 || BN == 256), "reduction-major images need 256- or 512-byte rows");
    extern __shared__ __attribute__((aligned(128))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int nbn = (p.N + BN - 1) / BN, nbm = (p.M + BM - 1) / BM;
    const int ntiles = nbm * nbn, nblocks = ntiles * p.x.splits;
    int tile;
    {
        const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3, qq = nblocks >> 3, rr = nblocks & 7;
        tile = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    }
    int nk = (p.K + BKB - 1) / BKB, kbeg = 0;
    if (EPI == EPI_F32 && p.x.splits > 1) {
        const int split = tile / ntiles, per = (nk + p.x.splits - 1) / p.x.splits, kt0 = split * per;
        tile -= split * ntiles;
        nk = __builtin_amdgcn_readfirstlane(nk - kt0 < per ? nk - kt0 : per);      // scalar from here on, like kbeg (k_of below feeds the DMA's scalar base)
        kbeg = __builtin_amdgcn_readfirstlane(kt0 * BKB);      // scalar from here on (the split index comes out of a vector-unit division)
        p.Y = static_cast<float *>(p.Y) + (size_t)split * (size_t)p.M * (size_t)p.ldy;
    }
    const int m0 = (tile / nbn) * BM, n0 = (tile % nbn) * BN;

    // ---- DMA geometry of this lane's pieces (8 * piece < BM: A operand, else W).  Eight-wave tiles: waves 0..3 - one per SIMD -
    // issue all of them (piece = wave + 4 u), waves 4..7 none: a piece holds its issuing wave for tens of cycles, which the SIMD
    // partner's MFMAs cover (gemm_bf16.hip, profiles/r02_dma_issue.txt).  The 320-row tile keeps all eight waves issuing (registers).
    constexpr bool SPLIT = NWAVES == 8 && TM < 5;
    constexpr int LW = SPLIT ? 4 : NWAVES;
    constexpr int NLW = (BM + BN) / (8 * LW);
    static_assert((BM + BN) % (8 * LW) == 0, "DMA pieces must split evenly over the issuing waves");
    const bool loader = !SPLIT || wave < LW;                  // wave-uniform
    unsigned src[NLW];      // element offset of the lane's 16-byte chunk at reduction row 0 / k = 0 of the tile
    auto piece_of = [&](int u) { return (wave % LW) + LW * u; };
    // reduction row inside the k-tile of a reduction-major piece (only the ragged last k-tile asks)
    auto krow_of = [&](int u) {
        const int piece = piece_of(u);
        const bool isA = 8 * piece < BM;
        const int BT = isA ? BM : BN, cpr = BT / 8, pa = isA ? piece : piece - BM / 8;
        return pa * (1024 / (BT * 2)) + lane / cpr;
    };
#pragma unroll
    for (int u = 0; u < NLW; ++u) {
        const int piece = piece_of(u);
        if (!TA && 8 * piece < BM) {
            const int row = 8 * piece + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
            int gm = m0 + row;
            gm = gm < p.M ? gm : p.M - 1;
            src[u] = (unsigned)gm * (unsigned)p.lda + c * 8;
        } else {
            const bool isA = 8 * piece < BM;
            const int BT = isA ? BM : BN, cpr = BT / 8, pa = isA ? piece : piece - BM / 8;
            const int kr = pa * (1024 / (BT * 2)) + lane / cpr, chunk = lane % cpr;
            int col = (isA ? m0 : n0) + ((((chunk >> 2) ^ (kr & 3)) << 5) | ((chunk & 3) << 3));
            const int ncols = isA ? p.M : p.N;
            col = col + 8 <= ncols ? col : ncols - 8;         // columns past the matrix: duplicates, discarded by the epilogue
            src[u] = (unsigned)kr * (unsigned)(isA ? p.lda : p.ldw) + (unsigned)col;
        }
    }
    // every k-tile but a ragged last one: the tile's advance travels in a scalar base, the lane's offset is the kernel constant src[u]
    // (glds16h_sbase, gemm_bf16_common.h) - no vector address arithmetic per piece.  Pieces lo .. hi-1 of this wave's NLW.
    // (k0 is scalar - kbeg was made so at its definition, far from here - so the bases are scalar-unit arithmetic on kernel-argument
    // pointers: no v_readfirstlane result reaches the DMA's scalar base within its five wait states, cdna guide 5.7 item 2)
    auto issue_sbase = [&](int stage, int k0, int lo, int hi) {
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)((__attribute__((address_space(3))) char *)(smem + stage * STAGE)));
        const bf16_t *a_k = p.A + (unsigned)k0, *a_r = p.A + (size_t)k0 * (size_t)p.lda, *w_r = p.W + (size_t)k0 * (size_t)p.ldw;
#pragma unroll
        for (int u = 0; u < NLW; ++u) {
            if (u < lo || u >= hi) continue;
            const int piece = piece_of(u);
            const bool isA = 8 * piece < BM;
            glds16h_sbase((!TA && isA) ? a_k : isA ? a_r : w_r, 2u * src[u], dst + piece * 1024);
        }
    };
    // (the 320-row tile has no register left for the three scalar bases' set-up)
#ifndef LDIT_BF16_VADDR_DMA
    constexpr bool SBASE = TM < 5;
#else
    constexpr bool SBASE = false;
#endif
    auto issue = [&](int stage, int k0) {                     // k0: first reduction index of the k-tile
        char *base = smem + stage * STAGE;
        const bool ragged = k0 + BKB > p.K;                   // block-uniform: only the last k-tile of the matrix
        if (SBASE && !ragged) {
            issue_sbase(stage, k0, 0, NLW);
            return;
        }
#pragma unroll
        for (int u = 0; u < NLW; ++u) {
            const int piece = piece_of(u);
            const bf16_t *g;
            if (!TA && 8 * piece < BM) {
                g = p.A + (src[u] + (unsigned)k0);
            } else {
                const bool isA = 8 * piece < BM;
                const bf16_t *opnd = isA ? p.A : p.W;
                g = opnd + ((size_t)k0 * (size_t)(isA ? p.lda : p.ldw) + src[u]);
                if (ragged && k0 + krow_of(u) >= p.K) g = static_cast<const bf16_t *>(p.x.zeros) + 8 * (lane & 7);
            }
            glds16h_vaddr(g, (unsigned)(uintptr_t)((__attribute__((address_space(3))) char *)(base + piece * 1024)));
        }
    };

    constexpr bool L16 = false;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // ---- fragment addressing --------------------------------------------------------------------------------------------
    // K-contiguous A (dgrad): as gemm_bf16.hip.  Reduction-major image: lane (group g16 = lane>>4, i16 = lane&15) addresses
    // reduction row 16 s + 8 h + 4 t + (i16 >> 2), columns cb + 16 (g16 & 1) + 4 (i16 & 3) .. +3 and receives column
    // cb + (lane & 31) of reduction rows 16 s + 8 h + 4 t + 0..3.
    const int c32 = lane & 31, h = lane >> 5;
    const int g16 = lane >> 4, i16 = lane & 15, swz = (i16 >> 2) & 3;
    unsigned ta_addr[TA ? TM : 1], tw_addr[TN], a0 = 0;
    {
        constexpr int BPRA = BM * 2, BPRW = BN * 2;
        const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char *)smem);
        const int inrow = (16 * (g16 & 1) + 4 * (i16 & 3)) * 2;
        if (TA) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                ta_addr[i] = lds0 + (8 * h + (i16 >> 2)) * BPRA + ((((wm * TM + i)) ^ swz) << 6) + inrow;
        } else {
            // K-contiguous rows: chunk (2 s + h) ^ (row >> 1 & 7) of row wm TM 32 + c32 (+ 32 i: an immediate); the step's
            // "2 s" is an XOR of address bit 5 - every other term is a multiple of 128 bytes (aligned(128) above)
            a0 = lds0 + (wm * TM * 32 + c32) * ROWB + ((h ^ ((c32 >> 1) & 7)) << 4);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
            tw_addr[j] = lds0 + A_BYTES + (8 * h + (i16 >> 2)) * BPRW + ((((wn * TN + j)) ^ swz) << 6) + inrow;
    }
#ifndef LDIT_TR_DEALT_READS
    // two transposing reads: reduction rows 16 s + 8 h + {0..3} and + {4..7} (both offsets are immediates)
    auto tr_frag = [&](unsigned addr, auto off_c, auto bpr4_c) -> bf16x8 {
        union { s16x4 v[2]; bf16x8 f; } u;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(u.v[0]) : "v"(addr), "n"(decltype(off_c)::value));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(u.v[1]) : "v"(addr), "n"(decltype(off_c)::value + decltype(bpr4_c)::value));
        return u.f;
    };
#else
    // (every LDS-DMA of this kernel is an asm statement, so hipcc sees plain LDS reads here: no vmcnt guard, its own counted lgkmcnt)
    auto tr_frag = [&](unsigned addr, auto off_c, auto bpr4_c) -> bf16x8 {
        union { s16x4 v[2]; bf16x8 f; } u;
        typedef __attribute__((address_space(3))) s16x4 *lds_s16x4;
        u.v[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(uintptr_t)(addr + (unsigned)decltype(off_c)::value));
        u.v[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(uintptr_t)(addr + (unsigned)(decltype(off_c)::value + decltype(bpr4_c)::value)));
        return u.f;
    };
#endif
    // (issued in the order the next step's MFMAs need them - wb[0], xa[0], wb[1], xa[1], xa[2] ... for MFMAs (0,0) (0,1) (1,0) ... -
    //  so the first MFMA behind a step boundary waits for the OLDEST reads of the previous step)
    auto load_frags = [&](int stage, auto s_c, bf16x8(&xa)[TM], bf16x8(&wb)[TN]) {
        constexpr int s = decltype(s_c)::value;
        const unsigned so = (unsigned)(stage * STAGE);
        const unsigned addr = (a0 + so) ^ (unsigned)(s << 5);
        auto read_a = [&](int, auto i_c) {
            constexpr int I = decltype(i_c)::value;
            if (TA) {
                xa[I] = tr_frag(ta_addr[TA ? I : 0] + so, std::integral_constant<int, 16 * s * BM * 2>{}, std::integral_constant<int, 4 * BM * 2>{});
            } else {
#ifndef LDIT_TR_DEALT_READS
                const unsigned ad = addr;       // (an asm operand alone does not capture a variable of the enclosing lambda)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xa[I]) : "v"(ad), "n"(I * 32 * ROWB));
#else
                xa[I] = *(const __attribute__((address_space(3))) bf16x8 *)(uintptr_t)(addr + (unsigned)(I * 32 * ROWB));
#endif
            }
        };
        auto read_w = [&](auto j_c) {
            constexpr int J = decltype(j_c)::value;
            wb[J] = tr_frag(tw_addr[J] + so, std::integral_constant<int, 16 * s * BN * 2>{}, std::integral_constant<int, 4 * BN * 2>{});
        };
        static_assert(TN == 2 && TM >= 2 && TM <= 5, "read order written out for two column fragments");
        read_w(std::integral_constant<int, 0>{});
        read_a(0, std::integral_constant<int, 0>{});
        read_w(std::integral_constant<int, 1>{});
        read_a(1, std::integral_constant<int, 1>{});
        if constexpr (TM > 2) read_a(2, std::integral_constant<int, 2>{});
        if constexpr (TM > 3) read_a(3, std::integral_constant<int, 3>{});
        if constexpr (TM > 4) read_a(4, std::integral_constant<int, 4>{});
    };
    auto mfma_step = [&](const bf16x8(&xa)[TM], const bf16x8(&wb)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0);
    };

#ifndef LDIT_TR_DEALT_READS
    // the hand-over step: a DMA-issuing wave's pieces of tile kt+2 go out in one burst in front of its MFMAs.  -DLDIT_TR_DEAL deals
    // them between the MFMAs instead, pinned pair by pair (gemm_bf16.hip's arrangement): measured 0.7 % SLOWER in the train step on
    // one box (profiles/r04_tr_pinned_order_ab.txt), kept for the A/B.  Either way the MFMAs stay outside every branch - an
    // accumulator written on two paths costs the register allocator a copy of the tile (hundreds of spilled registers).
    auto mfma_step_dma = [&](const bf16x8(&xa)[TM], const bf16x8(&wb)[TN], int stage, int k0) {
        constexpr int NM = TM * TN, PPM = (NLW + NM - 1) / NM;
#ifdef LDIT_TR_DEAL
        const bool fast = SBASE && k0 + BKB <= p.K;           // block-uniform: every k-tile but a ragged last one
#else
        const bool fast = false;
#endif
        if (loader && !fast) issue(stage, k0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0);
                if (loader && fast) issue_sbase(stage, k0, (i * TN + j) * PPM, (i * TN + j + 1) * PPM);
                __builtin_amdgcn_sched_barrier(0);
            }
    };
#endif
    bf16x8 xa0[TM], wb0[TN], xa1[TM], wb1[TN];
    // k-tile kt of this block -> its first reduction index (past the end: the last tile again - fetched, never multiplied)
    auto k_of = [&](int kt) { return kbeg + (kt < nk ? kt : nk - 1) * BKB; };
    if (loader) {
        issue(0, k_of(0));
        issue(1, k_of(1));
    }
    // explicit: the first tile has landed before anybody reads it (vmcnt counts down in issue order: this wave's NLW pieces of the
    // second tile stay in flight under the first k-tile; the hand-over in front of its last step waits for them)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLW) : "memory");
    __syncthreads();
    constexpr std::integral_constant<int, 0> S0{};
    constexpr std::integral_constant<int, 1> S1{};
    constexpr std::integral_constant<int, 2> S2{};
    constexpr std::integral_constant<int, 3> S3{};
    load_frags(0, S0, xa0, wb0);
#ifdef LDIT_TR_DEALT_READS
    // Pinned pipeline, gemm_bf16.hip's: a step multiplies the fragments read during the previous step while the next step's are
    // read - the LDS reads DEALT between the step's first MFMAs (sched_group_barrier), hipcc's own counted lgkmcnt in front of each
    // MFMA that needs a fragment (it can count: every read is its own instruction, every LDS-DMA is asm and invisible to it).
    // Rounds 2 - 4a had the transposing reads as asm statements with hand waits: the scheduler sank them behind the MFMAs, right in
    // front of the wait (profiles/r04_tr_pinned_order_ab.txt); round 4b pinned reads | MFMAs | wait with sched_barrier - the burst
    // of reads in front of every step's MFMAs (both waves of a SIMD leave the hand-over barrier together) left the k-loop 17 % slower
    // per k than the forward GEMM's (scripts/tr_fit.py).  THIS build (-DLDIT_TR_DEALT_READS) closes that gap alone - dgrad 70 -> 62e-3
    // us per k on the 320-row tile, 17.4 -> 16.0e-3 on N = 768 - and LOSES 0.7 - 0.8 % inside the train step on two boxes
    // (profiles/r04_tr_pinned_order_ab.txt): the burst build stays the default, this one is kept for the A/B.
    constexpr int SG_MFMA = 0x8, SG_DSR = 0x100;
    // MFMAs and LDS reads per step; the reads are dealt over the FIRST HALF of the step's MFMAs (the second half covers the latency
    // of the last ones: the next step's first MFMA needs fragments from the middle of the read order)
#ifndef LDIT_TR_DEAL_SPAN
#define LDIT_TR_DEAL_SPAN 2
#endif
    constexpr int NM = TM * TN, NR = (TA ? 2 * TM : TM) + 2 * TN, NMD = NM / LDIT_TR_DEAL_SPAN > 0 ? NM / LDIT_TR_DEAL_SPAN : 1, RPM = (NR + NMD - 1) / NMD;
    auto deal = [&](auto id_c) {
        constexpr int ID = decltype(id_c)::value;
#pragma unroll
        for (int g = 0; g < NM; ++g) {
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, ID);
#pragma unroll
            for (int r = 0; r < RPM; ++r)
                if (g * RPM + r < NR) __builtin_amdgcn_sched_group_barrier(SG_DSR, 1, ID);
        }
    };
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        __builtin_amdgcn_sched_barrier(0);
        load_frags(cur, S1, xa1, wb1);
        mfma_step(xa0, wb0);
        deal(S0);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(cur, S2, xa0, wb0);
        mfma_step(xa1, wb1);
        deal(S1);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(cur, S3, xa1, wb1);
        mfma_step(xa0, wb0);
        deal(S2);
        __builtin_amdgcn_sched_barrier(0);
        // ---- hand-over: own DMA of tile kt+1 landed (vmcnt 0: the pieces are asm, hipcc does not wait for them), own reads of
        //      stage cur done (hipcc's lgkmcnt in front of the barrier), then all waves
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // ---- step 3: first fragments of tile kt+1 | the last fragments' MFMAs | DMA of tile kt+2 -> stage cur (free now)
        //      (the pieces are asm statements with a memory clobber - no LDS read moves across them - so they follow the step's MFMAs
        //       and reads in program order and go out while the MFMAs drain; they still have three steps to land)
        load_frags(cur ^ 1, S0, xa0, wb0);
        mfma_step(xa1, wb1);
        deal(S3);
        __builtin_amdgcn_sched_barrier(0);
        if (loader) issue(cur, k_of(kt + 2));
    }
#else
    // Every LDS read of the loop is asm (hipcc would guard the transposing builtin against the in-flight LDS-DMA with vmcnt(0), and
    // would count its own ds_read_b128 against transposing reads it cannot see), so the order is pinned by hand, sched_barrier by
    // sched_barrier: a step ISSUES its reads, then its MFMAs (on the previous step's fragments), then waits for the reads.  Left to
    // the scheduler (rounds 2 - 4a) the reads sank behind the MFMAs, right in front of the wait, and every step exposed a full LDS
    // latency that only the SIMD partner covered.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        // ---- steps 0 .. 2: read the next step's fragments, multiply the ones read during the previous step
        load_frags(cur, S1, xa1, wb1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(xa0, wb0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        load_frags(cur, S2, xa0, wb0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(xa1, wb1);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        load_frags(cur, S3, xa1, wb1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(xa0, wb0);
        __builtin_amdgcn_sched_barrier(0);
        // ---- hand-over: own DMA of tile kt+1 landed (vmcnt 0), own reads of stage cur done (lgkmcnt 0), then all waves
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // ---- step 3: first fragments of tile kt+1 | the last fragments' MFMAs | DMA of tile kt+2 -> stage cur (free now)
        load_frags(cur ^ 1, S0, xa0, wb0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_step_dma(xa1, wb1, cur, k_of(kt + 2));
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }

#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the re-fetches of the tail must not outlive the LDS allocation
    __syncthreads();                                      // every wave is out of the k-loop: the stage memory becomes slab buffers

    const bool cols_in = (n0 + BN <= p.N) && ((p.ldy & 3) == 0);
    // the slab path stores 16 bytes per lane: eight bf16 columns unless the output is fp32 (epilogue_rows.h)
    const bool slab_ok = f32_out<EPI>() || ((p.ldy & 7) == 0 && (EPI != EPI_GELU_BWD || (p.x.ldaux & 7) == 0) &&
                                            (p.x.nsplit_out == 0 || ((p.ldy / p.x.nsplit_out) & 7) == 0));
    const int mw = m0 + wm * TM * 32, nw = n0 + wn * TN * 32;
    if (cols_in && slab_ok && (m0 + BM <= p.M || p.x.colsum)) {
        // (with column sums requested the ragged last row tile takes the slab path too, rows past M skipped)
        float *cs = (!f32_out<EPI>() && p.x.colsum) ? p.x.colsum + (size_t)((m0 / BM) * WM + wm) * (size_t)p.ldy : nullptr;
        store_rows_via_lds<TM, TN, EPI, f32_out<EPI>() ? EPI_OUT_F32 : EPI_OUT_BF16, L16>(
            acc, smem + wave * EPI_WAVE_BYTES, p.Y, p.Y2, p.R, p.bias, p.lam, nullptr, p.ldy, mw, nw, lane, 1.0f, 1.0f, p.x, p.M, cs);
    } else if (cols_in) store_h<TM, TN, EPI, 1, L16>(p, acc, mw, nw, lane);
    else store_h<TM, TN, EPI, 2, L16>(p, acc, mw, nw, lane);
}

template <int WM, int WN, int TM, int TN, int EPI, bool TA>
int launch_tr(const GemmArgsH &a, hipStream_t stream)
{
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int lds = 2 * (BM + BN) * ROWB;
    static_assert(lds >= WM * WN * EPI_WAVE_BYTES, "epilogue slabs must fit the stage memory");
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    auto kern = gemm_bf16_tr<WM, WN, TM, TN, EPI, TA>;
    LDIT_DYN_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3(tiles * a.x.splits), dim3(64 * WM * WN), lds, stream, a);
    LDIT_HIP_CHECK(hipGetLastError());
    return LDIT_OK;
}

// The tile a launch takes for an [M, N] output with `splits` K-splits - the ONE place that decides (the launcher and the column-sum
// row count below both ask here).  2: 128 x 128 (4 waves, two workgroups per CU), 3: 256 x 256 (8 waves) when that fills most of a
// round, 6: 256 x 128 (wgrad form, forced only).  A K-contiguous A operand (dgrad) may also take the 192- / 320-row tiles (4 / 5)
// of gemm_bf16.hip against round quantisation: whole rounds of 256 workgroups, priced as rounds x tile height.
int tr_pick(long M, long N, int splits, bool ta)
{
    const long t256 = (long)splits * ((M + 255) / 256) * ((N + 255) / 256);
    int pick = t256 >= 120 ? 3 : 2;
#ifndef LDIT_TR_NO_TALL
    if (!ta && pick == 3) {
        double best = (double)((t256 + 255) / 256);
        for (int bm : {192, 320}) {
            const long t = (long)splits * ((M + bm - 1) / bm) * ((N + 255) / 256);
            const double c = (double)((t + 255) / 256) * (bm / 256.0) * 1.03;
            if (c < best) { best = c; pick = bm == 192 ? 4 : 5; }
        }
    }
#endif
    // (wgrad form, round 4: a 256 x 128 eight-wave tile - one workgroup per CU, one DMA-issuing wave per SIMD - is selectable
    //  (LDIT_GEMM_BF16_TR_TILE=6).  Alone it beats the 128 x 128 tiles at the same split counts by 1 - 3 %; inside the train step it
    //  lost 5 % (75.6 vs ~70 us per call: profiles/r04_wgrad_tiles.txt), so two four-wave workgroups per CU stay the choice.)
    if (const int force = diag().bf16_tr_tile; force >= 2 && (force <= (ta ? 3 : 5) || (ta && force == 6))) pick = force;
    return pick;
}

template <int EPI, bool TA>
int launch_tr_tiled(const GemmArgsH &a, hipStream_t stream)
{
    const int pick = tr_pick(a.M, a.N, a.x.splits, TA);
    if constexpr (TA) {
        if (pick == 6) return launch_tr<4, 2, 2, 2, EPI, TA>(a, stream);     // 256 x 128, 4 x 2 waves of 64 x 64
    }
    if constexpr (!TA) {
        if (pick == 4) return launch_tr<2, 4, 3, 2, EPI, TA>(a, stream);
        if (pick == 5) return launch_tr<2, 4, 5, 2, EPI, TA>(a, stream);
    }
    if (pick == 3) return launch_tr<2, 4, 4, 2, EPI, TA>(a, stream);
    return launch_tr<2, 2, 2, 2, EPI, TA>(a, stream);
}

}  // namespace

// Partial rows a dgrad launch with GemmExtra::colsum writes for an M-row output: row tiles x 2 wave rows; `bm` receives the tile
// height (tr_pick: the tile the launcher takes for this shape).
int gemm_bf16_tr_colsum_rows(int M, int N, int *bm)
{
    const int pick = tr_pick(M, N, 1, false);
    const int h = pick == 2 ? 128 : pick == 3 ? 256 : pick == 4 ? 192 : 320;
    if (bm) *bm = h;
    return ((M + h - 1) / h) * 2;
}

// Y[M, N] = epi( A . W ) with W [K rows (reduction), N cols] row stride ldw, and A either K-contiguous [M, K] (a_reduction_major
// = false; K % 64 == 0) or reduction-major [K rows, M cols] (true; any K, rows past K read x.zeros).  lda / ldw multiples of 8,
// M (when reduction-major) and N multiples of 8.  Epilogues: EPI_F32 (+ split-K), EPI_BIAS, EPI_GELU_BWD.
int launch_gemm_bf16_tr(const void *A, int lda, bool a_reduction_major, const void *W, int ldw, const float *bias, void *Y, int ldy,
                        int M, int N, int K, int epi, const GemmExtra &x, hipStream_t stream)
{
    if (M <= 0 || N <= 0 || K <= 0) return fail(LDIT_EINVAL, "gemm_bf16_tr: empty problem");
    if (!A || !W || !Y || !x.zeros) return fail(LDIT_EINVAL, "gemm_bf16_tr: null operand (a zero page is required)");
    if (!aligned16(A) || !aligned16(W) || !aligned16(x.zeros) || (lda & 7) || (ldw & 7) || (N & 7) || N < 8)
        return fail(LDIT_EINVAL, "gemm_bf16_tr: operands must be 16-byte aligned, strides and N multiples of 8");
    if (a_reduction_major ? ((M & 7) || M < 8) : (K % BKB != 0)) return fail(LDIT_EUNSUPPORTED, "gemm_bf16_tr: M %% 8 (reduction-major A) or K %% 64 (K-contiguous A) violated");
    const int nk = (K + BKB - 1) / BKB;
    if (x.splits < 1 || x.splits > nk || (x.splits > 1 && (epi != EPI_F32 || ((nk + x.splits - 1) / x.splits) * (x.splits - 1) >= nk)))
        return fail(LDIT_EINVAL, "gemm_bf16_tr: bad K-split %d", x.splits);
    GemmArgsH a{};
    a.A = static_cast<const bf16_t *>(A); a.W = static_cast<const bf16_t *>(W); a.Y = Y; a.bias = bias;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldy = ldy; a.x = x;
    if (x.colsum && (a_reduction_major || epi == EPI_F32 || x.splits != 1 || (N % 256) || ldy != N || !aligned16(x.colsum) || (x.aux && (x.ldaux & 7))))
        return fail(LDIT_EINVAL, "gemm_bf16_tr: column sums need the dgrad form with a bf16 output, N a multiple of 256 and ldy == N");
    if (a_reduction_major) {
        if (epi != EPI_F32) return fail(LDIT_EINVAL, "gemm_bf16_tr: wgrad form has the fp32 epilogue only");
        return launch_tr_tiled<EPI_F32, true>(a, stream);
    }
    switch (epi) {
        case EPI_F32: return launch_tr_tiled<EPI_F32, false>(a, stream);
        case EPI_BIAS: return launch_tr_tiled<EPI_BIAS, false>(a, stream);
        case EPI_GELU_BWD:
            if (!x.aux || (x.ldaux & 3)) return fail(LDIT_EINVAL, "gemm_bf16_tr: GELU-backward epilogue needs its factor operand");
            return launch_tr_tiled<EPI_GELU_BWD, false>(a, stream);
        default: return fail(LDIT_EINVAL, "gemm_bf16_tr: epilogue %d not available", epi);
    }
}

}  // namespace ldit
