// bf16 MFMA GEMM with fused epilogues:  Y[M,N] = epi( A[M,K] . W[N,K]^T ),  A, W bf16 (K-contiguous), fp32 accumulate.
// First bf16 build of the path (BASELINE configs 3-5 are bf16 / fp8; config 4 = ViT-L/16 512x512 bs=16 forward).
//
// Same skeleton as the fp32 kernels: K-contiguous operands -> LDS by LDS-DMA (global_load_lds_dwordx4), 128-B LDS rows
// (64 bf16), 16-B chunk XOR-swizzled with (row>>1)&7 on the source address and on the read, two stages, fragments of
// k-step s+1 read while step s multiplies.  v_mfma_f32_32x32x16_bf16: lane (r, h) supplies 8 consecutive k (16 B =
// ONE ds_read_b128) per 32-row tile and 16-deep step, so a 64-deep k-tile is four steps of TM*TN MFMAs.
// Operands are swapped in the MFMA (A-operand <- W rows, B-operand <- activation rows): an accumulator register quad
// holds 4 consecutive output columns of one row -> 16-B fp32 / 8-B bf16 stores.
// Block = 2 x 2 waves, wave tile TM x TN 32x32 tiles; 256x256 (TM = TN = 4) keeps the DMA issue (one 1-KiB piece per
// ~60 cycles per wave) under the MFMA time of a k-tile; smaller tiles are DMA-issue bound.
//
// Epilogues: bias -> bf16 ; bias + erf-GELU -> bf16 ; R + lam (.) (acc + bias) -> fp32 (in place on the fp32 residual
// stream, optional fp32 tap copy).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "gemm_bf16_common.h"

#ifdef LDIT_GEMM_STAMPS
// diagnostic build only (make dbg; scripts/gemm_bf16_stamps.py): per-workgroup cycles of wave 0 (a loader) and of wave NWAVES / 2
// (its SIMD partner) in gemm_bf16_mfma - 2 x 8 x int64: k-loop total, of which waiting for its own DMA (vmcnt) at the hand-over, of
// which waiting at the barrier, epilogue, kernel entry -> k-loop, k-loop in s_memrealtime ticks (100 MHz: the in-kernel clock)
__device__ unsigned long long *g_gemm_bf16_stamps = nullptr;
#endif

namespace ldit {

namespace {

// WM x WN waves per workgroup, each owning TM x TN 32x32 tiles.  2 x 4 waves (512 threads, two waves per SIMD) let one
// wave's MFMAs cover the other's LDS reads, DMA issue and barrier waits.
// One wave, one 32 x 16 output tile over all of K (the tail kernel's body - see gemm_bf16_tail below for what it promises; it stands
// in front of the tile kernel because a peeled tail now rides in the main launch as a few extra workgroups, each wave one tile).
template <int EPI>
__device__ __forceinline__ void tail_tile(const GemmArgsH &p, int tile, int lane)
{
    // A register set holds GPS GROUPS; a group = one 64-deep k-tile of one plane segment = two 32-deep steps (ordinary GEMM: one
    // segment, so groups are the k-tiles in order).  Split-fp32 operands (GemmExtra::nseg): the groups are walked in the tile kernels'
    // order (k-tile outermost when seg_inner, else segment outermost), so that a row gets the tile kernels' bits here too - which
    // lets small batches of those builds run on this latency-oriented kernel.
    constexpr int GPS = 4;                                // groups per register set (two sets: 16 steps of 32 in flight per wave)
    const int r16 = lane & 15, q16 = lane >> 4;
    const int nct = (p.N + 15) / 16;
    const int n0 = (tile % nct) * 16, m0 = (tile / nct) * 32;
    const bool two = m0 + 16 < p.M;                       // block-uniform: the second 16-row sub-tile has a valid row
    const int ra0 = m0 + r16 < p.M ? m0 + r16 : p.M - 1, ra1 = m0 + 16 + r16 < p.M ? m0 + 16 + r16 : p.M - 1;
    const int rw = n0 + r16 < p.N ? n0 + r16 : p.N - 1;
    const bf16_t *ap0 = p.A + (size_t)ra0 * p.lda + 8 * q16, *ap1 = p.A + (size_t)ra1 * p.lda + 8 * q16;
    const bf16_t *wp = p.W + (size_t)rw * p.ldw + 8 * q16;
    const int nkb = p.K / BKB, nseg = p.x.nseg > 0 ? p.x.nseg : 1, ngroups = nkb * nseg;
    f32x4 acc[2][1];
#pragma unroll
    for (int e = 0; e < 4; ++e) { acc[0][0][e] = 0.0f; acc[1][0][e] = 0.0f; }
    struct Set { bf16x8 a0[2 * GPS], a1[2 * GPS], w[2 * GPS]; };
    Set s0, s1;
    auto ld = [&](Set &s, int g0) {
#pragma unroll
        for (int gi = 0; gi < GPS; ++gi) {
            int G = g0 + gi < ngroups ? g0 + gi : ngroups - 1;            // clamped: a group past the end is loaded, never multiplied
            int seg = 0, kb = G;
            if (p.x.nseg > 1) {
                if (p.x.seg_inner) { kb = G / nseg; seg = G - kb * nseg; }
                else { seg = G / nkb; kb = G - seg * nkb; }
            }
            const unsigned ka = ((p.x.seg_a >> (4 * seg)) & 15u) * (unsigned)p.K + (unsigned)kb * BKB;
            const unsigned kw = ((p.x.seg_w >> (4 * seg)) & 15u) * (unsigned)p.K + (unsigned)kb * BKB;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                s.a0[2 * gi + st] = *reinterpret_cast<const bf16x8 *>(ap0 + ka + 32 * st);
                if (two) s.a1[2 * gi + st] = *reinterpret_cast<const bf16x8 *>(ap1 + ka + 32 * st);
                s.w[2 * gi + st] = *reinterpret_cast<const bf16x8 *>(wp + kw + 32 * st);
            }
        }
    };
    auto mm = [&](const Set &s, int g0) {
#pragma unroll
        for (int gi = 0; gi < GPS; ++gi)
            if (g0 + gi < ngroups)
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(s.w[2 * gi + st], s.a0[2 * gi + st], acc[0][0], 0, 0, 0);
                    if (two) acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(s.w[2 * gi + st], s.a1[2 * gi + st], acc[1][0], 0, 0, 0);
                }
    };
    ld(s0, 0);
    for (int g0 = 0; g0 < ngroups; g0 += 2 * GPS) {
        ld(s1, g0 + GPS);
        mm(s0, g0);
        ld(s0, g0 + 2 * GPS);
        mm(s1, g0 + GPS);
    }
    // rows of a skipped second sub-tile are >= M: the row check of the direct store drops them
    if ((n0 + 16 <= p.N) && ((p.ldy & 3) == 0)) store_h<1, 1, EPI, 1, true, 1>(p, acc, m0, n0, lane);
    else store_h<1, 1, EPI, 2, true, 1>(p, acc, m0, n0, lane);
}

template <int WM, int WN, int TM, int TN, int EPI>
__global__ void __launch_bounds__(64 * WM * WN, (WM * WN) / 4) gemm_bf16_mfma(const GemmArgsH p0)
{
    GemmArgsH p = p0;
    constexpr int NWAVES = WM * WN;
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, ROWS = BM + BN;
    static_assert(BM % 8 == 0, "a DMA piece is eight rows");
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifndef LDIT_BF16_MFMA32
    constexpr bool L16 = true;       // v_mfma_f32_16x16x32_bf16 (default)
#else
    constexpr bool L16 = false;      // A/B build: v_mfma_f32_32x32x16_bf16, as rounds 1-3 (same bits)
#endif

#ifdef LDIT_GEMM_STAMPS
    const unsigned long long st_entry = __builtin_amdgcn_s_memtime();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int nbn = (p.N + BN - 1) / BN, nbm = (p.M + BM - 1) / BM;
    const int ntiles = nbm * nbn, nblocks = ntiles * p.x.splits;
    if (p.x.tail_rows > 0 && (int)blockIdx.x >= nblocks) {
        // The peeled tail of THIS GEMM (launch_gemm_bf16_ex): rows M .. M + tail_rows - 1, one 32 x 16 tile per wave of the workgroups
        // behind the last tile - the same code, hence the same bits, as the separate gemm_bf16_tail launch it replaces (round 4:
        // one launch and one kernel boundary less per GEMM of ViT-L 512^2, M = 64 x 256 + 16).
        GemmArgsH t = p;
        const size_t r = (size_t)p.M;
        t.A = p.A + r * (size_t)p.lda;
        t.Y = static_cast<char *>(p.Y) + r * (size_t)p.ldy * (f32_out<EPI>() ? 4 : 2);
        if (p.R) t.R = p.R + r * (size_t)p.ldy;
        if (p.Y2) t.Y2 = p.Y2 + r * (size_t)p.ldy;
        if (p.x.Ypre) t.x.Ypre = static_cast<char *>(p.x.Ypre) + r * (size_t)p.ldy * 2;
        if (p.x.rowscale) t.x.rowscale = p.x.rowscale + r;
        if (p.x.aux) t.x.aux = static_cast<const char *>(p.x.aux) + r * (size_t)p.x.ldaux * 2;
        t.M = p.x.tail_rows;
        t.x.tail_rows = 0;
        // (ONE tile per workgroup, on its first wave: the tail chain is load-issue bound, eight of them on one CU ran 3 x slower)
        if (wave == 0) tail_tile<EPI>(t, (int)blockIdx.x - nblocks, lane);
        return;
    }
    int tile;
    {
        const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3, qq = nblocks >> 3, rr = nblocks & 7;
        tile = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    }
    int nk = p.K / BKB;
    if (EPI == EPI_F32 && p.x.splits > 1) {
        // split-K (wgrad: K = tokens of the batch, output tiles few): this block multiplies k-tiles [kt0, kt0 + nk) into its
        // own fp32 slab; ldit's reduce kernel adds the slabs in a fixed order
        const int split = tile / ntiles, per = (nk + p.x.splits - 1) / p.x.splits, kt0 = split * per;
        tile -= split * ntiles;
        nk = nk - kt0 < per ? nk - kt0 : per;
        p.A += kt0 * BKB;
        p.W += kt0 * BKB;
        p.Y = static_cast<float *>(p.Y) + (size_t)split * (size_t)p.M * (size_t)p.ldy;
    }
    const int m0 = (tile / nbn) * BM, n0 = (tile % nbn) * BN;
    // operand origins in scalar registers, made so HERE (the split-K offset above comes out of a vector-unit division): the DMA's
    // scalar base is formed from them by scalar adds only, far from this v_readfirstlane (cdna guide 5.7 item 2)
    const bf16_t *const A_s = uniform_ptr(p.A), *const W_s = uniform_ptr(p.W);

    // LDS-DMA roles.  With two waves per SIMD (the 8-wave tiles) only waves 0..3 - one per SIMD - issue DMA pieces, twice as
    // many each; waves 4..7 issue none.  A piece blocks the issuing wave's instruction stream for tens of cycles; with both
    // waves of a SIMD issuing at the 256 x 256 tile's density (4 pieces per 16 MFMAs and wave) the matrix pipe idles 46 % of
    // the time, with one loader per SIMD the partner's MFMAs cover the stall: 123 -> 70 cycles per MFMA against 64 without
    // any DMA (scripts/ubench/dma_issue.hip, profiles/r02_dma_issue.txt).  In the kernel the gain is smaller - the loader must
    // still reach every hand-over barrier: ViT-L/16 512x512 bs=16 forward 15.75 -> 15.40 ms (raising the loaders' priority
    // with s_setprio changes nothing).  The 4-wave tiles stay symmetric.
    // (the 320-row tile with the residual / fp32 epilogues has no registers left for the loaders' 18 source offsets: symmetric)
#ifdef LDIT_BF16_NO_SPLIT            // A/B build only (scripts/ab_bf16_split.sh): every wave issues its own pieces, as in round 1
    constexpr bool SPLIT = false;
#else
    constexpr bool SPLIT = NWAVES == 8 && !(TM == 5 && (EPI == EPI_SCALE_RESID || EPI == EPI_F32 || EPI == EPI_EMBED));
#endif
#ifdef LDIT_BF16_STAGGER
    // A/B build: all eight waves issue (half the pieces each); waves 0..3 right behind the hand-over, waves 4..7 - their SIMD
    // partners - a step or two later: one wave of a SIMD multiplies while the other is held up by its pieces
    constexpr bool STAGGER = SPLIT;
    constexpr int LW = NWAVES;
#else
    constexpr bool STAGGER = false;
    constexpr int LW = SPLIT ? NWAVES / 2 : NWAVES;      // waves that issue
#endif
    constexpr int NLW = ROWS / (8 * LW);                 // pieces per issuing wave and k-tile
    static_assert(ROWS % (8 * LW) == 0, "DMA pieces must split evenly over the issuing waves");
    const bool loader = !SPLIT || STAGGER || wave < LW;  // wave-uniform
    unsigned src[NLW];   // element (bf16) offsets
#pragma unroll
    for (int u = 0; u < NLW; ++u) {
        const int row = 8 * ((wave % LW) + LW * u) + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        if (8 * ((wave % LW) + LW * u) < BM) {
            int gm = m0 + row;
            gm = gm < p.M ? gm : p.M - 1;
            src[u] = (unsigned)gm * (unsigned)p.lda + c * 8;
        } else {
            int gn = n0 + row - BM;
            gn = gn < p.N ? gn : p.N - 1;
            src[u] = (unsigned)gn * (unsigned)p.ldw + c * 8;
        }
    }
    // k-tile t of the product -> column offsets into the A row and the W row.  Ordinary GEMM: both t * 64.  Split-fp32 operands
    // (GemmExtra::nseg): segment t / nk selects one bf16 plane of each operand, t % nk walks its K columns.  All wave-uniform.
    const int nseg = p.x.nseg > 0 ? p.x.nseg : 1;
    const int nkt = nk * nseg;
    auto tile_off = [&](int t, unsigned &ka, unsigned &kw) {
        t = t < nkt ? t : nkt - 1;
        int seg = 0;
        if (p.x.nseg > 1) {
            if (p.x.seg_inner) { const int kb = t / p.x.nseg; seg = t - kb * p.x.nseg; t = kb; }
            else { seg = t / nk; t -= seg * nk; }
        }
        // Made scalar HERE, a k-tile ahead of their use: hipcc evaluates the divisions above on the vector unit, and a v_readfirstlane
        // result must not reach the LDS-DMA's scalar base within five wait states (cdna guide 5.7 item 2) - issue_range, which only
        // ADDS these to the kernel-argument pointers on the scalar unit, runs hundreds of instructions later (prologue: s_nop below).
        ka = __builtin_amdgcn_readfirstlane((unsigned)(((p.x.seg_a >> (4 * seg)) & 15u) * (unsigned)p.K + (unsigned)t * BKB));
        kw = __builtin_amdgcn_readfirstlane((unsigned)(((p.x.seg_w >> (4 * seg)) & 15u) * (unsigned)p.K + (unsigned)t * BKB));
    };
    auto issue_range = [&](int stage, unsigned ka, unsigned kw, int lo, int hi) {
#ifdef LDIT_BF16_VADDR_DMA           // A/B build only: per-lane 64-bit source addresses through the builtin, as in rounds 1-2
        char *base = smem + stage * (ROWS * ROWB);
#pragma unroll
        for (int u = 0; u < NLW; ++u) {
            if (u < lo || u >= hi) continue;
            const int piece = (wave % LW) + LW * u;
            const bool isA = 8 * piece < BM;
            const bf16_t *opnd = isA ? p.A : p.W;
            glds16h(opnd + (src[u] + (isA ? ka : kw)), base + piece * 1024);
        }
#else
        // the k-tile's column offset travels in the scalar base, the lane's row / chunk offset is the kernel constant src[u]
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)((__attribute__((address_space(3))) char *)(smem + stage * (ROWS * ROWB))));
        // (scalar arithmetic on kernel-argument pointers and the scalar tile offsets: nothing here comes fresh out of a v_readfirstlane)
        const bf16_t *abase = A_s + ka, *wbase = W_s + kw;
#pragma unroll
        for (int u = 0; u < NLW; ++u) {
            if (u < lo || u >= hi) continue;
            const int piece = (wave % LW) + LW * u;
            glds16h_sbase(8 * piece < BM ? abase : wbase, 2u * src[u], dst + piece * 1024);
        }
#endif
    };

#ifndef LDIT_BF16_MFMA32
    // v_mfma_f32_16x16x32_bf16 build: the same LDS image, a 32 x 32 tile = 2 x 2 tiles of 16 x 16, lane (r16, q) supplies the 8
    // consecutive k of chunk q (of the four of a 32-deep step) for row r16 - ONE ds_read_b128 per 16-row fragment and 32-deep step.
    // A k-tile is four HALF steps: (k32, half) = (0, a) (0, b) (1, a) (1, b); a half multiplies TM of the 2 TM activation
    // fragments with all 2 TN weight fragments (16 MFMAs of 16 cycles for the 128 x 64 wave tile = the 8 x 32 cycles of a 16-deep
    // step of the 32x32x16 build).  MI355X_MICROARCH.md, DVFS give-back item 7: this shape holds a higher clock on random data.
    f32x4 acc[2 * TM][2 * TN];
#pragma unroll
    for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
        for (int j = 0; j < 2 * TN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.0f;
    const int r16 = lane & 15, q16 = lane >> 4;
    const int sw = (r16 >> 1) & 7;
    const int a_row = (wm * TM * 32 + r16) * ROWB, b_row = (BM + wn * TN * 32 + r16) * ROWB;
    // activation fragments of half `hf` (fragments hf * TM .. hf * TM + TM - 1) of 32-deep step k32
    auto load_a = [&](int stage, int k32, int hf, bf16x8(&xa)[TM]) {
        const char *base = smem + stage * (ROWS * ROWB) + ((k32 * 4 + q16) ^ sw) * 16;
#pragma unroll
        for (int i = 0; i < TM; ++i) xa[i] = *reinterpret_cast<const bf16x8 *>(base + a_row + (hf * TM + i) * 16 * ROWB);
    };
    auto load_w = [&](int stage, int k32, bf16x8(&wb)[2 * TN]) {
        const char *base = smem + stage * (ROWS * ROWB) + ((k32 * 4 + q16) ^ sw) * 16;
#pragma unroll
        for (int j = 0; j < 2 * TN; ++j) wb[j] = *reinterpret_cast<const bf16x8 *>(base + b_row + j * 16 * ROWB);
    };
    auto mfma_half = [&](int hf, const bf16x8(&xa)[TM], const bf16x8(&wb)[2 * TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 2 * TN; ++j) {
                if (hf == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0);
                else acc[TM + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[TM + i][j], 0, 0, 0);
            }
    };
#else
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int c32 = lane & 31, h = lane >> 5;
    const int sw = (c32 >> 1) & 7;
    const int a_row = (wm * TM * 32 + c32) * ROWB, b_row = (BM + wn * TN * 32 + c32) * ROWB;
    auto load_frags = [&](int stage, int s, bf16x8(&xa)[TM], bf16x8(&wb)[TN]) {
        const char *base = smem + stage * (ROWS * ROWB) + ((s * 2 + h) ^ sw) * 16;
#pragma unroll
        for (int i = 0; i < TM; ++i) xa[i] = *reinterpret_cast<const bf16x8 *>(base + a_row + i * 32 * ROWB);
#pragma unroll
        for (int j = 0; j < TN; ++j) wb[j] = *reinterpret_cast<const bf16x8 *>(base + b_row + j * 32 * ROWB);
    };
    auto mfma_step = [&](const bf16x8(&xa)[TM], const bf16x8(&wb)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0);
    };

#endif

    // Pinned pipeline (sched_group_barrier), four 16-deep steps per 64-deep k-tile:
    //   step s multiplies fragments read during step s-1 while the fragments of step s+1 are read, ONE LDS read or DMA
    //   piece per MFMA;  the hand-over barrier (tile kt+1 landed, stage cur released) sits in front of step 3's MFMAs;
    //   the NLD DMA pieces of a tile are spread over step 3 of the previous iteration (right after the hand-over that
    //   freed their stage) and steps 0 and 1, so the last of them still has ~2 steps of MFMA time to land.
    constexpr int SG_MFMA = 0x8, SG_VMEM = 0x20, SG_DSR = 0x100;
#ifndef LDIT_BF16_MFMA32
    constexpr int NM = 2 * TM * TN;                     // MFMAs per half step
    bf16x8 xaA[TM], xaB[TM], wbX[2 * TN], wbY[2 * TN];
#ifdef LDIT_GEMM_STAMPS
    unsigned long long st_vm = 0, st_bar = 0, st_loop0 = 0, st_real0 = 0;
#endif
    // one half step's schedule: `nr` LDS reads and `nd` DMA pieces dealt between its NM MFMAs, one per MFMA
    auto deal = [&](auto nr_c, auto nd_c, auto id_c) {
        constexpr int NR = decltype(nr_c)::value, ND = decltype(nd_c)::value, ID = decltype(id_c)::value;
        static_assert(NR + ND <= NM, "more reads and pieces than MFMAs in a half step");
#pragma unroll
        for (int g = 0; g < NR; ++g) {
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, ID);
            __builtin_amdgcn_sched_group_barrier(SG_DSR, 1, ID);
        }
#pragma unroll
        for (int g = 0; g < ND; ++g) {
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, ID);
            __builtin_amdgcn_sched_group_barrier(SG_VMEM, 1, ID);
        }
        if (NM - NR - ND > 0) __builtin_amdgcn_sched_group_barrier(SG_MFMA, NM - NR - ND, ID);
    };
    auto kloop = [&](auto np_c, auto late_c) {
    constexpr int NP = decltype(np_c)::value;
    constexpr int LATE = decltype(late_c)::value;        // 0: pieces dealt as always; 1 / 2: all of them in half step (0, a) / (0, b)
    constexpr int D3 = LATE ? 0 : (SPLIT ? NP : (NP + 2) / 3), D0 = LATE == 1 ? NP : LATE == 2 ? 0 : (SPLIT ? 0 : (NP - D3 + 1) / 2), D1 = NP - D3 - D0;
    unsigned ka0, kw0, ka1, kw1, ka2, kw2;
    tile_off(0, ka0, kw0);
    tile_off(1, ka1, kw1);
    asm volatile("s_nop 4" ::: "memory");
    issue_range(0, ka0, kw0, 0, NP);
    issue_range(1, ka1, kw1, 0, D3);
    // tile 0 has landed (vmcnt counts down in issue order); the D3 pieces of tile 1 stay in flight under the first k-tile - the
    // hand-over in front of its last half step waits for them as for every later tile
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D3) : "memory");
    __syncthreads();
#ifdef LDIT_GEMM_STAMPS
    st_loop0 = __builtin_amdgcn_s_memtime();
    st_real0 = __builtin_amdgcn_s_memrealtime();
#endif
    load_a(0, 0, 0, xaA);
    load_w(0, 0, wbX);
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        tile_off(kt + 2, ka2, kw2);
        __builtin_amdgcn_sched_barrier(0);
        // ---- half step (0, a): reads the activation fragments of (0, b)
        load_a(cur, 0, 1, xaB);
        issue_range(cur ^ 1, ka1, kw1, D3, D3 + D0);
        mfma_half(0, xaA, wbX);
        deal(std::integral_constant<int, TM>{}, std::integral_constant<int, (D0 < NM - TM ? D0 : NM - TM)>{}, std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        // ---- (0, b): reads all fragments of (1, a)
        load_a(cur, 1, 0, xaA);
        load_w(cur, 1, wbY);
        issue_range(cur ^ 1, ka1, kw1, D3 + D0, NP);
        mfma_half(1, xaB, wbX);
        deal(std::integral_constant<int, TM + 2 * TN>{}, std::integral_constant<int, (D1 < NM - TM - 2 * TN ? D1 : NM - TM - 2 * TN)>{}, std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        // ---- (1, a): reads the activation fragments of (1, b)
        load_a(cur, 1, 1, xaB);
        mfma_half(0, xaA, wbY);
        deal(std::integral_constant<int, TM>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
        __builtin_amdgcn_sched_barrier(0);
        // ---- hand-over
#ifdef LDIT_GEMM_STAMPS
        {
            const unsigned long long h0 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            const unsigned long long h1 = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const unsigned long long h2 = __builtin_amdgcn_s_memtime();
            st_vm += h1 - h0; st_bar += h2 - h1;
        }
#else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#endif
        // ---- (1, b): MFMAs of the last fragments | first fragments of tile kt+1 | first DMA pieces of tile kt+2 -> stage cur
        load_a(cur ^ 1, 0, 0, xaA);
        load_w(cur ^ 1, 0, wbX);
        issue_range(cur, ka2, kw2, 0, D3);
        ka1 = ka2; kw1 = kw2;
        mfma_half(1, xaB, wbY);
        deal(std::integral_constant<int, TM + 2 * TN>{}, std::integral_constant<int, (D3 < NM - TM - 2 * TN ? D3 : NM - TM - 2 * TN)>{}, std::integral_constant<int, 3>{});
    }
    };
#else
    constexpr int NM = TM * TN, NF = TM + TN;
    static_assert(NF <= NM, "fewer MFMAs than fragment reads per step");
    bf16x8 xa0[TM], wb0[TN], xa1[TM], wb1[TN];
#ifdef LDIT_GEMM_STAMPS
    unsigned long long st_vm = 0, st_bar = 0, st_loop0 = 0, st_real0 = 0;
#endif
    // the k-loop, instantiated per role: NP = DMA pieces this wave issues per k-tile (0 for the non-loaders of a SPLIT tile)
    auto kloop = [&](auto np_c) {
    constexpr int NP = decltype(np_c)::value;
    // pieces issued in steps 3 / 0 / 1; a SPLIT loader issues all of them right behind the hand-over (twice the pieces: the last
    // ones need the whole k-tile to land)
    constexpr int D3 = SPLIT ? NP : (NP + 2) / 3, D0 = SPLIT ? 0 : (NP - D3 + 1) / 2, D1 = NP - D3 - D0;
    unsigned ka0, kw0, ka1, kw1, ka2, kw2;
    tile_off(0, ka0, kw0);
    tile_off(1, ka1, kw1);
    asm volatile("s_nop 4" ::: "memory");                 // the only place where a tile offset is used right after it was made scalar
    issue_range(0, ka0, kw0, 0, NP);
    issue_range(1, ka1, kw1, 0, D3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // explicit: the first tile has landed before anybody reads it
    __syncthreads();
#ifdef LDIT_GEMM_STAMPS
    st_loop0 = __builtin_amdgcn_s_memtime();
    st_real0 = __builtin_amdgcn_s_memrealtime();
#endif
    load_frags(0, 0, xa0, wb0);
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        tile_off(kt + 2, ka2, kw2);
        __builtin_amdgcn_sched_barrier(0);
        // ---- step 0
        load_frags(cur, 1, xa1, wb1);
        issue_range(cur ^ 1, ka1, kw1, D3, D3 + D0);
        mfma_step(xa0, wb0);
#pragma unroll
        for (int g = 0; g < NF; ++g) {
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
            __builtin_amdgcn_sched_group_barrier(SG_DSR, 1, 0);
        }
#pragma unroll
        for (int g = 0; g < D0; ++g) {
            if (NF + g < NM) __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
            __builtin_amdgcn_sched_group_barrier(SG_VMEM, 1, 0);
        }
        if (NM - NF - D0 > 0) __builtin_amdgcn_sched_group_barrier(SG_MFMA, NM - NF - D0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // ---- step 1
        load_frags(cur, 2, xa0, wb0);
        issue_range(cur ^ 1, ka1, kw1, D3 + D0, NP);
        mfma_step(xa1, wb1);
#pragma unroll
        for (int g = 0; g < NF; ++g) {
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 1);
            __builtin_amdgcn_sched_group_barrier(SG_DSR, 1, 1);
        }
#pragma unroll
        for (int g = 0; g < D1; ++g) {
            if (NF + g < NM) __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 1);
            __builtin_amdgcn_sched_group_barrier(SG_VMEM, 1, 1);
        }
        if (NM - NF - D1 > 0) __builtin_amdgcn_sched_group_barrier(SG_MFMA, NM - NF - D1, 1);
        __builtin_amdgcn_sched_barrier(0);
        // ---- step 2
        load_frags(cur, 3, xa1, wb1);
        mfma_step(xa0, wb0);
#pragma unroll
        for (int g = 0; g < NF; ++g) {
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 2);
            __builtin_amdgcn_sched_group_barrier(SG_DSR, 1, 2);
        }
        if (NM - NF > 0) __builtin_amdgcn_sched_group_barrier(SG_MFMA, NM - NF, 2);
        __builtin_amdgcn_sched_barrier(0);
        // ---- hand-over: own DMA of tile kt+1 landed (vmcnt 0), own reads of stage cur done, then all waves
#ifdef LDIT_GEMM_STAMPS
        {
            const unsigned long long h0 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            const unsigned long long h1 = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const unsigned long long h2 = __builtin_amdgcn_s_memtime();
            st_vm += h1 - h0; st_bar += h2 - h1;
        }
#else
        // explicit: hipcc does NOT put a vmcnt wait in front of this barrier for LDS-DMA pieces issued behind the PREVIOUS
        // hand-over (a loader's pieces all are) - other waves would read a stage that has not landed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#endif
        // ---- step 3: MFMAs of the last fragments | first fragments of tile kt+1 | first DMA pieces of tile kt+2 -> stage cur
        load_frags(cur ^ 1, 0, xa0, wb0);
        issue_range(cur, ka2, kw2, 0, D3);
        ka1 = ka2; kw1 = kw2;
        mfma_step(xa1, wb1);
#pragma unroll
        for (int g = 0; g < NF; ++g) {
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 3);
            __builtin_amdgcn_sched_group_barrier(SG_DSR, 1, 3);
        }
#pragma unroll
        for (int g = 0; g < D3; ++g) {
            if (NF + g < NM) __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 3);
            __builtin_amdgcn_sched_group_barrier(SG_VMEM, 1, 3);
        }
        if (NM - NF - D3 > 0) __builtin_amdgcn_sched_group_barrier(SG_MFMA, NM - NF - D3, 3);
    }
    };
#endif
#ifndef LDIT_BF16_MFMA32
#ifndef LDIT_BF16_STAGGER_STEP
#define LDIT_BF16_STAGGER_STEP 2
#endif
    if (STAGGER && wave >= NWAVES / 2) kloop(std::integral_constant<int, NLW>{}, std::integral_constant<int, LDIT_BF16_STAGGER_STEP>{});
    else if (loader) kloop(std::integral_constant<int, NLW>{}, std::integral_constant<int, 0>{});
    else kloop(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
#else
    if (loader) kloop(std::integral_constant<int, NLW>{});
    else kloop(std::integral_constant<int, 0>{});
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped re-fetches of the tail must not outlive the LDS allocation
#ifdef LDIT_GEMM_STAMPS
    asm volatile("s_nop 0" :: "v"(acc[0][0][0]), "v"(acc[(L16 ? 2 : 1) * TM - 1][(L16 ? 2 : 1) * TN - 1][L16 ? 3 : 15]));
    const unsigned long long st_loop1 = __builtin_amdgcn_s_memtime(), st_real1 = __builtin_amdgcn_s_memrealtime();
#endif

    const bool cols_in = (n0 + BN <= p.N) && ((p.ldy & 3) == 0);
    // the slab path stores 16 bytes per lane: eight bf16 columns unless the output is fp32 (epilogue_rows.h)
    const bool slab_ok = f32_out<EPI>() || ((p.ldy & 7) == 0 && (EPI != EPI_GELU_BWD || (p.x.ldaux & 7) == 0) &&
                                            (p.x.nsplit_out == 0 || ((p.ldy / p.x.nsplit_out) & 7) == 0));
    const int mw = m0 + wm * TM * 32, nw = n0 + wn * TN * 32;
    if (cols_in && slab_ok && m0 + BM <= p.M && !p.direct_epi) {
        __syncthreads();     // every wave is out of the k-loop (and its DMA drained): the stage memory becomes slab buffers
        store_rows_via_lds<TM, TN, EPI, f32_out<EPI>() ? EPI_OUT_F32 : (EPI == EPI_GELU_SPLIT || EPI == EPI_BIAS_SPLIT) ? EPI_OUT_SPLIT : EPI_OUT_BF16, L16>(
            acc, smem + wave * EPI_WAVE_BYTES, p.Y, p.Y2, p.R, p.bias, p.lam, nullptr, p.ldy, mw, nw, lane, 1.0f, 1.0f, p.x);
    } else if (cols_in && m0 + BM <= p.M) store_h<TM, TN, EPI, 0, L16>(p, acc, mw, nw, lane);
    else if (cols_in) store_h<TM, TN, EPI, 1, L16>(p, acc, mw, nw, lane);
    else store_h<TM, TN, EPI, 2, L16>(p, acc, mw, nw, lane);
#ifdef LDIT_GEMM_STAMPS
    if (g_gemm_bf16_stamps && (wave == 0 || wave == NWAVES / 2) && lane == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long *d = g_gemm_bf16_stamps + (size_t)blockIdx.x * 16 + (wave == 0 ? 0 : 8);
        d[0] = st_loop1 - st_loop0; d[1] = st_vm; d[2] = st_bar; d[3] = __builtin_amdgcn_s_memtime() - st_loop1;
        d[4] = st_loop0 - st_entry; d[5] = st_real1 - st_real0;
    }
#endif
}

template <int WM, int WN, int TM, int TN, int EPI>
int launch_h(const GemmArgsH &a, hipStream_t stream)
{
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int lds = 2 * (BM + BN) * ROWB;
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    auto kern = gemm_bf16_mfma<WM, WN, TM, TN, EPI>;
    LDIT_DYN_LDS(kern, lds);
    // (+ one workgroup per 32 x 16 tile of a peeled tail riding in this launch)
    const int tail_tiles = a.x.tail_rows > 0 ? ((a.N + 15) / 16) * ((a.x.tail_rows + 31) / 32) : 0;
    const int tail_blocks = tail_tiles;
    hipLaunchKernelGGL(kern, dim3(tiles * a.x.splits + tail_blocks), dim3(64 * WM * WN), lds, stream, a);
    LDIT_HIP_CHECK(hipGetLastError());
    return LDIT_OK;
}

// ---- tail kernel: up to 64 rows (the ragged tail peeled off by launch_gemm_bf16_ex, or a whole problem that small) ----------
// Round 3: BIT-IDENTICAL to the tile kernels.  A 16-row remainder (M = 16 x 1025 = 64 x 256 + 16) on the tile kernel costs a
// whole extra round of the machine; the round-1/2 answer was a split-K "skinny" kernel, whose eight K slices summed in another
// order than a tile's k-loop, so the rows it served (the LAST image of a batch) depended on the batch layout - the strict
// permutation test had to be relaxed for them - and it cost 11 - 38 us per call (9.6 % of the ViT-L step).  Here one wave owns
// one output tile and walks ALL of K in order with the tile kernels' own MFMA, the operands in the tile kernels' lane positions,
// into ONE accumulation chain per output element: per output element exactly the tile kernels' sequence of MFMAs, hence its bits.
// Fragments come straight from global memory (nothing is reused inside a wave; A is shared through L1 / L2 by the waves of a row
// block, W by nobody).  The epilogue is the tiles' direct store (store_h), spelled with the same fmas.
// Round 4 (with the tile kernels): v_mfma_f32_16x16x32_bf16, a wave = 32 rows x 16 columns (two row sub-tiles; the second one is
// skipped altogether when it lies past M - the peeled 16-row tail).  A load instruction now covers 16 rows x 64 contiguous bytes
// (32 rows x 32 B before) and a 32-deep step costs two or three loads instead of four: the kernel is load-issue bound (~94 cycles per
// pair of loads), so the per-wave chain - what its 6 - 17 us were made of - halves, and twice as many waves share the work.
template <int EPI>
__global__ void __launch_bounds__(64) gemm_bf16_tail(const GemmArgsH p)
{
    tail_tile<EPI>(p, (int)blockIdx.x, (int)threadIdx.x);
}

template <int EPI>
int launch_tail(const GemmArgsH &a, hipStream_t stream)
{
    const unsigned blocks = (unsigned)(((a.N + 15) / 16) * ((a.M + 31) / 32));
    hipLaunchKernelGGL(gemm_bf16_tail<EPI>, dim3(blocks), dim3(64), 0, stream, a);
    LDIT_HIP_CHECK(hipGetLastError());
    return LDIT_OK;
}

template <int EPI>
int launch_h_tiled(const GemmArgsH &a, hipStream_t stream)
{
    if (a.M <= 64 && a.x.splits == 1 && !diag().bf16_tile_env) return launch_tail<EPI>(a, stream);
    // Split-fp32 operands at serving sizes: the plane products make the k-loop 3 - 6 x as deep (K' = 2 304 ... 18 432) while a few
    // images give the tile kernels only a handful of tiles - twelve 128 x 128 tiles walking K' = 9 216 took 170 us.  The one-wave-per-
    // 32 x 32-tile kernel spreads the same work over hundreds of waves (bit-identical rows).  It re-reads its operands from L2 per
    // wave, so it only pays while there are few of them: up to 700 waves (scripts/latency_bench.py, f32x3 at bs = 1 / 2 / 4: 3.03 /
    // 3.04 / 3.07 ms on the tile kernels alone, 1.49 / 2.24 / 2.77 ms with this rule; 6 144 waves: 1.49 / 2.42 / 3.83 ms).
    if (a.x.nseg > 1 && a.x.splits == 1 && !diag().bf16_tile_env && (long)((a.M + 31) / 32) * ((a.N + 31) / 32) <= diag().planes_tail_waves)
        return launch_tail<EPI>(a, stream);
    // Time model fitted to scripts/gemm_bf16_bench.py on ViT-B / ViT-L shapes, M = 3 k .. 25 k (profiles/README.md), in us:
    //   256 x 256, 8 waves (one workgroup per CU):  strict rounds of 256 tiles, each  a[epi] + 19.5e-3 K
    //   128 x 128 (two per CU, they overlap each other's prologue / epilogue):  rounds of 256 tiles, each  r[epi] + 7.6e-3 K,
    //   but never less than one tile's own latency  5 + 11.4e-3 K.
    // Measured and dropped (profiles/README.md): a four-stage and a five-stage counted-vmcnt ring, a 256 x 128 ring at two
    // workgroups per CU, a barrier-phased ping-pong of the two waves per SIMD, a v_mfma_f32_16x16x32_bf16 build and a
    // four-wave 256 x 256 tile - none beat this kernel on any shape; the 256 x 128 tile stays selectable for experiments.
    // (EPI_EMBED and EPI_F32 priced like the residual epilogue, EPI_GELU_BWD like the GELU one)
    const double a256[8] = {19.0, 18.5, 25.0, 25.0, 25.0, 18.5, 25.0, 25.0}, r128[8] = {5.2, 5.0, 7.7, 7.7, 7.7, 5.0, 7.7, 7.7};
    const long sp = a.x.splits;
    const int depth = a.x.nseg > 1 ? a.K * a.x.nseg : a.K;     // k-depth of one output element (all plane segments)
    const long t256 = sp * ((a.M + 255) / 256) * ((a.N + 255) / 256), t128 = sp * ((a.M + 127) / 128) * ((a.N + 127) / 128);
    const double Ks = (double)depth / (double)sp;           // k-depth one block walks
    const double c256 = (double)((t256 + 255) / 256) * (a256[EPI] + 19.5e-3 * Ks);
    double c128 = (double)((t128 + 255) / 256) * (r128[EPI] + 7.6e-3 * Ks);
    if (c128 < 5.0 + 11.4e-3 * Ks) c128 = 5.0 + 11.4e-3 * Ks;
    int pick = c256 <= c128 ? 3 : 2;
    // Round 2: 192- and 320-row variants of the 8-wave tile against ROUND QUANTISATION - a launch is whole rounds of 256
    // workgroups, and M = 64 x 197 rows makes 150 / 450 / 600 tiles of 256 x 256 for N = 768 / 2304 / 3072 (0.59 / 1.76 / 2.34
    // rounds).  192 x 256 gives 198 tiles for N = 768 (one round, 77 % of the CUs busy instead of 59 %, each with 3/4 of
    // the work); 320 x 256 gives 480 tiles for N = 3072 (2 rounds of 1.25 instead of 3 of 1.0).  Same kernel, same k-order
    // per output element (bit-identical results), priced as rounds x tile height.
    double best = c256 <= c128 ? c256 : c128;
    const double per256 = a256[EPI] + 19.5e-3 * Ks;
    for (int bm : {192, 320}) {
        const long t = sp * ((a.M + bm - 1) / bm) * ((a.N + 255) / 256);
        const double c = (double)((t + 255) / 256) * per256 * (bm / 256.0) * 1.03;      // 3 % handicap: prefer the fitted tiles on ties
        if (c < best) { best = c; pick = bm == 192 ? 4 : 5; }
    }
    if (const int force = diag().bf16_tile; force >= 2 && force <= 5) pick = force;
    switch (pick) {
        case 3: return launch_h<2, 4, 4, 2, EPI>(a, stream);     // 256 x 256, 8 waves (2 per SIMD)
        case 4: return launch_h<2, 4, 3, 2, EPI>(a, stream);     // 192 x 256
        case 5: return launch_h<2, 4, 5, 2, EPI>(a, stream);     // 320 x 256
        default: return launch_h<2, 2, 2, 2, EPI>(a, stream);    // 128 x 128, 4 waves
    }
}

// fp32 -> bf16 (round to nearest even), n elements, 4 per thread
// (mul: pack-time factor - 1, or scale * log2 e on W_q for the pre-scaled attention; the product is rounded ONCE)
__global__ void __launch_bounds__(256) cvt_f32_bf16(const float *__restrict__ src, bf16_t *__restrict__ dst, size_t n, float mul)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(src + i);
        const bf16x4 o = {(bf16_t)(v[0] * mul), (bf16_t)(v[1] * mul), (bf16_t)(v[2] * mul), (bf16_t)(v[3] * mul)};
        *reinterpret_cast<bf16x4 *>(dst + i) = o;
    } else {
        for (size_t k = i; k < n; ++k) dst[k] = (bf16_t)(src[k] * mul);
    }
}

// fp32 [rows, cols] (row stride lds) -> `S` bf16 planes side by side, [rows, S * cols]: x ~= p0 + p1 (+ p2).  4 elements per thread.
template <int S>
__global__ void __launch_bounds__(256) split_planes(const float *__restrict__ src, int lds, bf16_t *__restrict__ dst, int rows, int cols,
                                                    float mul)
{
    const int c4 = cols >> 2;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)rows * c4) return;
    const int r = (int)(i / c4), c = (int)(i - (size_t)r * c4) * 4;
    f32x4 v = *reinterpret_cast<const f32x4 *>(src + (size_t)r * lds + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= mul;         // pack-time factor (1, or scale * log2 e on W_q): rounded once, then split
    bf16_t *d = dst + (size_t)r * (S * cols) + c;
#pragma unroll
    for (int sp = 0; sp < S; ++sp) {
        const bf16x4 pk = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        *reinterpret_cast<bf16x4 *>(d + sp * cols) = pk;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] -= (float)pk[e];
    }
}

}  // namespace

int launch_split_planes(const float *src, int lds, void *dst, int rows, int cols, int planes, hipStream_t stream, float mul)
{
    if (rows <= 0 || cols <= 0) return LDIT_OK;
    if (!src || !dst || (cols & 3) || (lds & 3) || lds < cols || !aligned16(src) || (reinterpret_cast<uintptr_t>(dst) & 7u))
        return fail(LDIT_EINVAL, "split_planes: null / misaligned operand or cols not a multiple of 4");
    const unsigned blocks = (unsigned)(((size_t)rows * (cols >> 2) + 255) / 256);
    if (planes == 2) hipLaunchKernelGGL(split_planes<2>, dim3(blocks), dim3(256), 0, stream, src, lds, static_cast<bf16_t *>(dst), rows, cols, mul);
    else if (planes == 3) hipLaunchKernelGGL(split_planes<3>, dim3(blocks), dim3(256), 0, stream, src, lds, static_cast<bf16_t *>(dst), rows, cols, mul);
    else return fail(LDIT_EINVAL, "split_planes: %d planes (2 or 3)", planes);
    LDIT_HIP_CHECK(hipGetLastError());
    return LDIT_OK;
}

static int launch_gemm_bf16_one(const void *A, int lda, const void *W, const float *bias, void *Y, int ldy, int M, int N, int K,
                                int epi, const float *lam, const float *R, float *Y2, const GemmExtra &x, hipStream_t stream);

// A few rows past a multiple of the 256-row tile (M = 16 x 1025 = 64 x 256 + 16) would cost a whole extra row of
// workgroups - a full extra round of the machine for N = 1024.  Such a ragged tail is peeled off into a second, tiny
// launch (gemm_bf16_tail: one wave per 32 x 32 tile over all of K, the tile kernels' own MFMA sequence per output element);
// the main part then fills 256 CUs in whole rounds.  Peeled rows get the bits a tile would have given them.
int launch_gemm_bf16_ex(const void *A, int lda, const void *W, const float *bias, void *Y, int ldy, int M, int N, int K, int epi,
                        const float *lam, const float *R, float *Y2, const GemmExtra &x, hipStream_t stream)
{
    const int rem = M % 256;
    const long nbn = (N + 255) / 256, full = ((long)M / 256 + 1) * nbn, mainp = ((long)M / 256) * nbn;
    // (the patch-embedding epilogue maps rows by their GLOBAL index: never peeled)
    // (nor is a split-fp32 product: the tail kernel walks one K range)
    if (rem != 0 && rem <= 64 && M > 256 && x.splits == 1 && epi != EPI_EMBED && x.nseg == 0 && (full + 255) / 256 > (mainp + 255) / 256) {   // peel only when it saves a round
        const int main_rows = M - rem;
        if (!diag().bf16_tail_launch) {
            // the tail rides in the main launch (extra workgroups behind the last tile, gemm_bf16_mfma); LDIT_GEMM_BF16_TAIL_LAUNCH=1
            // keeps the two launches of round 3 for the A/B - the rows get the same bits either way
            GemmExtra xm = x;
            xm.tail_rows = rem;
            return launch_gemm_bf16_one(A, lda, W, bias, Y, ldy, main_rows, N, K, epi, lam, R, Y2, xm, stream);
        }
        const size_t out_elt = (epi == EPI_SCALE_RESID || epi == EPI_F32) ? 4 : 2;
        int rc = launch_gemm_bf16_one(A, lda, W, bias, Y, ldy, main_rows, N, K, epi, lam, R, Y2, x, stream);
        if (rc != LDIT_OK) return rc;
        const char *At = static_cast<const char *>(A) + (size_t)main_rows * lda * 2;
        char *Yt = static_cast<char *>(Y) + (size_t)main_rows * ldy * out_elt;
        GemmExtra xt = x;
        if (x.Ypre) xt.Ypre = static_cast<char *>(x.Ypre) + (size_t)main_rows * ldy * 2;
        if (x.rowscale) xt.rowscale = x.rowscale + main_rows;
        if (x.aux) xt.aux = static_cast<const char *>(x.aux) + (size_t)main_rows * x.ldaux * 2;
        return launch_gemm_bf16_one(At, lda, W, bias, Yt, ldy, rem, N, K, epi, lam, R ? R + (size_t)main_rows * ldy : nullptr,
                                    Y2 ? Y2 + (size_t)main_rows * ldy : nullptr, xt, stream);
    }
    return launch_gemm_bf16_one(A, lda, W, bias, Y, ldy, M, N, K, epi, lam, R, Y2, x, stream);
}

int launch_gemm_bf16(const void *A, int lda, const void *W, const float *bias, void *Y, int ldy, int M, int N, int K, int epi,
                     const float *lam, const float *R, float *Y2, hipStream_t stream)
{
    return launch_gemm_bf16_ex(A, lda, W, bias, Y, ldy, M, N, K, epi, lam, R, Y2, GemmExtra{}, stream);
}

static int launch_gemm_bf16_one(const void *A, int lda, const void *W, const float *bias, void *Y, int ldy, int M, int N, int K,
                                int epi, const float *lam, const float *R, float *Y2, const GemmExtra &x, hipStream_t stream)
{
    if (M <= 0 || N <= 0 || K <= 0) return fail(LDIT_EINVAL, "gemm_bf16: empty problem");
    if (K % BKB) return fail(LDIT_EUNSUPPORTED, "gemm_bf16: K=%d must be a multiple of %d", K, BKB);
    if (!A || !W || !Y) return fail(LDIT_EINVAL, "gemm_bf16: null operand");
    if (!aligned16(A) || !aligned16(W) || (lda & 7)) return fail(LDIT_EINVAL, "gemm_bf16: operands must be 16-byte aligned");
    if (x.splits < 1 || x.splits > K / BKB || (x.splits > 1 && epi != EPI_F32))
        return fail(LDIT_EINVAL, "gemm_bf16: %d K-splits need the fp32 slab epilogue and at least one k-tile each", x.splits);
    if (x.splits > 1 && ((K / BKB + x.splits - 1) / x.splits) * (x.splits - 1) >= K / BKB)
        return fail(LDIT_EINVAL, "gemm_bf16: %d K-splits leave an empty slab for K=%d", x.splits, K);
    if (x.Ypre && (reinterpret_cast<uintptr_t>(x.Ypre) & 7u)) return fail(LDIT_EINVAL, "gemm_bf16: Ypre misaligned");
    GemmArgsH a{};
    a.A = static_cast<const bf16_t *>(A); a.W = static_cast<const bf16_t *>(W); a.Y = Y; a.Y2 = Y2; a.bias = bias; a.lam = lam;
    a.R = R; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldy = ldy; a.x = x;
    a.ldw = K;
    if (x.nseg != 0 && diag().seg_order >= 0) a.x.seg_inner = diag().seg_order;
    if (x.nseg != 0) {
        // split-fp32 operands: the planes of a row lie side by side; the widest plane index fixes the W row stride
        if (x.nseg < 1 || x.nseg > 8 || x.splits != 1) return fail(LDIT_EINVAL, "gemm_bf16: bad plane-segment list");
        unsigned pa = 0, pw = 0;
        for (int g = 0; g < x.nseg; ++g) {
            pa = std::max(pa, (x.seg_a >> (4 * g)) & 15u);
            pw = std::max(pw, (x.seg_w >> (4 * g)) & 15u);
        }
        if ((long)lda < (long)(pa + 1) * K) return fail(LDIT_EINVAL, "gemm_bf16: lda too small for the A planes");
        a.ldw = (int)(pw + 1) * K;
    }
    a.direct_epi = diag().direct_epi ? 1 : 0;
    switch (epi) {
        case EPI_BIAS: return launch_h_tiled<EPI_BIAS>(a, stream);
        case EPI_BIAS_GELU: return launch_h_tiled<EPI_BIAS_GELU>(a, stream);
        case EPI_SCALE_RESID:
            if (!lam || !R) return fail(LDIT_EINVAL, "gemm_bf16: scale+residual epilogue needs lam and R");
            return launch_h_tiled<EPI_SCALE_RESID>(a, stream);
        case EPI_F32: return launch_h_tiled<EPI_F32>(a, stream);
        case EPI_GELU_SPLIT:
            if (x.nsplit_out < 2 || x.nsplit_out > 3 || ldy != x.nsplit_out * N || (N & 3))
                return fail(LDIT_EINVAL, "gemm_bf16: split GELU epilogue needs 2 or 3 output planes and ldy = planes * N");
            return launch_h_tiled<EPI_GELU_SPLIT>(a, stream);
        case EPI_BIAS_SPLIT:
            if (x.nsplit_out < 2 || x.nsplit_out > 3 || ldy != x.nsplit_out * N || (N & 3))
                return fail(LDIT_EINVAL, "gemm_bf16: split bias epilogue needs 2 or 3 output planes and ldy = planes * N");
            return launch_h_tiled<EPI_BIAS_SPLIT>(a, stream);
        case EPI_EMBED:
            if (!x.pos || x.patches <= 0 || M % x.patches || !aligned16(x.pos) || (ldy & 3))
                return fail(LDIT_EINVAL, "gemm_bf16: patch-embedding epilogue needs pos, patches | M and 16-byte rows");
            return launch_h_tiled<EPI_EMBED>(a, stream);
        case EPI_GELU_BWD:
            if (!x.aux || (x.ldaux & 3) || (reinterpret_cast<uintptr_t>(x.aux) & 7u)) return fail(LDIT_EINVAL, "gemm_bf16: GELU-backward epilogue needs an aligned pre-activation operand");
            return launch_h_tiled<EPI_GELU_BWD>(a, stream);
        default: return fail(LDIT_EINVAL, "gemm_bf16: unknown epilogue %d", epi);
    }
}

int launch_cvt_bf16(const float *src, void *dst, size_t n, hipStream_t stream, float mul)
{
    if (n == 0) return LDIT_OK;
    hipLaunchKernelGGL(cvt_f32_bf16, dim3((unsigned)((n / 4 + 255) / 256 + 1)), dim3(256), 0, stream, src,
                       static_cast<bf16_t *>(dst), n, mul);
    LDIT_HIP_CHECK(hipGetLastError());
    return LDIT_OK;
}

}  // namespace ldit

#ifdef LDIT_GEMM_STAMPS
extern "C" int ldit_dbg_set_gemm_bf16_stamps(void *buf)
{
    unsigned long long *p = static_cast<unsigned long long *>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_bf16_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -3;
}
#endif
