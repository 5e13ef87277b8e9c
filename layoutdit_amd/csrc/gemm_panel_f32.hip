// fp32 MFMA GEMM, "panel" tiling:  Y[M,N] = epi( A[M,K] . W[N,K]^T ), block tile (32*T32 + 16*HALF) x 128,
// four waves side by side along N, each owning the full block height x 32 columns.
//
// Why a second kernel next to gemm_f32.hip: there every SIMD owns whole 32x32 output tiles and the block height is a
// multiple of 64.  M = 64 images x 197 tokens = 12608 rows gives 9.23 such tiles per SIMD for N = 768 - rounded up
// to 10, i.e. 92.3 % of the machine at best, for every GEMM of ViT-B at batch 64.  Here the wave tile is
// 304 x 32 = nine 32x32 tiles (v_mfma_f32_32x32x2_f32) + one 16-row remainder (two v_mfma_f32_16x16x4_f32 tiles):
// 42 x N/128 blocks = 252 / 756 / 1008 for N = 768 / 2304 / 3072, i.e. 0.98 / 2.95 / 3.94 rounds of 256 CUs -> 97.2 %.
// (An all-16x16x4 version of the same tile was measured 10 % slower in the main loop: hipcc rotates the 38 four-register
// accumulator tuples through VGPRs at the loop edge, 250 v_accvgpr moves per k-tile.  Nine 16-register tuples + two
// small ones are allocated in place.)
//
//   * operands are swapped in the MFMA (A-operand <- W rows, B-operand <- activation rows), so an accumulator register
//     quad holds 4 CONSECUTIVE output columns of one row: the whole epilogue (bias, erf-GELU, lambda, residual, tap copy,
//     position add) runs on float4 and stores 16 B per lane - 4x fewer memory instructions than gemm_f32.hip.
//   * K-contiguous operands -> LDS by LDS-DMA, 128-B rows, 16-B chunk XOR-swizzled with (row>>1)&7 on the source address
//     and on the read: the b128 reads of the nine 32-row tiles are conflict-free; the scalar reads of the 16-row remainder
//     are 4-way conflicted (41 % of the LDS-active cycles) but off the critical path - see load_frags for the measured
//     conflict-free alternative and why it is not the default.
//   * 8-deep chunks: a 32-row tile lane (r, h) reads k = 8c+4h..+3 (b128), MFMA step s uses k-pair {s, 4+s};
//     a 16-row tile lane (r, q) reads k = 8c+4(q&1)+(q>>1) and +2, so its two 16x16x4 steps add k0,k4,k1,k5 | k2,k6,k3,k7:
//     the same product order as the 32x32x2 steps - results are bit-identical across the two shapes.
//   * two LDS stages, branch-free k-tile body of four chunks pinned with sched_group_barrier: fragments of chunk c+1 are
//     read under the MFMAs of chunk c, the DMA pieces of tile kt+1 are dribbled between the MFMAs of chunk 0, the
//     hand-over barrier sits in front of the last chunk's MFMAs (same pipeline as gemm_f32.hip).
//   * optional PERSISTENT tile loop (round 4, LDIT_GEMM_PERSIST=1; measured, NOT the default): a launch of more tiles than CUs
//     (q|k|v: 2.95 rounds, fc1: 3.94) runs 256 workgroups that walk their tiles; after a tile's k-loop the first k-tile of the
//     NEXT tile is DMA'd into the LDS stage the last hand-over freed, BEFORE the epilogue's stores are issued, so the next
//     prologue and the drain of the stores overlap.  Bit-identical (tested) - and no faster: q|k|v 362.6 vs 353.1 us, fc1 470.5 vs
//     472.9 us, layer 0.995x (profiles/r04_f32_persistent_ab.txt).  The hardware dispatcher already starts the next workgroup of
//     a CU while the previous one's stores drain, and its dynamic order absorbs per-CU speed differences that a static walk
//     turns into stragglers.  Kept selectable as evidence.
//   * exact fp32: every output element is one fma chain over k in one fixed order, the same for both MFMA shapes, so a
//     row's result does not depend on where in a block or batch it sits (tests/test_gpu_forward.py checks bit equality).
#include <cstdlib>

#include "ldit_common.h"

namespace ldit {

namespace {

constexpr int BK = 32;
constexpr int ROW_BYTES = BK * 4;
constexpr int BNP = 128;

__device__ __forceinline__ void glds16p(const float *gsrc, char *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}


// One float4 of the output: row m, columns n..n+3.  MODE 1: 16-B accesses, no bounds checks (block inside the matrix,
// ldy % 4 == 0); MODE 2: 16-B accesses behind a row check (the ragged last row panel: columns inside, rows past M
// skipped - the element-wise path costs ~20 us there because a lane owns a row, so its 4-byte accesses do not coalesce);
// MODE 0: element-wise with checks.  `res` = residual values already fetched (EPI_SCALE_RESID).
template <int EPI, int MODE, bool DUAL>
__device__ __forceinline__ void emit4(const GemmArgs &p, int m, int n, f32x4 acc, f32x4 bias, f32x4 lam, f32x4 res)
{
    constexpr bool VEC = MODE != 0;
    if (MODE != 1 && m >= p.M) return;
    unsigned o = (unsigned)m * (unsigned)p.ldy + (unsigned)n;
    f32x4 pos = {0.f, 0.f, 0.f, 0.f};
    if (EPI == EPI_EMBED) {
        const int b = m / p.patches, pi = m - b * p.patches;
        o = ((unsigned)b * p.tokens + 1 + pi) * (unsigned)p.ldy + (unsigned)n;
        const unsigned po = (unsigned)(1 + pi) * (unsigned)p.N + (unsigned)n;
        if (VEC) pos = *reinterpret_cast<const f32x4 *>(p.pos + po);
        else
#pragma unroll
            for (int r = 0; r < 4; ++r) pos[r] = (n + r < p.N) ? p.pos[po + r] : 0.0f;
    }
    f32x4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float t = acc[r] + bias[r];
        if (EPI == EPI_EMBED) t += pos[r];
        if (EPI == EPI_BIAS_GELU) t = gelu_erf(t);
        if (EPI == EPI_SCALE_RESID) t = res[r] + lam[r] * t;
        v[r] = t;
    }
    if (VEC) {
        *reinterpret_cast<f32x4 *>(p.Y + o) = v;
        if (DUAL) *reinterpret_cast<f32x4 *>(p.Y2 + o) = v;
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (n + r < p.N) {
                p.Y[o + r] = v[r];
                if (DUAL) p.Y2[o + r] = v[r];
            }
    }
}

template <int MODE>
__device__ __forceinline__ f32x4 load4(const GemmArgs &p, const float *base, int m, int n, bool row_major_out)
{
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 1 || (MODE == 2 && m < p.M)) return *reinterpret_cast<const f32x4 *>(base + ((unsigned)m * (unsigned)p.ldy + (unsigned)n));
    if (MODE == 2) return v;
    if (m < p.M)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (n + r < p.N) v[r] = base[(unsigned)m * (unsigned)p.ldy + (unsigned)(n + r)];
    (void)row_major_out;
    return v;
}

template <int MODE>
__device__ __forceinline__ f32x4 vec4(const float *v, int n, int N)
{
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (!v) return o;
    if (MODE != 0) return *reinterpret_cast<const f32x4 *>(v + n);
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (n + r < N) o[r] = v[n + r];
    return o;
}

// Epilogue of one wave.  32-row tile t, lane (c = lane&31, h = lane>>5): row m0+32t+c, register quad g holds columns
// nw + 8g + 4h .. +3.  16-row remainder, lane (c = lane&15, q = lane>>4): row m0+32*T32+c, tile it holds nw+16it+4q..+3.
// The residual quads of the next tile are fetched before the current tile is stored (R may alias Y: fenced so that
// hipcc neither hoists all loads to the top nor serialises them - see gemm_f32.hip).
template <int T32, bool HALF, int EPI, int VEC, bool DUAL>
__device__ __forceinline__ void store_panel(const GemmArgs &p, const f32x16 (&acc32)[T32], const f32x4 (&acc16)[2], int m0,
                                            int nw, int lane)
{
    const int c32 = lane & 31, h = lane >> 5, c16 = lane & 15, q = lane >> 4;
    f32x4 bias32[4], lam32[4], res[2][4];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        bias32[g] = vec4<VEC>(p.bias, nw + 8 * g + 4 * h, p.N);
        lam32[g] = EPI == EPI_SCALE_RESID ? vec4<VEC>(p.lam, nw + 8 * g + 4 * h, p.N) : zero;
        res[0][g] = zero;
        res[1][g] = zero;
    }
    auto fetch = [&](int t, f32x4(&dst)[4]) {
#pragma unroll
        for (int g = 0; g < 4; ++g) dst[g] = load4<VEC>(p, p.R, m0 + 32 * t + c32, nw + 8 * g + 4 * h, true);
    };
    if constexpr (EPI == EPI_SCALE_RESID && VEC == 1 && T32 > 0) {
        // Interior tile (round 4): EVERY residual quad of the wave tile is fetched before the first store, waited for once, and the
        // stores then go out back to back.  The tile-ahead pipeline below left an s_waitcnt vmcnt(0) in front of every 32-row tile's
        // stores (nine per wave tile) - and vmcnt counts stores: each of them sat out the acknowledgement of the previous tile's
        // stores (profiles/r04_epilogue_prefetch_ab.txt, the same finding in the bf16 slab epilogue).  R may alias Y (in-place
        // residual): every element is read before it is written, by the same lane.  144 + 8 registers beside the accumulators: a
        // workgroup is four waves, one per SIMD - the register file has them.
        f32x4 ra[T32][4], rh[2], bh[2], lh[2];
#pragma unroll
        for (int t = 0; t < T32; ++t) fetch(t, ra[t]);
        if (HALF) {
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                rh[it] = load4<VEC>(p, p.R, m0 + 32 * T32 + c16, nw + 16 * it + 4 * q, true);
                bh[it] = vec4<VEC>(p.bias, nw + 16 * it + 4 * q, p.N);      // (the remainder's bias / lambda quads too: a load behind the
                lh[it] = vec4<VEC>(p.lam, nw + 16 * it + 4 * q, p.N);       //  stores would wait for every one of them)
            }
        }
#pragma unroll
        for (int t = 0; t < T32; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(ra[t][g]));      // hipcc waits here, once, and knows of no pending load below
        if (HALF) asm volatile("" : "+v"(rh[0]), "+v"(rh[1]), "+v"(bh[0]), "+v"(bh[1]), "+v"(lh[0]), "+v"(lh[1]));
        asm volatile("" ::: "memory");
#pragma unroll
        for (int t = 0; t < T32; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 a = {acc32[t][4 * g + 0], acc32[t][4 * g + 1], acc32[t][4 * g + 2], acc32[t][4 * g + 3]};
                emit4<EPI, VEC, DUAL>(p, m0 + 32 * t + c32, nw + 8 * g + 4 * h, a, bias32[g], lam32[g], ra[t][g]);
            }
        if (HALF) {
            const int m = m0 + 32 * T32 + c16;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int n = nw + 16 * it + 4 * q;
                emit4<EPI, VEC, DUAL>(p, m, n, acc16[it], bh[it], lh[it], rh[it]);
            }
        }
        return;
    }
    if (EPI == EPI_SCALE_RESID && T32 > 0) fetch(0, res[0]);
#pragma unroll
    for (int t = 0; t < T32; ++t) {
        if (EPI == EPI_SCALE_RESID) {
            if (t + 1 < T32) fetch(t + 1, res[(t + 1) & 1]);
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 a = {acc32[t][4 * g + 0], acc32[t][4 * g + 1], acc32[t][4 * g + 2], acc32[t][4 * g + 3]};
            emit4<EPI, VEC, DUAL>(p, m0 + 32 * t + c32, nw + 8 * g + 4 * h, a, bias32[g], lam32[g], res[t & 1][g]);
        }
        if (EPI == EPI_SCALE_RESID) asm volatile("" ::: "memory");
    }
    if (HALF) {
        const int m = m0 + 32 * T32 + c16;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int n = nw + 16 * it + 4 * q;
            const f32x4 r = EPI == EPI_SCALE_RESID ? load4<VEC>(p, p.R, m, n, true) : zero;
            const f32x4 l = EPI == EPI_SCALE_RESID ? vec4<VEC>(p.lam, n, p.N) : zero;
            emit4<EPI, VEC, DUAL>(p, m, n, acc16[it], vec4<VEC>(p.bias, n, p.N), l, r);
        }
    }
}

template <int T32, bool HALF, int EPI, int AMODE, bool R16VEC = true>
__global__ void __launch_bounds__(256) gemm_panel_f32(const GemmArgs p)
{
    constexpr int BM = 32 * T32 + (HALF ? 16 : 0), BN = BNP, PIECES = (BM + BN) / 8, NLD = (PIECES + 3) / 4;
    constexpr int STAGE_BYTES = NLD * 4 * 1024;     // pieces past PIECES land in slack rows nobody reads
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c32 = lane & 31, h = lane >> 5, c16 = lane & 15, q = lane >> 4;
#ifdef LDIT_GEMM_STAMPS
    unsigned long long st_real0 = __builtin_amdgcn_s_memrealtime(), st_clk0 = __builtin_amdgcn_s_memtime();
    unsigned long long st_clk1 = 0, st_clk2 = 0;
#endif

    // ---- slot -> tile (XCD-aware, bijective; tiles of one A row-panel are consecutive).  Slots = tiles; a workgroup walks the
    // slots blockIdx.x, + gridDim.x, ... (gridDim.x is a multiple of 8 whenever it is smaller than the tile count, so a workgroup
    // stays inside one XCD's chunk of tiles)
    const int nbn = (p.N + BN - 1) / BN, nbm = (p.M + BM - 1) / BM;
    const int ntiles = nbm * nbn;
    auto tile_of = [&](int slot) {
        const int xcd = slot & 7, idx = slot >> 3, qq = ntiles >> 3, rr = ntiles & 7;
        return (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    };

    // ---- DMA sources: this wave moves pieces wave + 4u (8 rows x 128 B each).  Row-major operands: the tile origin
    // (m0 * lda, n0 * K) goes into the 64-bit wave-uniform base, only the offset INSIDE the tile (< 304 rows) is kept in
    // 32 bits - as a byte offset it stays below 2^32 for any row stride the host admits (ldit_linear_f32: lda, K < 2^21),
    // so M * lda may be anything the 2^31-element host limit allows.  Patch mode keeps 32-bit ELEMENT offsets into the
    // image batch (host limit: B * in_ch * H * W < 2^31).
    unsigned src[NLD];
    const float *a_tile, *w_tile;
    int m0, n0;
    auto setup = [&](int tile) {
        m0 = (tile / nbn) * BM;
        n0 = (tile % nbn) * BN;
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int row = 8 * (wave + 4 * u) + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            if (8 * (wave + 4 * u) < BM) {
                int gm = m0 + row;
                gm = gm < p.M ? gm : p.M - 1;
                if (AMODE == A_PATCH) {
                    const int b = gm / p.patches, pi = gm - b * p.patches;
                    const int gy = pi / p.gw, gx = pi - gy * p.gw;
                    src[u] = (unsigned)b * (unsigned)p.lda + (unsigned)(gy * p.patch * p.img_w + gx * p.patch);
                } else {
                    src[u] = ((unsigned)(gm - m0) * (unsigned)p.lda + c * 4) * 4u;      // BYTE offset inside the tile
                }
            } else {
                int gn = n0 + row - BM;
                gn = gn < p.N ? gn : p.N - 1;
                src[u] = ((unsigned)(gn - n0) * (unsigned)p.K + c * 4) * 4u;            // BYTE offset inside the tile
            }
        }
        a_tile = AMODE == A_PATCH ? p.A : p.A + (size_t)m0 * (size_t)p.lda;
        w_tile = p.W + (size_t)n0 * (size_t)p.K;
    };
    auto issue = [&](int stage, int k0) {
        char *base = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int piece = wave + 4 * u;
            const float *opnd = 8 * piece < BM ? a_tile : w_tile;
            const float *g;
            if (AMODE == A_PATCH && 8 * piece < BM) {
                const int row = 8 * piece + (lane >> 3);
                const int c = (lane & 7) ^ ((row >> 1) & 7);
                const int k = k0 + 4 * c, pp = p.patch * p.patch;
                const int ch = k / pp, rem = k - ch * pp, dy = rem / p.patch, dx = rem - dy * p.patch;
                g = opnd + (src[u] + (unsigned)((ch * p.img_h + dy) * p.img_w + dx));
            } else {
                // uniform base advanced by k0 (SALU) + fixed per-lane byte offset: the DMA is issued in its
                // saddr + 32-bit voffset form with no per-piece VALU address arithmetic
                g = reinterpret_cast<const float *>(reinterpret_cast<const char *>(opnd + k0) + src[u]);
            }
            glds16p(g, base + piece * 1024);
        }
    };

    f32x16 acc32[T32 > 0 ? T32 : 1];
    f32x4 acc16[2];

    const int nk = p.K / BK;
    const int sw32 = (c32 >> 1) & 7, sw16 = (c16 >> 1) & 7;
    const int x32_row = c32 * ROW_BYTES, w32_row = (BM + wave * 32 + c32) * ROW_BYTES;
    const int x16_row = (32 * T32 + c16) * ROW_BYTES, w16_row = (BM + wave * 32 + c16) * ROW_BYTES;
    const bool odd16 = (q >> 1) != 0;     // which two of the four k of its 16-B half-chunk this lane multiplies

    struct Frags {
        f32x4 x32[T32 > 0 ? T32 : 1];
        f32x4 w32;
        f32x2 x16;
        f32x2 w16[2];
    };
    auto load_frags = [&](int stage, int c, Frags &f) {
        const char *st = smem + stage * STAGE_BYTES;
        const char *b32 = st + (((c * 2 + h) ^ sw32) * 16);
#pragma unroll
        for (int t = 0; t < T32; ++t) f.x32[t] = *reinterpret_cast<const f32x4 *>(b32 + x32_row + t * 32 * ROW_BYTES);
        f.w32 = *reinterpret_cast<const f32x4 *>(b32 + w32_row);
        if (HALF) {
            // lane (r, q), element s <- k = 8c + 4(q&1) + (q>>1) + 2s: the 16x16x4 steps then add the products in the
            // order k0,k4,k1,k5 | k2,k6,k3,k7 - the order of the 32x32x2 steps above - so a row's result is bit-identical
            // whichever tile shape it falls into (rows keep their values when the batch composition changes)
            // Two ways to fetch the lane's two floats of its 16-B half-chunk:
            //  * scalar ds_read_b32 (default): the 32 lanes of a half-wave land on 8 banks (every lane reads the same offset
            //    inside its chunk) - a 4-way conflict on three reads per chunk, 41 % of SQ_LDS_IDX_ACTIVE in the PMC pass
            //    (profiles/r01_final_pmc_summary.txt).  The LDS is idle ~85 % of a chunk's 2.4 k MFMA cycles, the reads are
            //    issued a whole chunk ahead, so the conflict costs nothing measurable;
            //  * R16VEC: whole half-chunk by ds_read_b128 (conflict-free under the (row>>1)&7 swizzle, like the 32-row reads:
            //    SQ_LDS_BANK_CONFLICT = 0, profiles/r02_cfg1_vec_pmc_summary.txt) + a register select.  1.7 % SLOWER per layer
            //    (12 more VGPRs live per chunk, 6 selects, 3x the LDS bytes) - kept selectable as evidence, not shipped.
            const char *b16 = st + (((c * 2 + (q & 1)) ^ sw16) * 16);
            if (R16VEC) {
                const f32x4 xv = *reinterpret_cast<const f32x4 *>(b16 + x16_row);
                const f32x4 w0 = *reinterpret_cast<const f32x4 *>(b16 + w16_row);
                const f32x4 w1 = *reinterpret_cast<const f32x4 *>(b16 + w16_row + 16 * ROW_BYTES);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    f.x16[s] = odd16 ? xv[2 * s + 1] : xv[2 * s];
                    f.w16[0][s] = odd16 ? w0[2 * s + 1] : w0[2 * s];
                    f.w16[1][s] = odd16 ? w1[2 * s + 1] : w1[2 * s];
                }
            } else {        // LDIT_PANEL_R16=scalar: the round-1 scalar reads (4-way bank conflict), kept for A/B timing only
                const int o16 = odd16 ? 4 : 0;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    f.x16[s] = *reinterpret_cast<const float *>(b16 + x16_row + o16 + 8 * s);
                    f.w16[0][s] = *reinterpret_cast<const float *>(b16 + w16_row + o16 + 8 * s);
                    f.w16[1][s] = *reinterpret_cast<const float *>(b16 + w16_row + 16 * ROW_BYTES + o16 + 8 * s);
                }
            }
        }
    };
    auto mfma_chunk = [&](const Frags &f) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int t = 0; t < T32; ++t)
                acc32[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.w32[s], f.x32[t][s], acc32[t], 0, 0, 0);
            if (HALF && s < 2) {
                acc16[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w16[0][s], f.x16[s], acc16[0], 0, 0, 0);
                acc16[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w16[1][s], f.x16[s], acc16[1], 0, 0, 0);
            }
        }
    };

    constexpr int SG_MFMA = 0x8, SG_VMEM = 0x20, SG_DSR = 0x100;
    constexpr int NM = 4 * T32 + (HALF ? 4 : 0);          // MFMAs per 8-deep chunk
    constexpr int NF = T32 + 1 + (HALF ? 3 : 0);          // LDS fragment reads per chunk (three b128 for the 16-row remainder)
    constexpr int MPD = (NM - NF) / NLD;                  // MFMAs per DMA piece in chunk 0 (0: tile too small)
    static_assert(NF <= NM, "fewer MFMAs than fragment reads in a chunk");

    Frags f0, f1;
    int slot = blockIdx.x;
    setup(tile_of(slot));
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // explicit: the first tile has landed before anybody reads it
    __syncthreads();
#ifdef LDIT_GEMM_STAMPS
    st_clk1 = __builtin_amdgcn_s_memtime();
#endif
    int flip = 0;                // physical stage of this tile's k-tile 0 (wave-uniform)
    for (;;) {
#pragma unroll
    for (int t = 0; t < T32; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc32[t][e] = 0.0f;
    acc16[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc16[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    load_frags(flip, 0, f0);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = (kt & 1) ^ flip;
        const int knext = (kt + 1 < nk ? kt + 1 : nk - 1) * BK;
        __builtin_amdgcn_sched_barrier(0);
        // ---- chunk 0: MFMAs of f0 | read chunk 1 -> f1 | DMA tile kt+1 -> stage cur^1 (free since the last hand-over)
        load_frags(cur, 1, f1);
        issue(cur ^ 1, knext);
        mfma_chunk(f0);
#pragma unroll
        for (int g = 0; g < NF; ++g) {
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
            __builtin_amdgcn_sched_group_barrier(SG_DSR, 1, 0);
        }
        if (MPD > 0) {
#pragma unroll
            for (int g = 0; g < NLD; ++g) {
                __builtin_amdgcn_sched_group_barrier(SG_MFMA, MPD, 0);
                __builtin_amdgcn_sched_group_barrier(SG_VMEM, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, NM - NF - MPD * NLD, 0);
        } else {
            __builtin_amdgcn_sched_group_barrier(SG_VMEM, NLD, 0);
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, NM - NF, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- chunk 1: MFMAs of f1 | read chunk 2 -> f0
        load_frags(cur, 2, f0);
        mfma_chunk(f1);
#pragma unroll
        for (int g = 0; g < NF; ++g) {
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 1);
            __builtin_amdgcn_sched_group_barrier(SG_DSR, 1, 1);
        }
        __builtin_amdgcn_sched_group_barrier(SG_MFMA, NM - NF, 1);
        __builtin_amdgcn_sched_barrier(0);
        // ---- chunk 2: MFMAs of f0 | read chunk 3 -> f1
        load_frags(cur, 3, f1);
        mfma_chunk(f0);
#pragma unroll
        for (int g = 0; g < NF; ++g) {
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 2);
            __builtin_amdgcn_sched_group_barrier(SG_DSR, 1, 2);
        }
        __builtin_amdgcn_sched_group_barrier(SG_MFMA, NM - NF, 2);
        __builtin_amdgcn_sched_barrier(0);
        // ---- hand-over: own DMA landed (vmcnt 0), own reads of stage cur done (lgkmcnt 0), then all waves
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // explicit: do not rely on hipcc to drain the LDS-DMA in front of the barrier
        __syncthreads();
        // ---- chunk 3: MFMAs of f1 | read chunk 0 of the next stage -> f0
        load_frags(cur ^ 1, 0, f0);
        mfma_chunk(f1);
#pragma unroll
        for (int g = 0; g < NF; ++g) {
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 3);
            __builtin_amdgcn_sched_group_barrier(SG_DSR, 1, 3);
        }
        __builtin_amdgcn_sched_group_barrier(SG_MFMA, NM - NF, 3);
    }

#ifdef LDIT_GEMM_STAMPS
    st_clk2 = __builtin_amdgcn_s_memtime();
#endif
    // ---- next tile's first k-tile -> the stage the last hand-over freed (every wave has passed that barrier, so nobody reads it
    // any more; the other stage only feeds the discarded read-ahead of the last chunk), BEFORE this tile's stores are issued
    const int em0 = m0, en0 = n0;
    const int free_stage = ((nk - 1) & 1) ^ flip;
    const int next_slot = slot + (int)gridDim.x;
    const bool more = next_slot < ntiles;                                           // block-uniform
    if (more) {
        setup(tile_of(next_slot));
        issue(free_stage, 0);
    }
    // ---- epilogue ---------------------------------------------------------------------------------------------------
    const bool cols_in = (en0 + BN <= p.N) && ((p.ldy & 3) == 0);                   // block-uniform
    const int nw = en0 + wave * 32;
    const bool interior = cols_in && em0 + BM <= p.M;
    if (interior) {
        if (p.Y2) store_panel<T32, HALF, EPI, 1, true>(p, acc32, acc16, em0, nw, lane);
        else store_panel<T32, HALF, EPI, 1, false>(p, acc32, acc16, em0, nw, lane);
    } else if (cols_in) {
        if (p.Y2) store_panel<T32, HALF, EPI, 2, true>(p, acc32, acc16, em0, nw, lane);
        else store_panel<T32, HALF, EPI, 2, false>(p, acc32, acc16, em0, nw, lane);
    } else {
        if (p.Y2) store_panel<T32, HALF, EPI, 0, true>(p, acc32, acc16, em0, nw, lane);
        else store_panel<T32, HALF, EPI, 0, false>(p, acc32, acc16, em0, nw, lane);
    }
    if (!more) break;
    // The next k-tile 0 must have landed before anybody reads it.  Vector-memory operations retire in issue order, and this wave
    // issued at least 4 T32 store instructions (one 16-B store per accumulator quad of the 32-row tiles; global stores cannot be
    // moved above the DMA, which reads global memory) AFTER its DMA pieces: "all but the youngest 4 T32 done" therefore covers every
    // piece while the tail of the stores stays in flight under the next k-loop.  Any other epilogue shape drains everything.
    if (interior && !p.Y2 && T32 > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * T32) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    flip = free_stage;
    slot = next_slot;
    }
#ifdef LDIT_GEMM_STAMPS
    if (p.stamps) {
        const unsigned long long st_clk3 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long st_clk4 = __builtin_amdgcn_s_memtime(), st_real1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            unsigned long long *o = p.stamps + (size_t)blockIdx.x * 8;
            o[0] = st_real0; o[1] = st_real1; o[2] = st_clk1 - st_clk0; o[3] = st_clk2 - st_clk1; o[4] = st_clk3 - st_clk2;
            o[5] = st_clk4 - st_clk3; o[6] = 0; o[7] = slot;
        }
    }
#endif
}

template <int T32, bool HALF, int EPI, int AMODE, bool R16VEC>
int launch_panel_v(const GemmArgs &a, hipStream_t stream);

template <int T32, bool HALF, int EPI, int AMODE>
int launch_panel(const GemmArgs &a, hipStream_t stream)
{
    // LDIT_PANEL_R16=vec selects the conflict-free b128 remainder reads (see load_frags); the default scalar reads measured
    // 1.7 % faster (profiles/README.md, round 2: three interleaved A/B pairs on one device)
    const bool vec = diag().panel_r16_vec;
    return vec ? launch_panel_v<T32, HALF, EPI, AMODE, true>(a, stream) : launch_panel_v<T32, HALF, EPI, AMODE, false>(a, stream);
}

template <int T32, bool HALF, int EPI, int AMODE, bool R16VEC>
int launch_panel_v(const GemmArgs &a, hipStream_t stream)
{
    constexpr int BM = 32 * T32 + (HALF ? 16 : 0), PIECES = (BM + BNP) / 8, NLD = (PIECES + 3) / 4;
    constexpr int lds = 2 * NLD * 4 * 1024;
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BNP - 1) / BNP);
    auto kern = gemm_panel_f32<T32, HALF, EPI, AMODE, R16VEC>;
    LDIT_DYN_LDS(kern, lds);
    // LDIT_GEMM_PERSIST=1: as many workgroups as fit the chip at once (one per CU at > 80 KB of LDS each, two below), each walking its tiles
    const int resident = (lds > 80 * 1024 ? 1 : 2) * compute_units();
    const int grid = (tiles > resident && diag().panel_persist) ? resident : tiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
    LDIT_HIP_CHECK(hipGetLastError());
    return LDIT_OK;
}

}  // namespace

// Panel tiling with a (32 T + 16) x 128 block: 304 rows for the headline shapes, 144 / 80 / 48 rows where the taller panels
// would leave most of the chip idle (M = 1 576 .. 3 152 rows at N = 768: 36-66 panels of 304 against 240-198 of 80 / 48).
// Every height keeps the 32-row tiles + 16-row remainder structure, so a row's product order - and its bits - do not depend on
// the height.  Arguments already validated by launch_gemm.
template <int T32>
static int launch_panel_epi(const GemmArgs &a, int epi, hipStream_t stream)
{
    switch (epi) {
        case EPI_BIAS: return launch_panel<T32, true, EPI_BIAS, A_ROWMAJOR>(a, stream);
        case EPI_BIAS_GELU: return launch_panel<T32, true, EPI_BIAS_GELU, A_ROWMAJOR>(a, stream);
        default: return launch_panel<T32, true, EPI_SCALE_RESID, A_ROWMAJOR>(a, stream);
    }
}

int launch_gemm_panel(const GemmArgs &a, int epi, int amode, hipStream_t stream, int height)
{
    if (amode == A_PATCH) return launch_panel<9, true, EPI_EMBED, A_PATCH>(a, stream);
    switch (height) {
        case 48: return launch_panel_epi<1>(a, epi, stream);
        case 80: return launch_panel_epi<2>(a, epi, stream);
        case 144: return launch_panel_epi<4>(a, epi, stream);
        default: return launch_panel_epi<9>(a, epi, stream);
    }
}

}  // namespace ldit
