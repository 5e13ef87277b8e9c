// fp32 MFMA GEMM with fused epilogues:  Y[M,N] = epi( A[M,K] . W[N,K]^T ).
//
// The five GEMMs of a BEiT layer stack (patch-embed, fused qkv, o_proj, fc1, fc2; SURVEY.md 2.2 K1,K4,K7-K9) are
// this one kernel; it carries ~96 % of the forward's FLOPs.  gfx950 design:
//   * v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD = the fp32 matrix peak; no xf32 on gfx950).  One wave per
//     SIMD; each wave owns TM x TN accumulator tiles of 32x32 (TM*TN*16 registers) so an 8-deep k-chunk costs
//     TM+TN ds_read_b128 for 4*TM*TN MFMAs - LDS and issue bandwidth are nowhere near the 64-cycle MFMA cadence.
//   * both operands are K-contiguous, staged HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, 8 rows x 128 B per
//     wave-instruction), double buffered: tile kt+1 streams while tile kt is multiplied; one barrier per k-tile.
//   * the LDS image is lane-linear (DMA requirement) with 128-B rows; the 16-B chunk index is XORed with (row>>1)&7
//     on the global SOURCE address and again on the fragment read, which makes every ds_read_b128 conflict-free.
//   * within an 8-deep chunk lane half h reads k = 4h..4h+3 as one b128, so MFMA step s multiplies k-pair
//     {s, 4+s}: a permutation of the k order applied identically to both operands.
//   * C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5): a register's 32 lanes store 128
//     contiguous bytes of one output row.
//   * blockIdx -> tile map is XCD-aware (tiles of one A row-panel land on one XCD so the panel is fetched into
//     that XCD's L2 once) and bijective for any tile count.
#include <cstdlib>

#include "ldit_common.h"

namespace ldit {

namespace {

constexpr int BK = 32;            // floats per k-tile: one 128-B LDS row
constexpr int ROW_BYTES = BK * 4;

__device__ __forceinline__ void glds16(const float *gsrc, char *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}


// Epilogue of one wave: register r of lanes 0-31 / 32-63 = 128 contiguous bytes of output rows 8(r>>2)+(r&3) / +4 of
// each 32x32 tile.  CHECK = false for blocks wholly inside the matrix: no per-element bounds work at all.
// Addressing is a uniform base pointer + a 32-bit per-lane index (one v_add per element).  The residual loads of tile
// t+1 are issued before the stores of tile t and fenced with compiler barriers: left alone, hipcc hoists all
// TM*TN*16 residual loads (and their 64-bit addresses) to the top and spills them to scratch (72k cycles per block).
template <int TM, int TN, int EPI, bool CHECK, bool DUAL>
__device__ __forceinline__ void store_tile(const GemmArgs &p, const f32x16 (&acc)[TM][TN], int mw, int nw)
{
    constexpr int NT = TM * TN;
    const unsigned ldy = (unsigned)p.ldy;
    const unsigned lane_base = (unsigned)mw * ldy + (unsigned)nw;
    float resid[2][16];
    auto fetch = [&](int t, float(&dst)[16]) {
        const int i = t % TM, j = t / TM;
        const int n = nw + j * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mw + i * 32 + 8 * (r >> 2) + (r & 3);
            const unsigned idx = lane_base + (unsigned)(i * 32 + 8 * (r >> 2) + (r & 3)) * ldy + (unsigned)(j * 32);
            dst[r] = (!CHECK || (n < p.N && m < p.M)) ? p.R[idx] : 0.0f;
        }
    };
    if (EPI == EPI_SCALE_RESID) fetch(0, resid[0]);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (EPI == EPI_SCALE_RESID) {
            if (t + 1 < NT) fetch(t + 1, resid[(t + 1) & 1]);
            asm volatile("" ::: "memory");
        }
        const int i = t % TM, j = t / TM;
        const int n = nw + j * 32;
        const bool nok = !CHECK || n < p.N;
        const float bias = (nok && p.bias) ? p.bias[n] : 0.0f;
        float lam = 0.0f;
        if (EPI == EPI_SCALE_RESID) lam = nok ? p.lam[n] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mw + i * 32 + 8 * (r >> 2) + (r & 3);
            if (CHECK && (!nok || m >= p.M)) continue;
            float v = acc[i][j][r] + bias;
            unsigned o;
            if (EPI == EPI_EMBED) {
                const int b = m / p.patches, pi = m - b * p.patches;
                v += p.pos[(unsigned)(1 + pi) * (unsigned)p.N + (unsigned)n];
                o = ((unsigned)b * p.tokens + 1 + pi) * ldy + (unsigned)n;
            } else {
                o = lane_base + (unsigned)(i * 32 + 8 * (r >> 2) + (r & 3)) * ldy + (unsigned)(j * 32);
                if (EPI == EPI_BIAS_GELU) v = gelu_erf(v);
                if (EPI == EPI_SCALE_RESID) v = resid[t & 1][r] + lam * v;
            }
            p.Y[o] = v;
            if (DUAL) p.Y2[o] = v;
        }
        if (EPI == EPI_SCALE_RESID) asm volatile("" ::: "memory");
    }
}

template <int TM, int TN, int EPI, int AMODE>
__global__ void __launch_bounds__(256) gemm_f32_mfma(const GemmArgs p)
{
    constexpr int BM = 64 * TM, BN = 64 * TN, ROWS = BM + BN, NLD = ROWS / 32;
    static_assert(ROWS % 32 == 0 && BM % 32 == 0, "tile rows must split into 8-row DMA pieces over 4 waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
#ifdef LDIT_GEMM_STAMPS
    // diagnostic build only (never shipped in libldit_hip.so): wall/cycle stamps of the block's phases
    unsigned long long st_real0 = __builtin_amdgcn_s_memrealtime(), st_clk0 = __builtin_amdgcn_s_memtime();
    unsigned long long st_clk1 = 0, st_clk2 = 0;
#endif

    // ---- block -> tile (XCD-aware, bijective) ------------------------------------------------------------------
    const int nbn = (p.N + BN - 1) / BN, nbm = (p.M + BM - 1) / BM;
    const int ntiles = nbm * nbn;
    int tile;
    {
        const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3, q = ntiles >> 3, r = ntiles & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (tile / nbn) * BM, n0 = (tile % nbn) * BN;

    // ---- DMA source addresses: this wave moves pieces q = wave + 4u, piece = 8 rows x 128 B -------------------------
    // 32-bit element offsets from the (uniform) operand base: every operand is < 2^31 elements (checked on the host)
    unsigned src[NLD];
    int yx[NLD];                                        // A_CONV3 only: (y << 16) | x of the lane's pixel, per A piece
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
        const int row = 8 * (wave + 4 * u) + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);      // logical 16-B chunk this lane fetches
        if (8 * (wave + 4 * u) < BM) {
            int gm = m0 + row;
            gm = gm < p.M ? gm : p.M - 1;
            if (AMODE == A_CONV3) {
                const int hw = p.conv_h * p.conv_w, b = gm / hw, r2 = gm - b * hw, y = r2 / p.conv_w, x = r2 - y * p.conv_w;
                src[u] = (unsigned)gm * (unsigned)p.conv_c + c * 4;      // centre tap, channel chunk c
                yx[u] = (y << 16) | x;
            } else if (AMODE == A_PATCH) {
                const int b = gm / p.patches, pi = gm - b * p.patches;
                const int gy = pi / p.gw, gx = pi - gy * p.gw;
                // per-image channel stride folded in at issue time (depends on k)
                src[u] = (unsigned)b * (unsigned)p.lda + (unsigned)(gy * p.patch * p.img_w + gx * p.patch);
            } else {
                src[u] = (unsigned)gm * (unsigned)p.lda + c * 4;
            }
        } else {
            int gn = n0 + row - BM;
            gn = gn < p.N ? gn : p.N - 1;
            src[u] = (unsigned)gn * (unsigned)p.K + c * 4;
        }
    }

    auto issue = [&](int stage, int k0) {
        char *base = smem + stage * (ROWS * ROW_BYTES);
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int piece = wave + 4 * u;
            const float *opnd = 8 * piece < BM ? p.A : p.W;   // wave-uniform
            const float *g;
            if (AMODE == A_CONV3 && 8 * piece < BM) {
                // k0 lies inside ONE tap (conv_c % 32 == 0): tap (ky, kx) uniform, channel offset c0 uniform
                const int t = k0 / p.conv_c, c0 = k0 - t * p.conv_c, ky = t / 3, kx = t - 3 * ky;
                const int yy = (yx[u] >> 16) + ky - 1, xx = (yx[u] & 0xffff) + kx - 1;
                const bool in = yy >= 0 && yy < p.conv_h && xx >= 0 && xx < p.conv_w;
                const int shift = ((ky - 1) * p.conv_w + (kx - 1)) * p.conv_c + c0;
                g = in ? opnd + (int)(src[u] + (unsigned)shift) : p.zeros + 4 * (lane & 7);
            } else if (AMODE == A_PATCH && 8 * piece < BM) {
                // k = (ch, dy, dx) with dx fastest; this lane's chunk starts at k0 + 4*c, c recovered from src
                const int row = 8 * piece + (lane >> 3);
                const int c = (lane & 7) ^ ((row >> 1) & 7);
                const int k = k0 + 4 * c, pp = p.patch * p.patch;
                const int ch = k / pp, rem = k - ch * pp, dy = rem / p.patch, dx = rem - dy * p.patch;
                g = opnd + (src[u] + (unsigned)((ch * p.img_h + dy) * p.img_w + dx));
            } else {
                g = opnd + (src[u] + (unsigned)k0);
            }
            glds16(g, base + piece * 1024);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int sw = (li >> 1) & 7;
    const int nk = p.K / BK;
    const int a_row = (wm * TM * 32 + li) * ROW_BYTES, b_row = (BM + wn * TN * 32 + li) * ROW_BYTES;

    // fragment of 8-deep chunk kc of a stage: lane half lh takes k = 8kc+4lh .. +3 as one (swizzled) b128 per 32-row tile
    auto load_frags = [&](int stage, int kc, f32x4(&a)[TM], f32x4(&b)[TN]) {
        const char *base = smem + stage * (ROWS * ROW_BYTES) + ((kc * 2 + lh) ^ sw) * 16;
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4 *>(base + a_row + i * 32 * ROW_BYTES);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4 *>(base + b_row + j * 32 * ROW_BYTES);
    };
    auto mfma_chunk = [&](const f32x4(&a)[TM], const f32x4(&b)[TN]) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
    };

    // Software pipeline (one basic block per k-tile, instruction order pinned with sched_group_barrier):
    //   * fragments of chunk c+1 are read from LDS while the 4*TM*TN MFMAs of chunk c execute;
    //   * the NLD LDS-DMA pieces of tile kt+1 are dribbled out one per two MFMAs of chunk 0 - issued back to back
    //     they cost ~60 cycles each with the matrix pipe idle (measured: 0.5 us of a 4.8 us k-tile);
    //   * the k-tile hand-over barrier sits in front of the LAST chunk's MFMAs, whose operands are already in
    //     registers, so neither an LDS latency nor the barrier drains the matrix pipe.
    // The body is branch-free: the last iteration re-fetches the last tile into the idle stage (drained by the final
    // hand-over's vmcnt(0)) and re-reads fragments it never uses.
    constexpr int SG_MFMA = 0x8, SG_VMEM = 0x20, SG_DSR = 0x100;
    constexpr int NM = 4 * TM * TN, NF = TM + TN;
    constexpr int MPD = (NM - NF) / NLD;       // MFMAs issued per DMA piece in chunk 0 (0: tile too small to interleave)
    static_assert(NF <= NM, "fewer MFMAs than fragment reads in a chunk");
    f32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // explicit: the first tile has landed before anybody reads it
    __syncthreads();
#ifdef LDIT_GEMM_STAMPS
    st_clk1 = __builtin_amdgcn_s_memtime();
#endif
    load_frags(0, 0, fa0, fb0);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const int knext = (kt + 1 < nk ? kt + 1 : nk - 1) * BK;
        __builtin_amdgcn_sched_barrier(0);
        // ---- chunk 0: MFMAs of f0 | read chunk 1 -> f1 | DMA tile kt+1 -> stage cur^1 (free since the last hand-over)
        load_frags(cur, 1, fa1, fb1);
        issue(cur ^ 1, knext);
        mfma_chunk(fa0, fb0);
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
        load_frags(cur, 2, fa0, fb0);
        mfma_chunk(fa1, fb1);
#pragma unroll
        for (int g = 0; g < NF; ++g) {
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 1);
            __builtin_amdgcn_sched_group_barrier(SG_DSR, 1, 1);
        }
        __builtin_amdgcn_sched_group_barrier(SG_MFMA, NM - NF, 1);
        __builtin_amdgcn_sched_barrier(0);
        // ---- chunk 2: MFMAs of f0 | read chunk 3 -> f1
        load_frags(cur, 3, fa1, fb1);
        mfma_chunk(fa0, fb0);
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
        load_frags(cur ^ 1, 0, fa0, fb0);
        mfma_chunk(fa1, fb1);
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
    // ---- epilogue ---------------------------------------------------------------------------------------------------
    const bool interior = (m0 + BM <= p.M) && (n0 + BN <= p.N);    // block-uniform
    const int mw = m0 + wm * TM * 32 + 4 * lh, nw = n0 + wn * TN * 32 + li;
    if (interior) {
        if (p.Y2) store_tile<TM, TN, EPI, false, true>(p, acc, mw, nw);
        else store_tile<TM, TN, EPI, false, false>(p, acc, mw, nw);
    } else {
        if (p.Y2) store_tile<TM, TN, EPI, true, true>(p, acc, mw, nw);
        else store_tile<TM, TN, EPI, true, false>(p, acc, mw, nw);
    }
#ifdef LDIT_GEMM_STAMPS
    if (p.stamps) {
        const unsigned long long st_clk3 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long st_clk4 = __builtin_amdgcn_s_memtime(), st_real1 = __builtin_amdgcn_s_memrealtime();
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if (tid == 0) {
            unsigned long long *o = p.stamps + (size_t)blockIdx.x * 8;
            o[0] = st_real0; o[1] = st_real1; o[2] = st_clk1 - st_clk0; o[3] = st_clk2 - st_clk1; o[4] = st_clk3 - st_clk2;
            o[5] = st_clk4 - st_clk3; o[6] = ((unsigned long long)xcc << 32) | hwid; o[7] = tile;
        }
    }
#endif
}

template <int TM, int TN, int EPI, int AMODE>
int launch_one(const GemmArgs &a, hipStream_t stream)
{
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int lds = 2 * (BM + BN) * ROW_BYTES;
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    auto kern = gemm_f32_mfma<TM, TN, EPI, AMODE>;
    LDIT_DYN_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), lds, stream, a);
    LDIT_HIP_CHECK(hipGetLastError());
    return LDIT_OK;
}

template <int EPI, int AMODE>
int launch_tiled(const GemmArgs &a, hipStream_t stream)
{
    // pick the tile that minimises (rounds over 256 CUs) x (tile area) / (tile efficiency): tile quantisation is the
    // main loss at M = B*197, and small tiles pay more LDS/L2 traffic per FLOP.  LDIT_GEMM_TILE=0|1|2|3 forces one
    // (tests use it to cover every instantiation).
    struct Cand { int bm, bn, id; double eff; };
    // ids 5-7: shorter panels of the same kernel (144 / 80 / 48 rows) for the mid-size batches, priced with a fixed cost of
    // ~24 rows per tile (the 128 weight rows are staged whatever the height)
    const Cand cands[7] = {{304, 128, 3, 1.0}, {320, 128, 0, 1.0}, {128, 128, 1, 0.9}, {64, 64, 2, 0.6},
                           {144, 128, 5, 144.0 / 168.0}, {80, 128, 6, 80.0 / 104.0}, {48, 128, 7, 48.0 / 72.0}};
    double best = -1.0;
    int pick = 2;
    for (const Cand &c : cands) {
        if (c.id >= 5 && (AMODE != A_ROWMAJOR || EPI == EPI_EMBED)) continue;
        const long tiles = (long)((a.M + c.bm - 1) / c.bm) * ((a.N + c.bn - 1) / c.bn);
        const long rounds = (tiles + 255) / 256;
        const double cost = (double)rounds * c.bm * c.bn / c.eff;
        if (best < 0 || cost < best) { best = cost; pick = c.id; }
    }
    if (const int force = diag().gemm_tile; force >= 0 && force != 4) pick = force;
    if ((AMODE != A_ROWMAJOR || EPI == EPI_EMBED) && pick >= 5) pick = 3;
    if (AMODE == A_CONV3 && pick == 3) pick = 0;                      // the panel kernel has no implicit-im2col loader
    if (pick == 3) return launch_gemm_panel(a, EPI, AMODE, stream);   // 304 x 128 panel tiling (gemm_panel_f32.hip)
    if (pick >= 5) return launch_gemm_panel(a, EPI, AMODE, stream, pick == 5 ? 144 : pick == 6 ? 80 : 48);
    switch (pick) {
        case 0: return launch_one<5, 2, EPI, AMODE>(a, stream);
        case 1: return launch_one<2, 2, EPI, AMODE>(a, stream);
        default: return launch_one<1, 1, EPI, AMODE>(a, stream);
    }
}

}  // namespace

#ifdef LDIT_GEMM_STAMPS
extern "C" int ldit_dbg_linear_stamps(const void *X, const void *W, const void *bias, void *Y, int M, int N, int K, int epi,
                                      const void *lam, const void *R, void *stamps, void *stream)
{
    GemmArgs a{};
    a.A = (const float *)X; a.W = (const float *)W; a.Y = (float *)Y; a.bias = (const float *)bias;
    a.lam = (const float *)lam; a.R = (const float *)R; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldy = N;
    a.stamps = (unsigned long long *)stamps;
    return launch_gemm(a, epi, A_ROWMAJOR, (hipStream_t)stream);
}
#endif

int launch_gemm(const GemmArgs &a, int epi, int amode, hipStream_t stream)
{
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return fail(LDIT_EINVAL, "gemm: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
    if (a.K % BK) return fail(LDIT_EUNSUPPORTED, "gemm: K=%d must be a multiple of %d", a.K, BK);
    // the kernels keep the offset INSIDE one block tile (<= 320 rows) as a 32-bit byte offset: 320 * 2^21 * 4 < 2^32
    if (amode != A_PATCH && (a.lda >= (1 << 21) || a.K >= (1 << 21)))
        return fail(LDIT_EUNSUPPORTED, "gemm: row stride lda=%d / K=%d must be below 2^21 elements", a.lda, a.K);
    if (!a.A || !a.W || !a.Y) return fail(LDIT_EINVAL, "gemm: null operand");
    if (!aligned16(a.A) || !aligned16(a.W) || (a.lda & 3)) return fail(LDIT_EINVAL, "gemm: operands must be 16-byte aligned");
    if (amode == A_CONV3) {
        if (epi != EPI_BIAS) return fail(LDIT_EINVAL, "gemm: the convolution loader feeds the bias epilogue only");
        if (a.conv_c % BK || a.K != 9 * a.conv_c || !a.zeros || a.conv_h <= 0 || a.conv_w <= 0 || a.conv_h >= 32768 || a.conv_w >= 32768)
            return fail(LDIT_EUNSUPPORTED, "gemm: 3x3 convolution needs channels %% 32 == 0, K = 9 channels, a zero page");
        return launch_tiled<EPI_BIAS, A_CONV3>(a, stream);
    }
    if (amode == A_PATCH) {
        if (epi != EPI_EMBED) return fail(LDIT_EINVAL, "gemm: patch gather only feeds the embedding epilogue");
        if ((a.patch & 3) || (a.img_w & 3)) return fail(LDIT_EUNSUPPORTED, "gemm: patch and image width must be multiples of 4");
        return launch_tiled<EPI_EMBED, A_PATCH>(a, stream);
    }
    // serving-size batches: the 32 x 32 / 16x16x4 kernel (same k order, bit-identical results; gemm_thin_f32.hip).
    // LDIT_GEMM_TILE=4 forces it for any M, 0..3 and 5..7 force one of the big tilings (tests cover every instantiation that way).
    {
        const int force = diag().gemm_tile;
        const bool thin = force >= 0 ? force == 4 : gemm_thin_prefers(a.M, a.N);
        if (thin && epi >= EPI_BIAS && epi <= EPI_SCALE_RESID) {
            if (epi == EPI_SCALE_RESID && (!a.lam || !a.R)) return fail(LDIT_EINVAL, "gemm: scale+residual epilogue needs lam and R");
            return launch_gemm_thin(a, epi, stream);
        }
    }
    switch (epi) {
        case EPI_BIAS: return launch_tiled<EPI_BIAS, A_ROWMAJOR>(a, stream);
        case EPI_BIAS_GELU: return launch_tiled<EPI_BIAS_GELU, A_ROWMAJOR>(a, stream);
        case EPI_SCALE_RESID:
            if (!a.lam || !a.R) return fail(LDIT_EINVAL, "gemm: scale+residual epilogue needs lam and R");
            return launch_tiled<EPI_SCALE_RESID, A_ROWMAJOR>(a, stream);
        default: return fail(LDIT_EINVAL, "gemm: unknown epilogue %d", epi);
    }
}

}  // namespace ldit
