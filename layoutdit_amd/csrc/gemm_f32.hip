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

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

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
    const float *src[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
        const int row = 8 * (wave + 4 * u) + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);      // logical 16-B chunk this lane fetches
        if (8 * (wave + 4 * u) < BM) {
            int gm = m0 + row;
            gm = gm < p.M ? gm : p.M - 1;
            if (AMODE == A_PATCH) {
                const int b = gm / p.patches, pi = gm - b * p.patches;
                const int gy = pi / p.gw, gx = pi - gy * p.gw;
                // per-image channel stride folded in at issue time (depends on k)
                src[u] = p.A + ((size_t)b * p.lda + (size_t)gy * p.patch * p.img_w + (size_t)gx * p.patch);
            } else {
                src[u] = p.A + (size_t)gm * p.lda + c * 4;
            }
        } else {
            int gn = n0 + row - BM;
            gn = gn < p.N ? gn : p.N - 1;
            src[u] = p.W + (size_t)gn * p.K + c * 4;
        }
    }

    auto issue = [&](int stage, int k0) {
        char *base = smem + stage * (ROWS * ROW_BYTES);
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int piece = wave + 4 * u;
            const float *g;
            if (AMODE == A_PATCH && 8 * piece < BM) {
                // k = (ch, dy, dx) with dx fastest; this lane's chunk starts at k0 + 4*c, c recovered from src
                const int row = 8 * piece + (lane >> 3);
                const int c = (lane & 7) ^ ((row >> 1) & 7);
                const int k = k0 + 4 * c, pp = p.patch * p.patch;
                const int ch = k / pp, rem = k - ch * pp, dy = rem / p.patch, dx = rem - dy * p.patch;
                g = src[u] + ((size_t)ch * p.img_h + dy) * p.img_w + dx;
            } else {
                g = src[u] + k0;
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
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();   // drains this wave's DMA (vmcnt 0), then every wave has finished reading the other stage
        if (kt + 1 < nk) issue((kt + 1) & 1, (kt + 1) * BK);
        const char *As = smem + (kt & 1) * (ROWS * ROW_BYTES) + (wm * TM * 32 + li) * ROW_BYTES;
        const char *Bs = smem + (kt & 1) * (ROWS * ROW_BYTES) + (BM + wn * TN * 32 + li) * ROW_BYTES;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            const int off = ((kc * 2 + lh) ^ sw) * 16;
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4 *>(As + i * 32 * ROW_BYTES + off);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4 *>(Bs + j * 32 * ROW_BYTES + off);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue ---------------------------------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 32 + j * 32 + li;
        const bool nok = n < p.N;
        const float bias = (nok && p.bias) ? p.bias[n] : 0.0f;
        float lam = 0.0f;
        if (EPI == EPI_SCALE_RESID) lam = nok ? p.lam[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * TM * 32 + i * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
                if (!nok || m >= p.M) continue;
                float v = acc[i][j][r] + bias;
                size_t o;
                if (EPI == EPI_EMBED) {
                    const int b = m / p.patches, pi = m - b * p.patches;
                    v += p.pos[(size_t)(1 + pi) * p.N + n];
                    o = ((size_t)b * p.tokens + 1 + pi) * p.ldy + n;
                } else {
                    o = (size_t)m * p.ldy + n;
                    if (EPI == EPI_BIAS_GELU) v = gelu_erf(v);
                    if (EPI == EPI_SCALE_RESID) v = p.R[o] + lam * v;
                }
                p.Y[o] = v;
                if (p.Y2) p.Y2[o] = v;
            }
        }
    }
}

template <int TM, int TN, int EPI, int AMODE>
int launch_one(const GemmArgs &a, hipStream_t stream)
{
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int lds = 2 * (BM + BN) * ROW_BYTES;
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    auto kern = gemm_f32_mfma<TM, TN, EPI, AMODE>;
    static bool attr_set = false;   // benign race: idempotent attribute
    if (!attr_set) {
        LDIT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), lds, stream, a);
    LDIT_HIP_CHECK(hipGetLastError());
    return LDIT_OK;
}

template <int EPI, int AMODE>
int launch_tiled(const GemmArgs &a, hipStream_t stream)
{
    // pick the tile that minimises (rounds over 256 CUs) x (tile area) / (tile efficiency): tile quantisation is the
    // main loss at M = B*197, and small tiles pay more LDS/L2 traffic per FLOP.  LDIT_GEMM_TILE=0|1|2 forces one
    // (tests use it to cover every instantiation).
    struct Cand { int bm, bn, id; double eff; };
    const Cand cands[3] = {{320, 128, 0, 1.0}, {128, 128, 1, 0.9}, {64, 64, 2, 0.6}};
    double best = -1.0;
    int pick = 2;
    for (const Cand &c : cands) {
        const long tiles = (long)((a.M + c.bm - 1) / c.bm) * ((a.N + c.bn - 1) / c.bn);
        const long rounds = (tiles + 255) / 256;
        const double cost = (double)rounds * c.bm * c.bn / c.eff;
        if (best < 0 || cost < best) { best = cost; pick = c.id; }
    }
    if (const char *force = getenv("LDIT_GEMM_TILE")) {
        if (force[0] >= '0' && force[0] <= '2' && force[1] == 0) pick = force[0] - '0';
    }
    switch (pick) {
        case 0: return launch_one<5, 2, EPI, AMODE>(a, stream);
        case 1: return launch_one<2, 2, EPI, AMODE>(a, stream);
        default: return launch_one<1, 1, EPI, AMODE>(a, stream);
    }
}

}  // namespace

int launch_gemm(const GemmArgs &a, int epi, int amode, hipStream_t stream)
{
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return fail(LDIT_EINVAL, "gemm: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
    if (a.K % BK) return fail(LDIT_EUNSUPPORTED, "gemm: K=%d must be a multiple of %d", a.K, BK);
    if (!a.A || !a.W || !a.Y) return fail(LDIT_EINVAL, "gemm: null operand");
    if (!aligned16(a.A) || !aligned16(a.W) || (a.lda & 3)) return fail(LDIT_EINVAL, "gemm: operands must be 16-byte aligned");
    if (amode == A_PATCH) {
        if (epi != EPI_EMBED) return fail(LDIT_EINVAL, "gemm: patch gather only feeds the embedding epilogue");
        if ((a.patch & 3) || (a.img_w & 3)) return fail(LDIT_EUNSUPPORTED, "gemm: patch and image width must be multiples of 4");
        return launch_tiled<EPI_EMBED, A_PATCH>(a, stream);
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
