// Shared by gemm_bf16.hip (K-contiguous operands) and gemm_bf16_tr.hip (operands read through transposing LDS reads):
// launch arguments and the direct (ragged-tile) epilogue of the swapped-operand bf16 GEMMs.
#pragma once
#include "ldit_common.h"
#include "epilogue_rows.h"

namespace ldit {
namespace {

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int BKB = 64;            // bf16 per k-tile: one 128-B LDS row
constexpr int ROWB = 128;

__device__ __forceinline__ void glds16h(const void *gsrc, char *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}


// LDS-DMA piece with a wave-uniform 64-bit base in SGPRs and a per-lane 32-bit BYTE offset: `global_load_lds_dwordx4 voff, s[base]`.
// The builtin takes a per-lane 64-bit pointer: two or three VALU instructions per piece (offset + k, widen, add) and a live VGPR
// pair; here the k-tile's advance lives in the scalar base and the lane offset is a constant of the kernel.  M0 (the LDS
// destination) is compiler-reserved: saved and restored inside the statement (cdna guide 5.7).  Invisible to hipcc's waitcnt pass:
// every consumer of the staged data sits behind an explicit s_waitcnt vmcnt.
__device__ __forceinline__ void glds16h_sbase(const void *ubase, unsigned voff_bytes, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff_bytes), "s"(ubase), "s"(lds_dst)
                 : "memory");
}

// The same piece from a per-lane 64-bit address (ragged tiles, zero-page redirects), also as an asm statement: a kernel whose every
// LDS-DMA is invisible to hipcc keeps its LDS reads free of the vmcnt(0) guards hipcc puts in front of reads that may alias a
// builtin LDS-DMA in flight.  `lds_dst` must be wave-uniform (it travels in M0).
__device__ __forceinline__ void glds16h_vaddr(const void *gsrc, unsigned lds_dst)
{
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst)
                 : "memory");
}

// a wave-uniform pointer that hipcc computed on the vector unit (integer divisions, 64-bit multiplies) -> scalar registers
template <typename T>
__device__ __forceinline__ const T *uniform_ptr(const T *q)
{
    const unsigned long long v = reinterpret_cast<unsigned long long>(q);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const T *>(((unsigned long long)hi << 32) | lo);
}

struct GemmArgsH {
    const bf16_t *A;     // [M, K] bf16
    const bf16_t *W;     // [N, K] bf16
    void *Y;             // [M, N] bf16 (BIAS, BIAS_GELU) or fp32 (SCALE_RESID)
    float *Y2;           // optional fp32 tap copy (SCALE_RESID)
    const float *bias;   // [N] fp32 or null
    const float *lam;    // [N] fp32
    const float *R;      // [M, N] fp32, may alias Y
    int M, N, K, lda, ldy;
    int ldw;             // gemm_bf16_tr.hip only: row stride of the reduction-major W operand
    int direct_epi;      // LDIT_GEMM_DIRECT_EPILOGUE=1: interior tiles stored straight from the accumulators (A/B experiments)
    GemmExtra x;         // train-step operands (Ypre, rowscale, aux, split-K); defaults = inference
};

template <int EPI> constexpr bool f32_out() { return EPI == EPI_SCALE_RESID || EPI == EPI_F32 || EPI == EPI_EMBED; }

// MODE 0: tile inside the matrix, 16-B / 8-B accesses unchecked; MODE 1: columns inside, rows past M skipped (the ragged
// last row tile keeps its vector accesses - a lane owns a row, so the element-wise path does not coalesce and cost ~20 us);
// MODE 2: element-wise with checks.
// L16: the accumulators are 2 TM x 2 TN tiles of 16 x 16 (v_mfma_f32_16x16x32_bf16: acc[i][j][e] = row 16 i + (lane & 15), column
// 16 j + 4 (lane >> 4) + e): per 32 x 32 block a lane then owns two rows x two column quads instead of one row x four quads.  The
// arithmetic per element is the same statement either way (same bits).
// NQU (L16 only): column-quad groups of each 32-column block that exist in `acc` - 1 = the wave owns 16 columns per block, acc is
// [2 TM][TN] (gemm_bf16_tail: one wave per 32 x 16 tile).
template <int TM, int TN, int EPI, int MODE, bool L16 = false, int NQU = 0, typename ACC>
__device__ __forceinline__ void store_h(const GemmArgsH &p, const ACC &acc, int mw, int nw, int lane)
{
    constexpr int NQ = NQU > 0 ? NQU : (L16 ? 2 : 4), NR = L16 ? 2 : 1;      // column quads per row, rows - per lane and 32 x 32 block
    constexpr int JW = (L16 && NQU == 1) ? 1 : 2;           // 16-column accumulator tiles per 32-column block
    const int row_in = L16 ? (lane & 15) : (lane & 31), quad_in = L16 ? 4 * (lane >> 4) : 4 * (lane >> 5);
    constexpr int QSTEP = L16 ? 16 : 8;
    const bool dual = p.Y2 != nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        f32x4 bias[NQ], lam[NQ];
#pragma unroll
        for (int g = 0; g < NQ; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = nw + j * 32 + QSTEP * g + quad_in + e;
                const bool ok = MODE != 2 || n < p.N;
                bias[g][e] = (ok && p.bias) ? p.bias[n] : 0.0f;
                lam[g][e] = (EPI == EPI_SCALE_RESID && ok) ? p.lam[n] : 0.0f;
            }
#pragma unroll
        for (int ir = 0; ir < TM * NR; ++ir) {
            const int i = ir / NR, rr = ir % NR;
            const int m = mw + i * 32 + 16 * rr + row_in;
            if (MODE != 0 && m >= p.M) continue;
            f32x4 res[NQ];
            if (EPI == EPI_SCALE_RESID) {
#pragma unroll
                for (int g = 0; g < NQ; ++g) {
                    const int n = nw + j * 32 + QSTEP * g + quad_in;
                    const unsigned o = (unsigned)m * (unsigned)p.ldy + (unsigned)n;
                    if (MODE != 2) res[g] = *reinterpret_cast<const f32x4 *>(p.R + o);
                    else
#pragma unroll
                        for (int e = 0; e < 4; ++e) res[g][e] = (n + e < p.N) ? p.R[o + e] : 0.0f;
                }
                asm volatile("" ::: "memory");
            }
            const float rs = (EPI == EPI_SCALE_RESID && p.x.rowscale) ? p.x.rowscale[m] : 1.0f;
            const unsigned img = EPI == EPI_EMBED ? (unsigned)m / (unsigned)p.x.patches : 0u;
#pragma unroll
            for (int g = 0; g < NQ; ++g) {
                const int n = nw + j * 32 + QSTEP * g + quad_in;
                unsigned o = (unsigned)m * (unsigned)p.ldy + (unsigned)n;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if constexpr (L16) v[e] = acc[2 * i + rr][JW * j + g][e] + bias[g][e];
                    else v[e] = acc[i][j][4 * g + e] + bias[g][e];
                }
                if (EPI == EPI_EMBED) {
                    // token row of patch row m (one CLS slot in front of every image) + its position row, as the slab epilogue
                    const float *pq = p.x.pos + (((unsigned)m - img * (unsigned)p.x.patches + 1u) * (unsigned)p.ldy + (unsigned)n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (MODE != 2 || n + e < p.N) ? pq[e] : 0.0f;
                    o = ((unsigned)m + img + 1u) * (unsigned)p.ldy + (unsigned)n;
                }
                bool gelu_done = false;
                if ((EPI == EPI_BIAS_GELU || EPI == EPI_SCALE_RESID) && p.x.Ypre) {
                    // saved for the backward: the pre-LayerScale value, or the GELU derivative at the pre-activation (train forward:
                    // the erf form and ITS derivative from one set of sub-expressions, exactly as the slab epilogue computes them)
                    f32x4 sv = v;
                    if (EPI == EPI_BIAS_GELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float ge, gr;
                            gelu_and_grad_lp(v[e], ge, gr);
                            v[e] = ge;
                            sv[e] = gr;
                        }
                        gelu_done = true;
                    }
                    bf16_t *yp = static_cast<bf16_t *>(p.x.Ypre);
                    if (MODE != 2) {
                        const bf16x4 pk = {(bf16_t)sv[0], (bf16_t)sv[1], (bf16_t)sv[2], (bf16_t)sv[3]};
                        *reinterpret_cast<bf16x4 *>(yp + o) = pk;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (n + e < p.N) yp[o + e] = (bf16_t)sv[e];
                    }
                }
                if (EPI == EPI_GELU_SPLIT) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                }
                if (EPI == EPI_GELU_BWD) {
                    const bf16_t *ax = static_cast<const bf16_t *>(p.x.aux) + ((unsigned)m * (unsigned)p.x.ldaux + (unsigned)n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= (MODE != 2 || n + e < p.N) ? (float)ax[e] : 0.0f;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = v[e];
                    if (EPI == EPI_BIAS_GELU && !gelu_done) t = gelu_lp(t);
                    if (EPI == EPI_SCALE_RESID) t = p.x.rowscale ? __builtin_fmaf(lam[g][e] * rs, t, res[g][e]) : __builtin_fmaf(lam[g][e], t, res[g][e]);
                    v[e] = t;
                }
                if (f32_out<EPI>()) {
                    float *y = static_cast<float *>(p.Y);
                    if (MODE != 2) {
                        *reinterpret_cast<f32x4 *>(y + o) = v;
                        if (dual) *reinterpret_cast<f32x4 *>(p.Y2 + o) = v;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (n + e < p.N) { y[o + e] = v[e]; if (dual) p.Y2[o + e] = v[e]; }
                    }
                } else if (EPI == EPI_GELU_SPLIT || EPI == EPI_BIAS_SPLIT) {
                    bf16_t *y = static_cast<bf16_t *>(p.Y);
                    const unsigned plane = (unsigned)p.ldy / (unsigned)p.x.nsplit_out;
                    f32x4 r = v;
                    for (int sp = 0; sp < p.x.nsplit_out; ++sp) {
                        const bf16x4 pk = {(bf16_t)r[0], (bf16_t)r[1], (bf16_t)r[2], (bf16_t)r[3]};
                        if (MODE != 2) *reinterpret_cast<bf16x4 *>(y + o + sp * plane) = pk;
                        else
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (n + e < p.N) y[o + sp * plane + e] = pk[e];
#pragma unroll
                        for (int e = 0; e < 4; ++e) r[e] -= (float)pk[e];
                    }
                } else {
                    bf16_t *y = static_cast<bf16_t *>(p.Y);
                    if (MODE != 2) {
                        const bf16x4 pk = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                        *reinterpret_cast<bf16x4 *>(y + o) = pk;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (n + e < p.N) y[o + e] = (bf16_t)v[e];
                    }
                }
            }
            if (EPI == EPI_SCALE_RESID) asm volatile("" ::: "memory");
        }
    }
}

}  // namespace
}  // namespace ldit
