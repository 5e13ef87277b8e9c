// Row-contiguous epilogue for the swapped-operand MFMA GEMMs (gemm_bf16.hip, gemm_fp8.hip).
//
// In those kernels an accumulator tile has the output ROW on the lane (acc[i][j][4g+e] = row 32i + (lane&31), column
// 32j + 8g + 4(lane>>5) + e), so storing straight from the registers makes every store instruction touch 64 different
// rows with 8 or 16 bytes each - the epilogue becomes store-issue bound (it measured ~25 k cycles per 256 x 256 bf16 tile,
// as long as half a K = 1024 main loop).  Here each wave passes its 32-row x 64-column slabs through a private LDS
// buffer (the k-loop's stage memory, free by then) and writes them back row-major: 16 lanes cover one 64-column row
// segment, so every global access is a full contiguous 64 ... 256-byte run, and the residual is read the same way.
#pragma once

#include "ldit_common.h"

namespace ldit {

typedef __bf16 epi_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 epi_bf16x8 __attribute__((ext_vector_type(8)));

constexpr int EPI_ROW_BYTES = 272;                    // 64 fp32 + 16 B pad: 68 dwords -> b128 writes of 8 lanes hit 32 banks once
constexpr int EPI_WAVE_BYTES = 32 * EPI_ROW_BYTES;    // one 32 x 64 slab per wave

enum { EPI_OUT_BF16 = 0, EPI_OUT_FP8 = 1, EPI_OUT_F32 = 2,
       EPI_OUT_SPLIT = 3 };   // x.nsplit_out bf16 planes of the fp32 value, side by side in the row (split-fp32 build)

// acc: the wave's TM x TN tiles; (mw, nw): its first row / column; buf: EPI_WAVE_BYTES of LDS private to this wave.
// The RAW accumulators travel through LDS; all arithmetic happens on the read-back side, where a lane owns four fixed
// columns (one bias / scale / lambda quad per 64-column slab instead of eight):
//   t = fma(acc, ab * wscale[n], bias[n]) ; GELU ; (resid: fma(lam[n], t, R)) ; bf16 / fp8 (times oinv, saturating) / fp32.
// `wscale` (optional, fp8 build) = per-output-channel weight scales; `ab` = the per-tensor part (activation scale, or
// activation x weight scale when wscale is null; 1 for bf16).  The arithmetic is spelled with explicit fmas, identically
// to the direct stores of the ragged tiles, so that a row gets the same bits whichever path its tile takes (batch
// invariance of the bf16 / fp8 builds).
// `x` (train step only): bf16 copy of t before GELU / LayerScale (Ypre), per-row factor on lam (rowscale), pre-activation of
// the GELU derivative (aux).
// L16: the accumulators are 2 TM x 2 TN tiles of 16 x 16 (v_mfma_f32_16x16x32_*: acc[i][j][e] = row 16 i + (lane & 15), column
// 16 j + 4 (lane >> 4) + e) instead of TM x TN tiles of 32 x 32; only the write into the slab differs.
// mrows: rows of the matrix (rows >= mrows are neither read nor written: the ragged last row tile may take this path too).
// csum (gemm_bf16_tr.hip, bf16 output only): fp32 [.., ldy]: receives, at row 0, the column sums of the values this wave stores
// (as rounded to bf16), rows < mrows only - the caller hands in the partial row of this (tile, wave row).
template <int TM, int TN, int EPI, int OUT, bool L16 = false, typename ACC>
__device__ __forceinline__ void store_rows_via_lds(const ACC &acc, char *buf, void *Yv, float *Y2, const float *R,
                                                   const float *bias, const float *lam, const float *wscale, int ldy, int mw,
                                                   int nw, int lane, float ab, float oinv, const GemmExtra x = GemmExtra{},
                                                   int mrows = 0x7fffffff, float *csum = nullptr)
{
    static_assert(TN % 2 == 0, "slabs are 64 columns wide");
    const int c32 = lane & 31, h = lane >> 5;
    // Read-back geometry.  fp32 output: a lane owns 4 columns (16 lanes per 64-column row segment, 4 rows per pass, 16-byte stores).
    // bf16 / plane output (round 4): a lane owns EIGHT columns (8 lanes per row segment, 8 rows per pass) so that its store is
    // 16 bytes of bf16 too - with 4 columns the bf16 epilogue issued twice the store instructions of the fp32 one for half the
    // bytes and was SLOWER than it (dgrad N = 3072, K = 768: 85 us with a bf16 output, 78 us with an fp32 one).  The arithmetic per
    // element is unchanged, so a row keeps its bits.  Callers take this path only when ldy (and ldaux) are multiples of 8 then.
    constexpr int NC = (OUT == EPI_OUT_F32 || OUT == EPI_OUT_FP8) ? 4 : 8, NQ = NC / 4, LPR = 64 / NC, RPP = 64 / LPR, NPASS = 32 / RPP;
    const int rrow = lane / LPR, rq = lane % LPR;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jp = 0; jp < TN / 2; ++jp) {
        const unsigned n = (unsigned)(nw + 64 * jp + NC * rq);
        f32x4 biasq[NQ], lamq[NQ], abq[NQ], colacc[NQ];
#pragma unroll
        for (int hq = 0; hq < NQ; ++hq) {
            biasq[hq] = bias ? *reinterpret_cast<const f32x4 *>(bias + n + 4 * hq) : zero4;
            lamq[hq] = EPI == EPI_SCALE_RESID ? *reinterpret_cast<const f32x4 *>(lam + n + 4 * hq) : zero4;
            colacc[hq] = zero4;
            abq[hq] = f32x4{ab, ab, ab, ab};
            if (wscale) {
                const f32x4 w = *reinterpret_cast<const f32x4 *>(wscale + n + 4 * hq);
#pragma unroll
                for (int e = 0; e < 4; ++e) abq[hq][e] = ab * w[e];
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            // The slab's global operands (residual rows, stochastic-depth factors, GELU-derivative rows) are fetched HERE, all passes at
            // once and without a branch in between, before the accumulators even enter LDS.  Fetched inside the pass loop (rounds 1 - 4a:
            // load, s_waitcnt vmcnt(0), fma, store, per pass) every pass also sat out the ACKNOWLEDGEMENT of the previous pass's
            // stores - vmcnt counts stores too and retires in order - eight times per slab.  Rows past `mrows` read row mrows - 1.
            f32x4 resq[(EPI == EPI_SCALE_RESID && OUT == EPI_OUT_F32) ? NPASS : 1];
            float rsq[(EPI == EPI_SCALE_RESID && OUT == EPI_OUT_F32) ? NPASS : 1];
            epi_bf16x8 auxq[(EPI == EPI_GELU_BWD && NQ == 2) ? NPASS : 1];
            if constexpr (EPI == EPI_SCALE_RESID && OUT == EPI_OUT_F32) {
#pragma unroll
                for (int r = 0; r < NPASS; ++r) {
                    int rg = mw + 32 * i + RPP * r + rrow;
                    rg = rg < mrows ? rg : mrows - 1;
                    resq[r] = *reinterpret_cast<const f32x4 *>(R + ((unsigned)rg * (unsigned)ldy + n));
                    rsq[r] = 1.0f;
                }
                if (x.rowscale) {
#pragma unroll
                    for (int r = 0; r < NPASS; ++r) {
                        int rg = mw + 32 * i + RPP * r + rrow;
                        rg = rg < mrows ? rg : mrows - 1;
                        rsq[r] = x.rowscale[rg];
                    }
                }
            }
            if constexpr (EPI == EPI_GELU_BWD && NQ == 2) {
#pragma unroll
                for (int r = 0; r < NPASS; ++r) {
                    int rg = mw + 32 * i + RPP * r + rrow;
                    rg = rg < mrows ? rg : mrows - 1;
                    auxq[r] = *reinterpret_cast<const epi_bf16x8 *>(static_cast<const __bf16 *>(x.aux) + ((unsigned)rg * (unsigned)x.ldaux + n));
                }
            }
            if constexpr (L16) {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const f32x4 v = {acc[2 * i + ii][4 * jp + jj][0], acc[2 * i + ii][4 * jp + jj][1], acc[2 * i + ii][4 * jp + jj][2],
                                         acc[2 * i + ii][4 * jp + jj][3]};
                        *reinterpret_cast<f32x4 *>(buf + (16 * ii + (lane & 15)) * EPI_ROW_BYTES + (16 * jj + 4 * (lane >> 4)) * 4) = v;
                    }
            } else {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = {acc[i][2 * jp + jj][4 * g + 0], acc[i][2 * jp + jj][4 * g + 1], acc[i][2 * jp + jj][4 * g + 2],
                                     acc[i][2 * jp + jj][4 * g + 3]};
                    *reinterpret_cast<f32x4 *>(buf + c32 * EPI_ROW_BYTES + (32 * jj + 8 * g + 4 * h) * 4) = v;
                }
            }
            // ONE wait for the slab's fetches, here - the empty asm statements take the fetched registers as operands, so hipcc waits for
            // all of them now (under the slab's LDS traffic) and knows of no pending load inside the pass loop: a vmcnt(N) there, however
            // generous, would also wait until all but N of the STORES issued so far are acknowledged
            if constexpr (EPI == EPI_SCALE_RESID && OUT == EPI_OUT_F32) {
#pragma unroll
                for (int r = 0; r < NPASS; ++r) asm volatile("" : "+v"(resq[r]), "+v"(rsq[r]));
            }
            if constexpr (EPI == EPI_GELU_BWD && NQ == 2) {
#pragma unroll
                for (int r = 0; r < NPASS; ++r) asm volatile("" : "+v"(auxq[r]));
            }
#pragma unroll
            for (int r = 0; r < NPASS; ++r) {
                const int row = RPP * r + rrow;
                if (mw + 32 * i + row >= mrows) continue;
                f32x4 v[NQ];
#pragma unroll
                for (int hq = 0; hq < NQ; ++hq) v[hq] = *reinterpret_cast<const f32x4 *>(buf + row * EPI_ROW_BYTES + (rq * NC + 4 * hq) * 4);
                unsigned o = (unsigned)(mw + 32 * i + row) * (unsigned)ldy + n;
#pragma unroll
                for (int hq = 0; hq < NQ; ++hq)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[hq][e] = __builtin_fmaf(v[hq][e], abq[hq][e], biasq[hq][e]);
                if (EPI == EPI_EMBED) {
                    const unsigned m = (unsigned)(mw + 32 * i + row), img = m / (unsigned)x.patches, pi = m - img * (unsigned)x.patches;
#pragma unroll
                    for (int hq = 0; hq < NQ; ++hq) {
                        const f32x4 pq = *reinterpret_cast<const f32x4 *>(x.pos + ((pi + 1u) * (unsigned)ldy + n + 4 * hq));
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[hq][e] += pq[e];
                    }
                    o = (m + img + 1u) * (unsigned)ldy + n;
                }
                if (EPI == EPI_SCALE_RESID && x.Ypre) {
#pragma unroll
                    for (int hq = 0; hq < NQ; ++hq) {
                        const epi_bf16x4 pre = {(__bf16)v[hq][0], (__bf16)v[hq][1], (__bf16)v[hq][2], (__bf16)v[hq][3]};
                        *reinterpret_cast<epi_bf16x4 *>(static_cast<__bf16 *>(x.Ypre) + o + 4 * hq) = pre;
                    }
                }
                if (EPI == EPI_BIAS_GELU) {
                    if (x.Ypre) {
                        f32x4 gp[NQ];
#pragma unroll
                        for (int hq = 0; hq < NQ; ++hq)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float ge, gr;
                                gelu_and_grad_lp(v[hq][e], ge, gr);
                                v[hq][e] = ge;
                                gp[hq][e] = gr;
                            }
                        if constexpr (NQ == 2) {
                            const epi_bf16x8 pre = {(__bf16)gp[0][0], (__bf16)gp[0][1], (__bf16)gp[0][2], (__bf16)gp[0][3],
                                                    (__bf16)gp[NQ - 1][0], (__bf16)gp[NQ - 1][1], (__bf16)gp[NQ - 1][2], (__bf16)gp[NQ - 1][3]};
                            *reinterpret_cast<epi_bf16x8 *>(static_cast<__bf16 *>(x.Ypre) + o) = pre;
                        } else {
                            const epi_bf16x4 pre = {(__bf16)gp[0][0], (__bf16)gp[0][1], (__bf16)gp[0][2], (__bf16)gp[0][3]};
                            *reinterpret_cast<epi_bf16x4 *>(static_cast<__bf16 *>(x.Ypre) + o) = pre;
                        }
                    } else {
#pragma unroll
                        for (int hq = 0; hq < NQ; ++hq)
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[hq][e] = gelu_lp(v[hq][e]);
                    }
                }
                if (EPI == EPI_GELU_SPLIT) {
#pragma unroll
                    for (int hq = 0; hq < NQ; ++hq)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[hq][e] = gelu_erf(v[hq][e]);
                }
                if (EPI == EPI_GELU_BWD) {
                    const __bf16 *ap = static_cast<const __bf16 *>(x.aux) + ((unsigned)(mw + 32 * i + row) * (unsigned)x.ldaux + n);
                    if constexpr (NQ == 2) {
                        (void)ap;
                        const epi_bf16x8 a = auxq[r];
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[0][e] *= (float)a[e]; v[NQ - 1][e] *= (float)a[4 + e]; }
                    } else {
                        const epi_bf16x4 a = *reinterpret_cast<const epi_bf16x4 *>(ap);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[0][e] *= (float)a[e];
                    }
                }
                if constexpr (OUT == EPI_OUT_F32) {
                    if (EPI == EPI_SCALE_RESID) {
                        const f32x4 res = resq[r];
                        if (x.rowscale) {
                            const float rs = rsq[r];
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[0][e] = __builtin_fmaf(lamq[0][e] * rs, v[0][e], res[e]);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[0][e] = __builtin_fmaf(lamq[0][e], v[0][e], res[e]);
                        }
                    }
                    *reinterpret_cast<f32x4 *>(static_cast<float *>(Yv) + o) = v[0];
                    if (Y2) *reinterpret_cast<f32x4 *>(Y2 + o) = v[0];
                } else if constexpr (OUT == EPI_OUT_SPLIT) {
                    // planes p0 = bf16(v), p1 = bf16(v - p0), (p2 = bf16(v - p0 - p1)): plane s at column s * (ldy / nsplit_out)
                    const unsigned plane = (unsigned)ldy / (unsigned)x.nsplit_out;
                    f32x4 rr[NQ];
#pragma unroll
                    for (int hq = 0; hq < NQ; ++hq) rr[hq] = v[hq];
                    for (int sp = 0; sp < x.nsplit_out; ++sp) {
                        const epi_bf16x8 pk = {(__bf16)rr[0][0], (__bf16)rr[0][1], (__bf16)rr[0][2], (__bf16)rr[0][3],
                                               (__bf16)rr[NQ - 1][0], (__bf16)rr[NQ - 1][1], (__bf16)rr[NQ - 1][2], (__bf16)rr[NQ - 1][3]};
                        *reinterpret_cast<epi_bf16x8 *>(static_cast<__bf16 *>(Yv) + o + sp * plane) = pk;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { rr[0][e] -= (float)pk[e]; rr[NQ - 1][e] -= (float)pk[4 + e]; }
                    }
                } else if constexpr (OUT == EPI_OUT_FP8) {
                    // (four columns per lane as the fp32 output: eight - 8-byte stores - measured 0.3 % slower on config 4)
                    *reinterpret_cast<unsigned *>(static_cast<unsigned char *>(Yv) + o) =
                        pack_fp8x4(v[0][0] * oinv, v[0][1] * oinv, v[0][2] * oinv, v[0][3] * oinv);
                } else {
                    const epi_bf16x8 pk = {(__bf16)v[0][0], (__bf16)v[0][1], (__bf16)v[0][2], (__bf16)v[0][3],
                                           (__bf16)v[NQ - 1][0], (__bf16)v[NQ - 1][1], (__bf16)v[NQ - 1][2], (__bf16)v[NQ - 1][3]};
                    *reinterpret_cast<epi_bf16x8 *>(static_cast<__bf16 *>(Yv) + o) = pk;
                    if (csum) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { colacc[0][e] += (float)pk[e]; colacc[NQ - 1][e] += (float)pk[4 + e]; }
                    }
                }
            }
        }
        if (OUT == EPI_OUT_BF16 && csum) {
            // the row groups (lane / LPR) of a column octet -> lane group 0, fixed order; two 16-B stores per 64-column slab and lane
#pragma unroll
            for (int hq = 0; hq < NQ; ++hq)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = colacc[hq][e];
#pragma unroll
                    for (int sh = LPR; sh < 64; sh <<= 1) t += __shfl_xor(t, sh, 64);
                    colacc[hq][e] = t;
                }
            if (rrow == 0) {
#pragma unroll
                for (int hq = 0; hq < NQ; ++hq) *reinterpret_cast<f32x4 *>(csum + n + 4 * hq) = colacc[hq];
            }
        }
    }
}

}  // namespace ldit
