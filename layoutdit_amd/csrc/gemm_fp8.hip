// fp8 (OCP e4m3, gfx950) MFMA GEMM with per-tensor scales:  Y[M,N] = epi( sa*sw * (A8[M,K] . W8[N,K]^T) ),
// A8, W8 fp8 e4m3 (K-contiguous), fp32 accumulate.  Groundwork for BASELINE configs[4] (fp8 forward).
//
// Same skeleton as gemm_bf16.hip: K-contiguous operands -> LDS by LDS-DMA, 128-B LDS rows (128 fp8), 16-B chunk
// XOR-swizzled with (row>>1)&7, two stages, 2 x 4 waves (two per SIMD) on a 256 x 256 tile.  A lane's b128 fragment read
// holds 16 consecutive fp8 of its row: the low 8 bytes feed one v_mfma_f32_32x32x16_fp8_fp8, the high 8 bytes the next
// (a permutation of the k order applied identically to both operands), so each LDS read pays for two MFMAs.
// The non-scaled fp8 MFMA issues at the bf16 rate; what fp8 buys here is half the DMA / LDS bytes per FLOP.
// Epilogues: bias -> bf16 ; bias + erf-GELU -> fp8 (times out_inv_scale, saturated to +-448) ; R + lam (.) (.) -> fp32.
#include <cstdlib>

#include "ldit_common.h"
#include "epilogue_rows.h"

namespace ldit {

namespace {

typedef long i64x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_8 __attribute__((ext_vector_type(4)));
constexpr int BKE = 128, ROW8 = 128;

__device__ __forceinline__ void glds16q(const void *gsrc, char *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

struct GemmArgs8 {
    const unsigned char *A, *W;   // fp8 e4m3
    void *Y;                      // bf16 (BIAS) / fp8 (BIAS_GELU) / fp32 (SCALE_RESID)
    float *Y2;                    // optional fp32 tap copy (SCALE_RESID)
    const float *bias, *lam, *R;
    int M, N, K, lda, ldy;
    float ab_scale;               // per-tensor dequantisation of the accumulator: sa * sw, or sa alone when d_wrow is given
    float out_inv_scale;          // 1 / scale of the fp8 output (BIAS_GELU)
    const float *d_act, *d_out;   // device-resident sa / so: override the two host values when non-null
    const float *d_wrow;          // optional per-output-channel weight scales [N] (multiplied onto the per-tensor part)
    int direct_epi;               // LDIT_GEMM_DIRECT_EPILOGUE=1: interior tiles stored straight from the accumulators
};

// MODE 0: tile inside the matrix, 16-B / 8-B accesses unchecked; MODE 1: columns inside, rows past M skipped (the ragged
// last row tile keeps its vector accesses - a lane owns a row, so the element-wise path does not coalesce and cost ~20 us);
// MODE 2: element-wise with checks.
template <int TM, int TN, int EPI, int MODE>
__device__ __forceinline__ void store_q(const GemmArgs8 &p, const f32x16 (&acc)[TM][TN], int mw, int nw, int lane)
{
    const int c32 = lane & 31, h = lane >> 5;
    const bool dual = p.Y2 != nullptr;
    const float ab = p.d_act ? p.d_act[0] : p.ab_scale;
    const float oinv = (EPI == EPI_BIAS_GELU && p.d_out) ? 1.0f / p.d_out[0] : p.out_inv_scale;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        f32x4 bias[4], lam[4], abq[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = nw + j * 32 + 8 * g + 4 * h + e;
                const bool ok = MODE != 2 || n < p.N;
                bias[g][e] = (ok && p.bias) ? p.bias[n] : 0.0f;
                lam[g][e] = (EPI == EPI_SCALE_RESID && ok) ? p.lam[n] : 0.0f;
                abq[g][e] = (ok && p.d_wrow) ? ab * p.d_wrow[n] : ab;
            }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = mw + i * 32 + c32;
            if (MODE != 0 && m >= p.M) continue;
            f32x4 res[4];
            if (EPI == EPI_SCALE_RESID) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = nw + j * 32 + 8 * g + 4 * h;
                    const unsigned o = (unsigned)m * (unsigned)p.ldy + (unsigned)n;
                    if (MODE != 2) res[g] = *reinterpret_cast<const f32x4 *>(p.R + o);
                    else
#pragma unroll
                        for (int e = 0; e < 4; ++e) res[g][e] = (n + e < p.N) ? p.R[o + e] : 0.0f;
                }
                asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nw + j * 32 + 8 * g + 4 * h;
                const unsigned o = (unsigned)m * (unsigned)p.ldy + (unsigned)n;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = __builtin_fmaf(acc[i][j][4 * g + e], abq[g][e], bias[g][e]);
                    if (EPI == EPI_BIAS_GELU) t = gelu_lp(t);
                    if (EPI == EPI_SCALE_RESID) t = __builtin_fmaf(lam[g][e], t, res[g][e]);
                    v[e] = t;
                }
                if (EPI == EPI_SCALE_RESID) {
                    float *y = static_cast<float *>(p.Y);
                    if (MODE != 2) {
                        *reinterpret_cast<f32x4 *>(y + o) = v;
                        if (dual) *reinterpret_cast<f32x4 *>(p.Y2 + o) = v;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (n + e < p.N) { y[o + e] = v[e]; if (dual) p.Y2[o + e] = v[e]; }
                    }
                } else if (EPI == EPI_BIAS_GELU) {
                    unsigned char *y = static_cast<unsigned char *>(p.Y);
                    const unsigned pk = pack_fp8x4(v[0] * oinv, v[1] * oinv, v[2] * oinv, v[3] * oinv);
                    if (MODE != 2) *reinterpret_cast<unsigned *>(y + o) = pk;
                    else
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (n + e < p.N) y[o + e] = (unsigned char)(pk >> (8 * e));
                } else {
                    __bf16 *y = static_cast<__bf16 *>(p.Y);
                    if (MODE != 2) {
                        const bf16x4_8 pk = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                        *reinterpret_cast<bf16x4_8 *>(y + o) = pk;
                    } else
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (n + e < p.N) y[o + e] = (__bf16)v[e];
                }
            }
            if (EPI == EPI_SCALE_RESID) asm volatile("" ::: "memory");
        }
    }
}

// K64 = false: v_mfma_f32_32x32x16_fp8_fp8 (bf16 rate), two per b128 fragment read, four chunk steps per k-tile.
// K64 = true : v_mfma_f32_32x32x64_f8f6f4 with e4m3 operands and unit block scales (2x the bf16 rate): a lane's operand is
//              32 consecutive fp8 of its row (two b128 reads), two chunk steps per k-tile.
template <int WM, int WN, int TM, int TN, int EPI, bool K64>
__global__ void __launch_bounds__(64 * WM * WN, (WM * WN) / 4) gemm_fp8_mfma(const GemmArgs8 p)
{
    constexpr int NWAVES = WM * WN;
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, ROWS = BM + BN, NLD = ROWS / (8 * NWAVES);
    static_assert(ROWS % (8 * NWAVES) == 0 && BM % 8 == 0, "DMA pieces must split evenly over the waves");
    extern __shared__ __attribute__((aligned(128))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int c32 = lane & 31, h = lane >> 5;

    const int nbn = (p.N + BN - 1) / BN, nbm = (p.M + BM - 1) / BM;
    const int ntiles = nbm * nbn;
    int tile;
    {
        const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3, qq = ntiles >> 3, rr = ntiles & 7;
        tile = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    }
    const int m0 = (tile / nbn) * BM, n0 = (tile % nbn) * BN;

    unsigned src[NLD];   // byte (= element) offsets
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
        const int row = 8 * (wave + NWAVES * u) + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        if (8 * (wave + NWAVES * u) < BM) {
            int gm = m0 + row;
            gm = gm < p.M ? gm : p.M - 1;
            src[u] = (unsigned)gm * (unsigned)p.lda + c * 16;
        } else {
            int gn = n0 + row - BM;
            gn = gn < p.N ? gn : p.N - 1;
            src[u] = (unsigned)gn * (unsigned)p.K + c * 16;
        }
    }
    auto issue = [&](int stage, int k0) {
        // the k-tile's advance travels in a scalar base, the lane's row / chunk offset is the kernel constant src[u]: no vector address
        // arithmetic per piece (`global_load_lds_dwordx4 voff, s[base]`; invisible to hipcc's waitcnt pass - every hand-over below
        // carries its explicit s_waitcnt vmcnt)
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)((__attribute__((address_space(3))) char *)(smem + stage * (ROWS * ROW8))));
        const unsigned char *abase = p.A + k0, *wbase = p.W + k0;
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int piece = wave + NWAVES * u;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(src[u]), "s"(8 * piece < BM ? abase : wbase), "s"(dst + piece * 1024)
                         : "memory");
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int sw = (c32 >> 1) & 7;
    const int nk = p.K / BKE;
    const int a_row = (wm * TM * 32 + c32) * ROW8, b_row = (BM + wn * TN * 32 + c32) * ROW8;

    if constexpr (!K64) {
        auto load_frags = [&](int stage, int c, i64x2(&xa)[TM], i64x2(&wb)[TN]) {
            const char *base = smem + stage * (ROWS * ROW8) + ((c * 2 + h) ^ sw) * 16;
#pragma unroll
            for (int i = 0; i < TM; ++i) xa[i] = *reinterpret_cast<const i64x2 *>(base + a_row + i * 32 * ROW8);
#pragma unroll
            for (int j = 0; j < TN; ++j) wb[j] = *reinterpret_cast<const i64x2 *>(base + b_row + j * 32 * ROW8);
        };
        auto mfma_chunk = [&](const i64x2(&xa)[TM], const i64x2(&wb)[TN]) {
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(wb[j][half], xa[i][half], acc[i][j], 0, 0, 0);
        };
        i64x2 xa0[TM], wb0[TN], xa1[TM], wb1[TN];
        issue(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // explicit: the first tile has landed before anybody reads it
        __syncthreads();
        load_frags(0, 0, xa0, wb0);
        // (round 4: the order reads | MFMAs is pinned with sched_barrier - left alone, the scheduler sinks a chunk's reads behind its
        //  MFMAs, right in front of the wait that needs them: profiles/r04_tr_pinned_order_ab.txt, same finding as gemm_bf16_tr)
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            const int knext = (kt + 1 < nk ? kt + 1 : nk - 1) * BKE;
            load_frags(cur, 1, xa1, wb1);
            __builtin_amdgcn_sched_barrier(0);
            issue(cur ^ 1, knext);
            mfma_chunk(xa0, wb0);
            __builtin_amdgcn_sched_barrier(0);
            load_frags(cur, 2, xa0, wb0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_chunk(xa1, wb1);
            __builtin_amdgcn_sched_barrier(0);
            load_frags(cur, 3, xa1, wb1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_chunk(xa0, wb0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // explicit: do not rely on hipcc to drain the LDS-DMA in front of the barrier
            __syncthreads();   // hand-over: tile kt+1 landed in every wave, stage cur released
            load_frags(cur ^ 1, 0, xa0, wb0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_chunk(xa1, wb1);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        auto mfma_chunk = [&](const i32x8(&xa)[TM], const i32x8(&wb)[TN]) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wb[j], xa[i], acc[i][j], 0, 0, 0, 0, 0, 0);
        };
        i32x8 xa0[TM], wb0[TN], xa1[TM], wb1[TN];
        // Round 3: the DMA of tile kt+2 is issued right BEHIND the hand-over barrier of iteration kt (stage `cur` is free there:
        // every wave's last fragments of tile kt are in registers) and waited for at the NEXT hand-over - a whole k-tile of MFMA
        // time (2048 cycles per SIMD) to land.  Rounds 1-2 issued tile kt+1 at the top of iteration kt and waited for it half a
        // k-tile later: shorter than an L2 / Infinity-Cache round trip under load, so every k-tile stalled on its own DMA.
        // Round 4: the order reads | DMA pieces | MFMAs | wait is PINNED (sched_barrier), and for that every fragment read is an asm
        // statement waited for by hand.  Left to hipcc, a chunk's reads sank behind its MFMAs - right in front of the wait that needs
        // them, so the wait in front of the hand-over barrier and the first MFMA of the next chunk each sat out an LDS latency - and
        // its own counted waits degenerated to lgkmcnt(0) right behind freshly issued reads (it cannot see across the asm DMA).
        // The 320-row tile keeps hipcc's order: both fragment sets live at once (112 registers beside 160 of accumulators) do not
        // fit, which is why the reads were sunk there in the first place.
        constexpr bool PIN = TM < 5;
        // fragment of rows r: chunks (4 c + 2 h) ^ sw and (4 c + 2 h + 1) ^ sw of the 128-byte row - the chunk index is an XOR of
        // address bits 6 (c) and 4 (second half); every other term is a multiple of 128 bytes (aligned(128) stage memory)
        const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char *)smem);
        const unsigned fa0 = lds0 + a_row + (((2 * h) ^ sw) << 4), fb0 = lds0 + b_row + (((2 * h) ^ sw) << 4);
        auto load_frags = [&](int stage, int c, i32x8(&xa)[TM], i32x8(&wb)[TN]) {
            if constexpr (PIN) {
                const unsigned so = (unsigned)(stage * (ROWS * ROW8));
                const unsigned a_lo = (fa0 + so) ^ (unsigned)(c << 6), a_hi = a_lo ^ 16u, b_lo = (fb0 + so) ^ (unsigned)(c << 6), b_hi = b_lo ^ 16u;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    union { i32x4 v[2]; i32x8 f; } u;
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(u.v[0]) : "v"(a_lo), "n"(i * 32 * ROW8));
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(u.v[1]) : "v"(a_hi), "n"(i * 32 * ROW8));
                    xa[i] = u.f;
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    union { i32x4 v[2]; i32x8 f; } u;
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(u.v[0]) : "v"(b_lo), "n"(j * 32 * ROW8));
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(u.v[1]) : "v"(b_hi), "n"(j * 32 * ROW8));
                    wb[j] = u.f;
                }
            } else {
                const char *base = smem + stage * (ROWS * ROW8);
                const int o0 = ((4 * c + 2 * h) ^ sw) * 16, o1 = ((4 * c + 2 * h + 1) ^ sw) * 16;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const i32x4 lo = *reinterpret_cast<const i32x4 *>(base + a_row + i * 32 * ROW8 + o0);
                    const i32x4 hi = *reinterpret_cast<const i32x4 *>(base + a_row + i * 32 * ROW8 + o1);
                    xa[i] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const i32x4 lo = *reinterpret_cast<const i32x4 *>(base + b_row + j * 32 * ROW8 + o0);
                    const i32x4 hi = *reinterpret_cast<const i32x4 *>(base + b_row + j * 32 * ROW8 + o1);
                    wb[j] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
            }
        };
        // (PIN: every group below is fenced; else the fences and hand waits for the reads drop out and hipcc orders / waits as before)
        auto fence = [&]() { if constexpr (PIN) __builtin_amdgcn_sched_barrier(0); };
        auto reads_done = [&]() { if constexpr (PIN) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
        issue(0, 0);
        issue(1, (nk > 1 ? 1 : 0) * BKE);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");      // tile 0 landed; tile 1's NLD pieces stay in flight
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        load_frags(0, 0, xa0, wb0);
        reads_done();
        fence();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            const int k2 = (kt + 2 < nk ? kt + 2 : nk - 1) * BKE;
            load_frags(cur, 1, xa1, wb1);
            fence();
            mfma_chunk(xa0, wb0);
            fence();
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // tile kt+1 landed; own reads of stage cur retired
            __builtin_amdgcn_s_barrier();   // hand-over: tile kt+1 visible to every wave, stage cur released
            asm volatile("" ::: "memory");
            fence();
            if constexpr (PIN) {
                load_frags(cur ^ 1, 0, xa0, wb0);
                fence();
                issue(cur, k2);             // tile kt+2 -> stage cur (a clamped re-fetch on the last two iterations, never read)
            } else {
                issue(cur, k2);
                load_frags(cur ^ 1, 0, xa0, wb0);
            }
            mfma_chunk(xa1, wb1);
            fence();
            reads_done();
            fence();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped re-fetches must not outlive the LDS allocation
    }

    const bool cols_in = (n0 + BN <= p.N) && ((p.ldy & 3) == 0);
    const int mw = m0 + wm * TM * 32, nw = n0 + wn * TN * 32;
    // (the slab path stores eight bf16 columns per lane - EPI_BIAS, the q|k|v output -: ldy a multiple of 8 then)
    if (cols_in && (EPI != EPI_BIAS || (p.ldy & 7) == 0) && m0 + BM <= p.M && !p.direct_epi) {
        __syncthreads();     // every wave is out of the k-loop (and its DMA drained): the stage memory becomes slab buffers
        const float ab = p.d_act ? p.d_act[0] : p.ab_scale;
        const float oinv = (EPI == EPI_BIAS_GELU && p.d_out) ? 1.0f / p.d_out[0] : p.out_inv_scale;
        store_rows_via_lds<TM, TN, EPI, EPI == EPI_SCALE_RESID ? EPI_OUT_F32 : EPI == EPI_BIAS_GELU ? EPI_OUT_FP8 : EPI_OUT_BF16>(
            acc, smem + wave * EPI_WAVE_BYTES, p.Y, p.Y2, p.R, p.bias, p.lam, p.d_wrow, p.ldy, mw, nw, lane, ab, oinv);
    } else if (cols_in && m0 + BM <= p.M) store_q<TM, TN, EPI, 0>(p, acc, mw, nw, lane);
    else if (cols_in) store_q<TM, TN, EPI, 1>(p, acc, mw, nw, lane);
    else store_q<TM, TN, EPI, 2>(p, acc, mw, nw, lane);
}

template <int WM, int WN, int TM, int TN, int EPI, bool K64>
int launch_q(const GemmArgs8 &a, hipStream_t stream)
{
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int lds = 2 * (BM + BN) * ROW8;
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    auto kern = gemm_fp8_mfma<WM, WN, TM, TN, EPI, K64>;
    LDIT_DYN_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(64 * WM * WN), lds, stream, a);
    LDIT_HIP_CHECK(hipGetLastError());
    return LDIT_OK;
}

// ---- tail kernel: up to 64 rows (the ragged tail peeled off by launch_gemm_fp8, or a whole problem that small) -------------
// Same scheme as gemm_bf16_tail (gemm_bf16.hip): one wave per 32 x 32 output tile walks ALL of K in order with the tile
// kernel's own instruction - v_mfma_f32_32x32x64_f8f6f4, a lane's operand = 32 consecutive fp8 of its row at k = 64 s + 32 h -
// so a peeled row gets the bits a 256 x 256 tile would have given it (the round-2 split-K kernel summed eight K slices of
// v_mfma_f32_32x32x16_fp8_fp8 products: another order AND another instruction).  Fragments straight from global memory,
// two register sets of four 64-deep steps.
template <int EPI>
__global__ void __launch_bounds__(64) gemm_fp8_tail(const GemmArgs8 p)
{
    constexpr int D = 4;
    const int lane = threadIdx.x, c32 = lane & 31, h = lane >> 5;
    const int nct = (p.N + 31) / 32;
    const int n0 = (blockIdx.x % nct) * 32, m0 = (blockIdx.x / nct) * 32;
    const int ra = m0 + c32 < p.M ? m0 + c32 : p.M - 1, rw = n0 + c32 < p.N ? n0 + c32 : p.N - 1;
    const unsigned char *ap = p.A + (size_t)ra * p.lda + 32 * h, *wp = p.W + (size_t)rw * p.K + 32 * h;
    const int nsteps = p.K / 64;
    f32x16 acc[1][1];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[0][0][e] = 0.0f;
    i32x8 xa0[D], wb0[D], xa1[D], wb1[D];
    auto ld = [&](i32x8(&xa)[D], i32x8(&wb)[D], int s0) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int s = s0 + u < nsteps ? s0 + u : nsteps - 1;
            const i32x4 al = *reinterpret_cast<const i32x4 *>(ap + 64 * s), ah = *reinterpret_cast<const i32x4 *>(ap + 64 * s + 16);
            const i32x4 wl = *reinterpret_cast<const i32x4 *>(wp + 64 * s), wh = *reinterpret_cast<const i32x4 *>(wp + 64 * s + 16);
            xa[u] = i32x8{al[0], al[1], al[2], al[3], ah[0], ah[1], ah[2], ah[3]};
            wb[u] = i32x8{wl[0], wl[1], wl[2], wl[3], wh[0], wh[1], wh[2], wh[3]};
        }
    };
    auto mm = [&](const i32x8(&xa)[D], const i32x8(&wb)[D], int s0) {
#pragma unroll
        for (int u = 0; u < D; ++u)
            if (s0 + u < nsteps)
                acc[0][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wb[u], xa[u], acc[0][0], 0, 0, 0, 0, 0, 0);
    };
    ld(xa0, wb0, 0);
    for (int s0 = 0; s0 < nsteps; s0 += 2 * D) {
        ld(xa1, wb1, s0 + D);
        mm(xa0, wb0, s0);
        ld(xa0, wb0, s0 + 2 * D);
        mm(xa1, wb1, s0 + D);
    }
    if ((n0 + 32 <= p.N) && ((p.ldy & 3) == 0)) store_q<1, 1, EPI, 1>(p, acc, m0, n0, lane);
    else store_q<1, 1, EPI, 2>(p, acc, m0, n0, lane);
}

template <int EPI>
int launch_qtail(const GemmArgs8 &a, hipStream_t stream)
{
    const unsigned blocks = (unsigned)(((a.N + 31) / 32) * ((a.M + 31) / 32));
    hipLaunchKernelGGL(gemm_fp8_tail<EPI>, dim3(blocks), dim3(64), 0, stream, a);
    LDIT_HIP_CHECK(hipGetLastError());
    return LDIT_OK;
}

template <int EPI>
int launch_q_tiled(const GemmArgs8 &a, hipStream_t stream)
{
    // LDIT_GEMM_FP8_K16=1 selects the K = 16 MFMA, LDIT_GEMM_FP8_TILE=0..2 forces a tile (both for experiments / tests)
    const bool k16 = diag().fp8_k16, noskinny = diag().fp8_noskinny;
    if (a.M <= 64 && !noskinny && !k16) return launch_qtail<EPI>(a, stream);     // peeled tail / tiny batch (bit-identical to the K = 64 tiles)
    // Time model fitted to scripts/gemm_fp8_bench.py on ViT-B / ViT-L shapes, M = 3 k .. 25 k (profiles/README.md), in us:
    //   256 x 256 (one workgroup per CU):  strict rounds of 256 tiles, each  a[epi] + 11.3e-3 K
    //   128 x 128 (two per CU, they overlap each other's prologue / epilogue):  rounds of 256 tiles, each  r[epi] + 4.25e-3 K,
    //   but never less than one tile's own latency  5.3 + 8e-3 K.   (256 x 128 never won a shape: kept for experiments only.)
    const double a256[3] = {14.3, 16.8, 21.3}, r128[3] = {3.65, 3.85, 7.2};
    const long t256 = (long)((a.M + 255) / 256) * ((a.N + 255) / 256), t128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    const double c256 = (double)((t256 + 255) / 256) * (a256[EPI] + 11.3e-3 * a.K);
    double c128 = (double)((t128 + 255) / 256) * (r128[EPI] + 4.25e-3 * a.K);
    if (c128 < 5.3 + 8e-3 * a.K) c128 = 5.3 + 8e-3 * a.K;
    int pick = c256 <= c128 ? 0 : 2;
    // Round 2: 192- and 320-row variants of the 8-wave tile against round quantisation (see gemm_bf16.hip launch_h_tiled)
    double best = c256 <= c128 ? c256 : c128;
    const double per256 = a256[EPI] + 11.3e-3 * a.K;
    for (int bm : {192, 320}) {
        if (bm == 320 && EPI == EPI_SCALE_RESID) continue;      // its residual epilogue does not fit 256 VGPRs (spills)
        const long t = (long)((a.M + bm - 1) / bm) * ((a.N + 255) / 256);
        const double c = (double)((t + 255) / 256) * per256 * (bm / 256.0) * 1.03;
        if (c < best) { best = c; pick = bm == 192 ? 3 : 4; }
    }
    if (const int force = diag().fp8_tile; force >= 0 && force <= 4) pick = force;
    if (k16 && pick > 2) pick = 0;
    if (pick == 3) return launch_q<2, 4, 3, 2, EPI, true>(a, stream);      // 192 x 256
    if (pick == 4 && EPI == EPI_SCALE_RESID) pick = 0;
    if constexpr (EPI != EPI_SCALE_RESID)
        if (pick == 4) return launch_q<2, 4, 5, 2, EPI, true>(a, stream);  // 320 x 256
    if (k16) {
        if (pick == 0) return launch_q<2, 4, 4, 2, EPI, false>(a, stream);
        if (pick == 1) return launch_q<2, 2, 4, 2, EPI, false>(a, stream);
        return launch_q<2, 2, 2, 2, EPI, false>(a, stream);
    }
    if (pick == 0) return launch_q<2, 4, 4, 2, EPI, true>(a, stream);      // 256 x 256, 8 waves
    if (pick == 1) return launch_q<2, 2, 4, 2, EPI, true>(a, stream);      // 256 x 128, 4 waves
    return launch_q<2, 2, 2, 2, EPI, true>(a, stream);                     // 128 x 128, 4 waves
}

// dst[i] = fp8(src[i] * inv_scale), saturating; 4 elements per thread
__global__ void __launch_bounds__(256) quant_fp8(const float *__restrict__ src, unsigned char *__restrict__ dst, size_t n,
                                                 float inv_scale, const float *__restrict__ d_scale)
{
    if (d_scale) inv_scale = 1.0f / d_scale[0];
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(src + i);
        *reinterpret_cast<unsigned *>(dst + i) = pack_fp8x4(v[0] * inv_scale, v[1] * inv_scale, v[2] * inv_scale, v[3] * inv_scale);
    } else {
        for (size_t k = i; k < n; ++k) dst[k] = (unsigned char)pack_fp8x4(src[k] * inv_scale, 0.f, 0.f, 0.f);
    }
}

// *out = max(*out, max |src|): non-negative floats order like their bit patterns, so one integer atomicMax per block
__global__ void __launch_bounds__(256) amax_f32(const float *__restrict__ src, size_t n, unsigned *__restrict__ out)
{
    float m = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = fmaxf(m, fabsf(src[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
        atomicMax(out, __float_as_uint(m));
    }
}


// One wave per weight row: scale[n] = max(|W[n, :]|) / 448 (never 0), codes[n, :] = fp8(W[n, :] / scale[n]).
// Per-output-channel scales cost nothing in the GEMM (one more factor on the accumulator in the epilogue) and keep
// a few large rows from flattening all the others.
__global__ void __launch_bounds__(256) quant_rows_fp8(const float *__restrict__ W, unsigned char *__restrict__ dst,
                                                      float *__restrict__ scales, int N, int K, float mul)
{
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const f32x4 *w4 = reinterpret_cast<const f32x4 *>(W + (size_t)row * K);
    float m = 0.0f;
    for (int i = lane; i < K / 4; i += 64) {
        const f32x4 v = w4[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    // (mul: pack-time factor on the row - scale * log2 e on W_q for the pre-scaled attention - carried by the row's scale: the
    // codes are those of the unscaled row)
    const float sc = fmaxf(m, 1e-30f) * (1.0f / 448.0f), inv = 1.0f / sc;
    if (lane == 0) scales[row] = sc * mul;
    unsigned *d4 = reinterpret_cast<unsigned *>(dst + (size_t)row * K);
    for (int i = lane; i < K / 4; i += 64) {
        const f32x4 v = w4[i];
        d4[i] = pack_fp8x4(v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv);
    }
}

__global__ void amax_to_scale(float *p, int n)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i < n) p[i] = fmaxf(p[i], 1e-30f) * (1.0f / 448.0f);
}

}  // namespace

static int launch_gemm_fp8_one(const void *A, int lda, const void *W, const float *bias, void *Y, int ldy, int M, int N, int K,
                               int epi, const float *lam, const float *R, float *Y2, float ab_scale, float out_inv_scale,
                               const float *d_act, const float *d_wrow, const float *d_out, hipStream_t stream);

// A ragged tail of up to 64 rows past a multiple of the 256-row tile is peeled off into a second, tiny launch (gemm_fp8_tail;
// see launch_gemm_bf16_ex): M = 16 x 1025 = 64 x 256 + 16 would otherwise cost a whole extra round of workgroups.
int launch_gemm_fp8(const void *A, int lda, const void *W, const float *bias, void *Y, int ldy, int M, int N, int K, int epi,
                    const float *lam, const float *R, float *Y2, float ab_scale, float out_inv_scale, const float *d_act,
                    const float *d_wrow, const float *d_out, hipStream_t stream)
{
    const int rem = M % 256;
    const long nbn = (N + 255) / 256, full = ((long)M / 256 + 1) * nbn, mainp = ((long)M / 256) * nbn;
    if (rem != 0 && rem <= 64 && M > 256 && (full + 255) / 256 > (mainp + 255) / 256) {   // peel only when it saves a round
        const int main_rows = M - rem;
        const size_t out_elt = epi == EPI_SCALE_RESID ? 4 : epi == EPI_BIAS_GELU ? 1 : 2;
        int rc = launch_gemm_fp8_one(A, lda, W, bias, Y, ldy, main_rows, N, K, epi, lam, R, Y2, ab_scale, out_inv_scale, d_act, d_wrow, d_out, stream);
        if (rc != LDIT_OK) return rc;
        const char *At = static_cast<const char *>(A) + (size_t)main_rows * lda;
        char *Yt = static_cast<char *>(Y) + (size_t)main_rows * ldy * out_elt;
        return launch_gemm_fp8_one(At, lda, W, bias, Yt, ldy, rem, N, K, epi, lam, R ? R + (size_t)main_rows * ldy : nullptr,
                                   Y2 ? Y2 + (size_t)main_rows * ldy : nullptr, ab_scale, out_inv_scale, d_act, d_wrow, d_out, stream);
    }
    return launch_gemm_fp8_one(A, lda, W, bias, Y, ldy, M, N, K, epi, lam, R, Y2, ab_scale, out_inv_scale, d_act, d_wrow, d_out, stream);
}

static int launch_gemm_fp8_one(const void *A, int lda, const void *W, const float *bias, void *Y, int ldy, int M, int N, int K,
                               int epi, const float *lam, const float *R, float *Y2, float ab_scale, float out_inv_scale,
                               const float *d_act, const float *d_wrow, const float *d_out, hipStream_t stream)
{
    if (M <= 0 || N <= 0 || K <= 0) return fail(LDIT_EINVAL, "gemm_fp8: empty problem");
    if (K % BKE) return fail(LDIT_EUNSUPPORTED, "gemm_fp8: K=%d must be a multiple of %d", K, BKE);
    if (!A || !W || !Y) return fail(LDIT_EINVAL, "gemm_fp8: null operand");
    if (!aligned16(A) || !aligned16(W) || (lda & 15)) return fail(LDIT_EINVAL, "gemm_fp8: operands must be 16-byte aligned");
    GemmArgs8 a{};
    a.A = static_cast<const unsigned char *>(A); a.W = static_cast<const unsigned char *>(W); a.Y = Y; a.Y2 = Y2;
    a.bias = bias; a.lam = lam; a.R = R; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldy = ldy;
    a.ab_scale = ab_scale; a.out_inv_scale = out_inv_scale; a.d_act = d_act; a.d_wrow = d_wrow; a.d_out = d_out;
    a.direct_epi = diag().direct_epi ? 1 : 0;
    switch (epi) {
        case EPI_BIAS: return launch_q_tiled<EPI_BIAS>(a, stream);
        case EPI_BIAS_GELU: return launch_q_tiled<EPI_BIAS_GELU>(a, stream);
        case EPI_SCALE_RESID:
            if (!lam || !R) return fail(LDIT_EINVAL, "gemm_fp8: scale+residual epilogue needs lam and R");
            return launch_q_tiled<EPI_SCALE_RESID>(a, stream);
        default: return fail(LDIT_EINVAL, "gemm_fp8: unknown epilogue %d", epi);
    }
}

int launch_quant_fp8(const float *src, void *dst, size_t n, float inv_scale, const float *d_scale, hipStream_t stream)
{
    if (n == 0) return LDIT_OK;
    hipLaunchKernelGGL(quant_fp8, dim3((unsigned)((n / 4 + 255) / 256 + 1)), dim3(256), 0, stream, src,
                       static_cast<unsigned char *>(dst), n, inv_scale, d_scale);
    LDIT_HIP_CHECK(hipGetLastError());
    return LDIT_OK;
}

int launch_quant_rows_fp8(const float *W, void *dst, float *scales, int N, int K, hipStream_t stream, float mul)
{
    if (N <= 0 || K <= 0 || (K & 3)) return fail(LDIT_EINVAL, "quant_rows_fp8: bad shape %d x %d", N, K);
    if (!W || !dst || !scales || !aligned16(W)) return fail(LDIT_EINVAL, "quant_rows_fp8: null or misaligned operand");
    hipLaunchKernelGGL(quant_rows_fp8, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, stream, W, static_cast<unsigned char *>(dst),
                       scales, N, K, mul);
    LDIT_HIP_CHECK(hipGetLastError());
    return LDIT_OK;
}

int launch_amax_to_scale(float *p, int n, hipStream_t stream)
{
    if (n <= 0) return LDIT_OK;
    hipLaunchKernelGGL(amax_to_scale, dim3((n + 63) / 64), dim3(64), 0, stream, p, n);
    LDIT_HIP_CHECK(hipGetLastError());
    return LDIT_OK;
}

int launch_amax_f32(const float *src, size_t n, float *out, bool accumulate, hipStream_t stream)
{
    if (!accumulate) LDIT_HIP_CHECK(hipMemsetAsync(out, 0, sizeof(float), stream));
    if (n == 0) return LDIT_OK;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(amax_f32, dim3(blocks), dim3(256), 0, stream, src, n, reinterpret_cast<unsigned *>(out));
    LDIT_HIP_CHECK(hipGetLastError());
    return LDIT_OK;
}

}  // namespace ldit
