// fp32 MFMA GEMM for SERVING-SIZE batches (M = B*197 of a few hundred rows):  Y[M,N] = epi( A[M,K] . W[N,K]^T ).
//
// At one to four images the big tilings (gemm_panel_f32.hip, gemm_f32.hip) leave most of the chip idle: a 64x64 tile of
// M = 197 is 48-192 workgroups, each a serial chain of K/32 k-tiles whose every step waits a full HBM round trip for the
// weights (two LDS stages) - 23 us for K = 768, 85 us for K = 3072, where the FLOPs are worth 2-7 us.  This kernel trades
// L2 traffic for parallelism and latency hiding:
//   * 32 x 32 output tile per workgroup, four waves, each ONE 16x16 accumulator on v_mfma_f32_16x16x4_f32: M = 197 is
//     7 x N/32 = 168-672 workgroups, and the dependent chain per output is K/4 MFMAs of 32 cycles (a 32x32x2 chain is
//     twice as long).
//   * the same k order as every other fp32 GEMM here - within an 8-deep chunk k0,k4,k1,k5 | k2,k6,k3,k7, which is what
//     two 32x32x2 steps with lane half h on k = 4h + s compute - so the results are BIT-IDENTICAL to the big tilings and
//     the batch-invariance guarantee (a row's value does not depend on how many rows ride along) is kept.
//   * a four-slot LDS ring filled by LDS-DMA, a slot = two k-tiles (2 x 8 KB: 32 A rows + 32 W rows of 128 B each): four
//     k-tiles of weights are in flight per workgroup (two workgroups per CU), so the HBM latency is paid once, not per
//     k-tile.  Counted vmcnt, one raw s_barrier per TWO k-tiles (a barrier per tile measured 25 % of the loop); the
//     fragments of tile kt+1 are read between tile kt's MFMAs; fragment reads are inline asm (the compiler cannot tell they
//     do not alias the DMA writes of later slots and would drain vmcnt to zero in front of each).
//   * fragment = ds_read_b128 of 16-B chunk 2c + (q & 1) of the XOR-swizzled 128-B row (lane = row r = lane & 15, quarter
//     q = lane >> 4): the 16 lanes of a quarter read 16 different rows, conflict-free (a ds_read2_b32 of just the two dwords
//     a lane needs costs twice the LDS cycles - a 32-lane half only reaches 16 banks - and LDS issue, not the MFMA chain,
//     then sets the pace).  The lane keeps dwords (q >> 1) and (q >> 1) + 2 - its k of MFMA 1 and MFMA 2 of chunk c - with
//     one select per MFMA operand.
//   * blockIdx -> tile: column tiles are dealt round-robin to the XCDs and the row tiles of one column tile are
//     consecutive on that XCD, so a weight row is fetched from HBM into ONE L2, once.
#include <cstdlib>
#include <type_traits>

#include "ldit_common.h"

namespace ldit {

namespace {

constexpr int BK = 32, ROW_BYTES = BK * 4, TB = 32;      // k-tile depth, LDS row, tile edge
constexpr int HALF = 2 * TB * ROW_BYTES;                 // 8 KB per k-tile: 32 A rows, then 32 W rows
constexpr int STAGE = 2 * HALF;                          // a stage = TWO k-tiles: one barrier per 64 of k
constexpr int NS = 4;                                    // ring depth: 64 KB, two workgroups per CU

typedef float f32x2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void glds16t(const float *gsrc, char *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

template <int EPI>
__global__ void __launch_bounds__(256) gemm_thin_f32(const GemmArgs p, const int nbm, const int nbn)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    const int tm = idx % nbm, tn = (idx / nbm) * 8 + xcd;
    if (tn >= nbn) return;                               // grid padding of the last column-tile group (block-uniform)
    const int m0 = tm * TB, n0 = tn * TB;
#ifdef LDIT_GEMM_STAMPS
    // diagnostic build only (never shipped in libldit_hip.so; scripts/thin_stamps.py): wall/cycle stamps of the block's phases
    const unsigned long long st_real0 = __builtin_amdgcn_s_memrealtime(), st_clk0 = __builtin_amdgcn_s_memtime();
    unsigned long long st_clk1 = 0, st_clk2 = 0;
#endif

    // ---- DMA: wave w moves A rows 8w..8w+7 (piece w) and W rows 8w..8w+7 (piece 4+w) of every stage ----------------
    unsigned src_a, src_w;
    {
        const int row = 8 * wave + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);     // logical 16-B chunk this lane fetches
        int gm = m0 + row, gn = n0 + row;
        gm = gm < p.M ? gm : p.M - 1;
        gn = gn < p.N ? gn : p.N - 1;
        src_a = (unsigned)gm * (unsigned)p.lda + c * 4;
        src_w = (unsigned)gn * (unsigned)p.K + c * 4;
    }
    const int nk = p.K / BK, nst = (nk + 1) / 2;         // k-tiles; stages of two (an odd tail's second half repeats the last tile, unused)
    auto issue = [&](int st) {                           // both k-tiles of stage st into slot st % NS: four 1-KB pieces per wave
        char *base = smem + (st % NS) * STAGE;
        const int sc = st < nst ? st : nst - 1;          // past the end: re-fetch the last stage (keeps the vmcnt count exact)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int kt = 2 * sc + t < nk ? 2 * sc + t : nk - 1;
            glds16t(p.A + (src_a + (unsigned)(kt * BK)), base + t * HALF + wave * 1024);
            glds16t(p.W + (src_w + (unsigned)(kt * BK)), base + t * HALF + (4 + wave) * 1024);
        }
    };

    // ---- fragment addresses inside a stage ------------------------------------------------------------------------------
    const int r16 = lane & 15, q = lane >> 4;
    unsigned fa[4], fb[4];
    {
        const int ra = wm * 16 + r16, rb = wn * 16 + r16;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            fa[c] = ra * ROW_BYTES + (((2 * c + (q & 1)) ^ ((ra >> 1) & 7)) * 16);
            fb[c] = (TB + rb) * ROW_BYTES + (((2 * c + (q & 1)) ^ ((rb >> 1) & 7)) * 16);
        }
    }
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) const char *)smem);
    const bool live = (m0 + wm * 16 < p.M) && (n0 + wn * 16 < p.N);          // wave-uniform: a 16x16 tile wholly outside idles

    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    // Fragments of k-tile kt live in half kt & 1 of slot (kt / 2) % NS.  The stage loop is unrolled over the ring, so a
    // tile's slot and half are compile-time constants and every fragment read is ds_read_b128 with an immediate offset from
    // eight lane-constant addresses (no address arithmetic inside the chain).
    struct Raw { f32x4 a[4], b[4]; };
    struct Op2 { float a, b; };                        // the operands of one MFMA
    unsigned fav[4], fbv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { fav[c] = lds0 + fa[c]; fbv[c] = lds0 + fb[c]; }
    const unsigned oddm = (q >> 1) ? 0xffffffffu : 0u;
    auto pick = [&](float even_v, float odd_v) {      // one instruction, and no select for hipcc to re-index as v[odd]
        return __uint_as_float((__float_as_uint(odd_v) & oddm) | (__float_as_uint(even_v) & ~oddm));
    };
    // One k-tile = a chain of eight dependent MFMAs (33 cycles each, nothing else to overlap them with), so everything
    // else is dealt out into their shadows, three instructions behind each MFMA m:
    //   * read #m of the NEXT tile's eight fragments (chunk m/2, A for even m, B for odd);
    //   * the two selects that make MFMA m+2's operands: a lane keeps dwords (0,2) of a chunk, or (1,3) in the upper two
    //     quarters.  They are plain C++ (hipcc emits v_cndmask) and sit two MFMAs ahead of their consumer.  NOT inline asm:
    //     on gfx950 a VALU result is not yet visible to an MFMA issued one wait state later; hipcc separates its own VALU
    //     from a dependent MFMA with s_nop 1, but an asm-written v_cndmask is invisible to its hazard recognizer - the
    //     sequence v_cndmask, v_cndmask, one instruction, v_mfma made the MFMA read the OLD value of the register written
    //     second (measured: products a[k=8] b[k=2]).  A select is a pure value that no sched_barrier holds in place: each
    //     pair passes through an empty volatile asm, which pins it without hiding the producer.
    //   * behind every even m a COUNTED lgkmcnt(5): LDS returns in order, and of the reads in flight only the five youngest
    //     may still be pending when chunk (m+2)/2 of this tile (or chunk 0 of the next, for m = 6) is selected from.
    // With the selects and four v_add per chunk next to the MFMAs the same loop ran 630 cycles per k-tile against the
    // chain's 266: in a dependent chain every instruction between two MFMAs is on the critical path.
    auto tile = [&](auto tc, const Raw &cur, Raw &nxt, Op2(&op)[2]) {
        constexpr int NT = (decltype(tc)::value + 1) & (2 * NS - 1);              // ring position of the next tile
        constexpr int OFF = (NT >> 1) * STAGE + (NT & 1) * HALF;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(op[m & 1].a, op[m & 1].b, acc, 0, 0, 0);
            if (m & 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(nxt.b[m >> 1]) : "v"(fbv[m >> 1]), "n"(OFF));
            else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(nxt.a[m >> 1]) : "v"(fav[m >> 1]), "n"(OFF));
            if (!(m & 1)) asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory");
            const int m2 = m + 2, c2 = (m2 & 7) >> 1, e = (m2 & 1) * 2;
            const Raw &src = m2 < 8 ? cur : nxt;
            Op2 n{pick(src.a[c2][e], src.a[c2][e + 1]), pick(src.b[c2][e], src.b[c2][e + 1])};
            asm volatile("" : "+v"(n.a), "+v"(n.b));
            op[m & 1] = n;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // Ring protocol, per stage j (slot j % NS): in front of the wait the wave has stages j+1 .. j+NS-2 in flight, four loads
    // each; stage j+1 is the oldest, so it has landed once at most 4(NS-3) are pending, and behind the barrier every wave's
    // pieces of it have - which the fragment reads of its first k-tile (during stage j's second) need.  Behind the same
    // barrier all waves are done reading stage j-1, which is refilled with stage j+NS-1.  Past the end the last stage is
    // re-fetched into the idle slot so the count stays exact (nobody reads it).
    Raw r0, r1;
    Op2 op[2];
    auto stage = [&](auto sc, int j) {
        constexpr int S = decltype(sc)::value;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NS - 3)) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue(j + NS - 1);
        if (live) {
            tile(std::integral_constant<int, 2 * S>{}, r0, r1, op);
            if (2 * j + 1 < nk) tile(std::integral_constant<int, 2 * S + 1>{}, r1, r0, op);
        }
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(s);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NS - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (live) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            asm volatile("ds_read_b128 %0, %1" : "=v"(r0.a[c]) : "v"(fav[c]));
            asm volatile("ds_read_b128 %0, %1" : "=v"(r0.b[c]) : "v"(fbv[c]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            op[i] = Op2{pick(r0.a[0][2 * i], r0.a[0][2 * i + 1]), pick(r0.b[0][2 * i], r0.b[0][2 * i + 1])};
            asm volatile("" : "+v"(op[i].a), "+v"(op[i].b));
        }
    }
#ifdef LDIT_GEMM_STAMPS
    st_clk1 = __builtin_amdgcn_s_memtime();
#endif
    static_assert(NS == 4, "the stage loop below is unrolled over a four-slot ring");
    for (int j0 = 0; j0 < nst; j0 += NS) {
        stage(std::integral_constant<int, 0>{}, j0);
        if (j0 + 1 < nst) stage(std::integral_constant<int, 1>{}, j0 + 1);
        if (j0 + 2 < nst) stage(std::integral_constant<int, 2>{}, j0 + 2);
        if (j0 + 3 < nst) stage(std::integral_constant<int, 3>{}, j0 + 3);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the fragment reads issued for a tile past the end: retire them
    __builtin_amdgcn_sched_barrier(0);
#ifdef LDIT_GEMM_STAMPS
    st_clk2 = __builtin_amdgcn_s_memtime();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the re-fetched tail tiles: no wave leaves with DMA in flight
#ifdef LDIT_GEMM_STAMPS
    if (p.stamps && lane == 0 && wave) {      // SIMD id of waves 1-3, four bits each, into word 5
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        atomicOr(p.stamps + (size_t)blockIdx.x * 8 + 5, (unsigned long long)(((hw >> 4) & 3) | 4) << (4 * wave));
    }
    if (p.stamps && tid == 0) {
        const unsigned long long st_clk3 = __builtin_amdgcn_s_memtime(), st_real1 = __builtin_amdgcn_s_memrealtime();
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long *o = p.stamps + (size_t)blockIdx.x * 8;
        o[0] = st_real0; o[1] = st_real1; o[2] = st_clk1 - st_clk0; o[3] = st_clk2 - st_clk1;
        o[4] = st_clk3 - st_clk2; o[7] = 0;
        atomicOr(o + 5, (unsigned long long)(((hwid >> 4) & 3) | 4)); o[6] = ((unsigned long long)xcc << 32) | hwid;
    }
#endif
    if (!live) return;

    // ---- epilogue: lane (column n = lane & 15, row group q) holds rows 4q..4q+3 of its column -------------------------------
    const int n = n0 + wn * 16 + r16, mb = m0 + wm * 16 + 4 * q;
    if (n >= p.N) return;
    const float bias = p.bias ? p.bias[n] : 0.0f;
    float lam = 0.0f;
    if (EPI == EPI_SCALE_RESID) lam = p.lam[n];
    const unsigned ldy = (unsigned)p.ldy;
    float resid[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (EPI == EPI_SCALE_RESID) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (mb + e < p.M) resid[e] = p.R[(unsigned)(mb + e) * ldy + (unsigned)n];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int m = mb + e;
        if (m >= p.M) continue;
        float v = acc[e] + bias;
        if (EPI == EPI_BIAS_GELU) v = gelu_erf(v);
        if (EPI == EPI_SCALE_RESID) v = resid[e] + lam * v;
        const unsigned o = (unsigned)m * ldy + (unsigned)n;
        p.Y[o] = v;
        if (p.Y2) p.Y2[o] = v;
    }
}

template <int EPI>
int launch_thin(const GemmArgs &a, hipStream_t stream)
{
    constexpr int lds = NS * STAGE;
    const int nbm = (a.M + TB - 1) / TB, nbn = (a.N + TB - 1) / TB;
    auto kern = gemm_thin_f32<EPI>;
    LDIT_DYN_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)(nbm * ((nbn + 7) / 8) * 8)), dim3(256), lds, stream, a, nbm, nbn);
    LDIT_HIP_CHECK(hipGetLastError());
    return LDIT_OK;
}

}  // namespace

// launch_gemm prefers this kernel while the 64 x 64 tiling would be at most this many workgroups (under one round of the
// chip with room to spare): measured on cold weights, M = 197..788 x N = 768 and M = 197 x N = 2304/3072 run 1.2-1.9x faster
// here, M = 394 x N = 2304 (252 tiles) and beyond run faster on the big tilings.  LDIT_GEMM_THIN_TILES overrides; 0
// switches the kernel off.
bool gemm_thin_prefers(int M, int N)
{
    const long limit = diag().thin_tiles;
    return (long)((M + 63) / 64) * ((N + 63) / 64) <= limit;
}

// Row-major A, bias / bias+GELU / LayerScale+residual epilogues; operands were validated by launch_gemm.
int launch_gemm_thin(const GemmArgs &a, int epi, hipStream_t stream)
{
    if ((long)a.M * a.lda >= (1L << 31) || (long)a.N * a.K >= (1L << 31) || (long)a.M * a.ldy >= (1L << 31))
        return fail(LDIT_EUNSUPPORTED, "gemm_thin: operand too large for 32-bit element offsets");
    switch (epi) {
        case EPI_BIAS: return launch_thin<EPI_BIAS>(a, stream);
        case EPI_BIAS_GELU: return launch_thin<EPI_BIAS_GELU>(a, stream);
        case EPI_SCALE_RESID: return launch_thin<EPI_SCALE_RESID>(a, stream);
        default: return fail(LDIT_EINVAL, "gemm_thin: unknown epilogue %d", epi);
    }
}

}  // namespace ldit
