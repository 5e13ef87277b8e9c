// HBM-bound kernels of the train step (BASELINE configs[2]): everything of the backward that is not a
// matmul - LayerScale / LayerNorm backward with their column reductions, the transposed bf16 copies the K-contiguous
// MFMA GEMM needs for wgrad, the embedding backward's reductions, and the fused AdamW update.
//
// Column reductions (bias / gamma / beta / lambda gradients: sums over all B*N rows) are two-stage and atomic-free:
// every workgroup writes the partial sums of its rows to part[block][column], `reduce_jobs` then adds the partials of up
// to 16 vectors in one launch, in a fixed order - gradients are bit-reproducible run to run.
#include "ldit_common.h"
#include "image_blend.h"

namespace ldit {

namespace {

typedef __bf16 bf16_t;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- 64 x 64 tile transposes ----------------------------------------------------------------------------------------
// dst[n][m] = bf16(src[row(m)][n]) for m < M (zero for M <= m < Mp: the pad columns are the GEMM's K padding and must be
// finite zeros), n < N.  src row stride lds_, dst row stride Mp (multiple of 64).  SRC_F32: fp32 source (else bf16).
// `skip`: tokens per image when the source is a [B, 1 + P, C] token tensor whose CLS rows are skipped (m = b P + i ->
// source row b (P + 1) + 1 + i; 0 = plain rows).  `colsum` (optional): part[blockIdx.y][n] = sum over the tile's rows.
// `rowmajor` (optional, fp32 sources): bf16 copy of the source in its own layout - weight packing reads a matrix once and
// writes both the forward's operand copy and the dgrad's transposed copy.
template <bool SRC_F32>
__global__ void __launch_bounds__(256) transpose_tile(const void *__restrict__ src, bf16_t *__restrict__ dst, int M, int N,
                                                      int lds_, int Mp, int skip, float *__restrict__ colsum,
                                                      bf16_t *__restrict__ rowmajor)
{
    __shared__ float tile[64][65];
    const int n0 = blockIdx.x * 64, m0 = blockIdx.y * 64, tid = threadIdx.x;
    // load: thread -> (row tid>>2 [+0], 16 consecutive columns (tid&3)*16)
    {
        const int r = tid >> 2, cb = (tid & 3) * 16, m = m0 + r;
        float v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = 0.0f;
        if (m < M) {
            const size_t srow = skip ? (size_t)(m / skip) * (skip + 1) + 1 + (m % skip) : (size_t)m;
            if (SRC_F32) {
                const float *p = static_cast<const float *>(src) + srow * lds_ + n0 + cb;
#pragma unroll
                for (int e = 0; e < 16; ++e) if (n0 + cb + e < N) v[e] = p[e];
                if (rowmajor) {
                    bf16_t *r = rowmajor + srow * lds_ + n0 + cb;
#pragma unroll
                    for (int e = 0; e < 16; ++e) if (n0 + cb + e < N) r[e] = (bf16_t)v[e];
                }
            } else {
                const bf16_t *p = static_cast<const bf16_t *>(src) + srow * lds_ + n0 + cb;
                if (n0 + cb + 16 <= N && ((lds_ | n0) & 7) == 0) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8 *>(p), b = *reinterpret_cast<const bf16x8 *>(p + 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { v[e] = (float)a[e]; v[8 + e] = (float)b[e]; }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) if (n0 + cb + e < N) v[e] = (float)p[e];
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) tile[r][cb + e] = v[e];
    }
    __syncthreads();
    // store: thread -> (column tid>>2, 16 consecutive rows (tid&3)*16): 32-byte runs along m
    {
        const int cn = tid >> 2, rb = (tid & 3) * 16, n = n0 + cn;
        if (n < N) {
            bf16x8 a, b;
#pragma unroll
            for (int e = 0; e < 8; ++e) { a[e] = (bf16_t)tile[rb + e][cn]; b[e] = (bf16_t)tile[rb + 8 + e][cn]; }
            bf16_t *d = dst + (size_t)n * Mp + m0 + rb;
            *reinterpret_cast<bf16x8 *>(d) = a;
            *reinterpret_cast<bf16x8 *>(d + 8) = b;
        }
    }
    if (colsum && tid < 64 && n0 + tid < N) {
        float s = 0.0f;
#pragma unroll 8
        for (int r = 0; r < 64; ++r) s += tile[r][tid];
        colsum[(size_t)blockIdx.y * N + n0 + tid] = s;
    }
}

// part[blockIdx.y][n] = sum over the 64 rows of the tile of src[m][n] (bf16 source, fp32 sums): first stage of a bias gradient.
// Thread = 8 consecutive columns (one 16-byte load per row) x 16 rows; a workgroup covers 64 rows x 512 columns.
__global__ void __launch_bounds__(256) colsum_tile(const bf16_t *__restrict__ src, int M, int N, int ld, float *__restrict__ part)
{
    __shared__ float red[4][64][9];
    const int cg = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int n = blockIdx.x * 512 + cg * 8, r0 = blockIdx.y * 64 + rg * 16;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (n < N)
#pragma unroll 4
        for (int r = 0; r < 16; ++r)
            if (r0 + r < M) {
                const bf16x8 v = *reinterpret_cast<const bf16x8 *>(src + (size_t)(r0 + r) * ld + n);
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
            }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rg][cg][e] = s[e];
    __syncthreads();
    // 512 columns, 256 threads: two columns each, fixed order over the four row groups
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int c = threadIdx.x * 2 + t, col = blockIdx.x * 512 + c;
        if (col < N)
            part[(size_t)blockIdx.y * N + col] = (red[0][c >> 3][c & 7] + red[1][c >> 3][c & 7]) + (red[2][c >> 3][c & 7] + red[3][c >> 3][c & 7]);
    }
}

// dst[m][n] = bf16(src[row(m)][n]), row(m) skipping the CLS rows of a [B, 1 + P, C] token tensor when skip = P (else plain)
__global__ void __launch_bounds__(256) rows_to_bf16(const float *__restrict__ src, bf16_t *__restrict__ dst, int M, int N4, int skip)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)M * N4) return;
    const int m = (int)(i / N4), c4 = (int)(i - (size_t)m * N4);
    const size_t srow = skip ? (size_t)(m / skip) * (skip + 1) + 1 + (m % skip) : (size_t)m;
    const f32x4 v = reinterpret_cast<const f32x4 *>(src + srow * (size_t)N4 * 4)[c4];
    const bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    reinterpret_cast<bf16x4 *>(dst + (size_t)m * N4 * 4)[c4] = o;
}

// ---- LayerScale + residual backward (TF:432-434, 440-442 differentiated) ---------------------------------------------
// h_out = h_in + (lam (.) z) rs[row]  =>  dz = dh (.) lam rs (bf16, row-major AND transposed [C][Mp] for the wgrad),
// partial sums  dlam_part[blk][c] = sum_rows dh z rs ,  db_part[blk][c] = sum_rows dz  (db = gradient of the bias inside z).
// dh itself passes through unchanged to the residual branch.  64 x 64 tiles, grid (C/64, ceil(M/64)).
__global__ void __launch_bounds__(256) resid_bwd_tile(const float *__restrict__ dh, const bf16_t *__restrict__ z,
                                                      const float *__restrict__ lam, const float *__restrict__ rowscale,
                                                      bf16_t *__restrict__ dz, bf16_t *__restrict__ dzT, int M, int C, int Mp,
                                                      float *__restrict__ dlam_part, float *__restrict__ db_part)
{
    __shared__ float tile[64][65];      // dz (fp32) for the transposed store and the column sums
    __shared__ float prod[64][65];      // dh z rs
    const int n0 = blockIdx.x * 64, m0 = blockIdx.y * 64, tid = threadIdx.x;
    {
        const int r = tid >> 2, cb = (tid & 3) * 16, m = m0 + r;
        float a[16], q[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) { a[e] = 0.0f; q[e] = 0.0f; }
        if (m < M) {
            const float rs = rowscale ? rowscale[m] : 1.0f;
            const float *pd = dh + (size_t)m * C + n0 + cb;
            const bf16_t *pz = z + (size_t)m * C + n0 + cb;
            bf16x8 o0, o1;
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const f32x4 d4 = *reinterpret_cast<const f32x4 *>(pd + 4 * e4);
                const f32x4 l4 = *reinterpret_cast<const f32x4 *>(lam + n0 + cb + 4 * e4);
                const bf16x4 z4 = *reinterpret_cast<const bf16x4 *>(pz + 4 * e4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = d4[e] * rs;
                    a[4 * e4 + e] = g * l4[e];
                    q[4 * e4 + e] = g * (float)z4[e];
                    if (e4 < 2) o0[4 * e4 + e] = (bf16_t)a[4 * e4 + e];
                    else o1[4 * (e4 - 2) + e] = (bf16_t)a[4 * e4 + e];
                }
            }
            bf16_t *po = dz + (size_t)m * C + n0 + cb;
            *reinterpret_cast<bf16x8 *>(po) = o0;
            *reinterpret_cast<bf16x8 *>(po + 8) = o1;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) { tile[r][cb + e] = a[e]; prod[r][cb + e] = q[e]; }
    }
    __syncthreads();
    {
        const int cn = tid >> 2, rb = (tid & 3) * 16;
        bf16x8 a, b;
#pragma unroll
        for (int e = 0; e < 8; ++e) { a[e] = (bf16_t)tile[rb + e][cn]; b[e] = (bf16_t)tile[rb + 8 + e][cn]; }
        if (dzT) {
            bf16_t *d = dzT + (size_t)(n0 + cn) * Mp + m0 + rb;
            *reinterpret_cast<bf16x8 *>(d) = a;
            *reinterpret_cast<bf16x8 *>(d + 8) = b;
        }
    }
    if (tid < 128) {
        const int cn = tid & 63;
        float s = 0.0f;
        if (tid < 64) {
#pragma unroll 8
            for (int r = 0; r < 64; ++r) s += tile[r][cn];
            db_part[(size_t)blockIdx.y * C + n0 + cn] = s;
        } else {
#pragma unroll 8
            for (int r = 0; r < 64; ++r) s += prod[r][cn];
            dlam_part[(size_t)blockIdx.y * C + n0 + cn] = s;
        }
    }
}

// ---- LayerNorm backward (nn.LayerNorm, eps inside the rsqrt) -----------------------------------------------------------
// y = xhat g + b, xhat = (x - mu) rstd.  With gy = dy (.) g:  dx = rstd (gy - mean(gy) - xhat mean(gy xhat)).
// dh[row] += dx (the residual stream's gradient accumulates the branch's input gradient);  partial sums
// dg_part[blk][c] = sum dy xhat,  db_part[blk][c] = sum dy.  One wave per row (row in registers, statistics recomputed
// from the saved LN input exactly as the forward computes them), a workgroup of 4 waves walks rows blk*4+w, +4*grid, ...
// NWV waves per workgroup: 512 workgroups x 8 waves fill the chip at four waves per SIMD (C <= 1024); rows of up to 4096
// floats keep 4 waves (498 registers, 128 KB of reduction buffer)
// RES (round 4): the LayerScale + residual backward of the branch BELOW this LayerNorm (resid_bwd_tile's arithmetic: dz = dh' lam rs,
// partial sums of dz and of dh' z rs) applied to the row while its new residual gradient dh' is still in registers - the
// attention branch's resid_bwd behind layernorm_after's backward.  Saves a launch and one fp32 read of dh per layer; the partial
// sums come out per workgroup (layernorm_bwd_blocks rows) like dgamma / dbeta, through the same LDS buffer in a second round.
struct ResidFused {
    const bf16_t *z;          // [rows, C] pre-LayerScale branch output saved by the forward
    const float *lam;         // [C]
    const float *rowscale;    // [rows] stochastic-depth factors or null
    bf16_t *dz;               // [rows, C] out
    float *dlam_part, *dzb_part;   // [blocks, C] out
    const float *add;         // [rows, C] or null: one more gradient arriving at this hidden state (a tap's), added to dh in the same pass (RES or not)
};

template <int VPL, int NWV, bool RES>
__global__ void __launch_bounds__(64 * NWV) layernorm_bwd_rows(const float *__restrict__ dy, const float *__restrict__ x,
                                                          const float *__restrict__ g, float *__restrict__ dh, int64_t rows,
                                                          int C, float eps, float *__restrict__ dg_part,
                                                          float *__restrict__ db_part, const ResidFused rf)
{
    extern __shared__ __attribute__((aligned(16))) char smem_ln[];
    float *red = reinterpret_cast<float *>(smem_ln);          // [2][NWV][C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = C >> 2;
    const float inv_c = 1.0f / (float)C;
    // RES: gamma and lambda are re-read per row (3 KB vectors, L1 hits) instead of living in 2 VPL register quads: with them the
    // kernel needs 160 VGPRs - three waves per SIMD, i.e. ONE eight-wave workgroup per CU instead of two, and the rows in flight
    // that hide the memory round trip halve (measured: 47 us against 28 + 16 for the two separate kernels)
    f32x4 gg[VPL], sg[VPL], sb[VPL];
    f32x4 sa[RES ? VPL : 1], sq[RES ? VPL : 1];
#pragma unroll
    for (int u = 0; u < VPL; ++u) {
        const int idx = lane + 64 * u;
        if (!RES) gg[u] = idx < nvec ? reinterpret_cast<const f32x4 *>(g)[idx] : f32x4{0.f, 0.f, 0.f, 0.f};
        sg[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        sb[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (RES) {
            sa[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            sq[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    for (int64_t row = (int64_t)blockIdx.x * NWV + wave; row < rows; row += (int64_t)gridDim.x * NWV) {
        const f32x4 *x4 = reinterpret_cast<const f32x4 *>(x + row * C);
        const f32x4 *d4 = reinterpret_cast<const f32x4 *>(dy + row * C);
        f32x4 *h4 = reinterpret_cast<f32x4 *>(dh + row * C);
        f32x4 v[VPL], d[VPL], acc[VPL];             // dh is fetched with x and dy: one memory round trip per row, not two
        bf16x4 zr[RES ? VPL : 1];
        const f32x4 *gp = reinterpret_cast<const f32x4 *>(g), *lp = reinterpret_cast<const f32x4 *>(rf.lam);
        if (RES) asm volatile("" : "+s"(gp), "+s"(lp));      // opaque per row: keeps the loads in the loop (see above)
        float s = 0.0f;
#pragma unroll
        for (int u = 0; u < VPL; ++u) {
            const int idx = lane + 64 * u;
            if (idx < nvec) {
                v[u] = x4[idx];
                d[u] = d4[idx];
                acc[u] = h4[idx];
                if (rf.add) {
                    const f32x4 t = reinterpret_cast<const f32x4 *>(rf.add + row * C)[idx];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[u][e] += t[e];
                }
                if (RES) {
                    zr[u] = reinterpret_cast<const bf16x4 *>(rf.z + row * C)[idx];
                    gg[u] = gp[idx];
                }
                s += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
            } else {
                v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                d[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        const float mu = wave_sum(s) * inv_c;
        float q = 0.0f;
#pragma unroll
        for (int u = 0; u < VPL; ++u) {
            const int idx = lane + 64 * u;
            if (idx < nvec) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = v[u][e] - mu;
                    v[u][e] = t;
                    q += t * t;
                }
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * inv_c + eps);
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int u = 0; u < VPL; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = v[u][e] * rstd, gy = d[u][e] * gg[u][e];
                v[u][e] = xh;
                s1 += gy;
                s2 += gy * xh;
                sg[u][e] += d[u][e] * xh;
                sb[u][e] += d[u][e];
            }
        const float m1 = wave_sum(s1) * inv_c, m2 = wave_sum(s2) * inv_c;
#pragma unroll
        for (int u = 0; u < VPL; ++u) {
            const int idx = lane + 64 * u;
            if (idx < nvec) {
                f32x4 o = acc[u];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] += rstd * (d[u][e] * gg[u][e] - m1 - v[u][e] * m2);
                h4[idx] = o;
                if (RES) {
                    // resid_bwd_tile's statements on the fresh dh row: g = dh rs, dz = g lam (bf16), sums of dz and of g z
                    const float rs = rf.rowscale ? rf.rowscale[row] : 1.0f;
                    const f32x4 l4 = lp[idx];
                    bf16x4 pk;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float gr = o[e] * rs;
                        const float a = gr * l4[e];
                        pk[e] = (bf16_t)a;
                        sa[u][e] += a;
                        sq[u][e] += gr * (float)zr[u][e];
                    }
                    reinterpret_cast<bf16x4 *>(rf.dz + row * C)[idx] = pk;
                }
            }
        }
    }
    // the waves' column sums -> one partial row per workgroup (fixed order)
#pragma unroll
    for (int u = 0; u < VPL; ++u) {
        const int idx = lane + 64 * u;
        if (idx < nvec) {
            reinterpret_cast<f32x4 *>(red + (0 * NWV + wave) * C)[idx] = sg[u];
            reinterpret_cast<f32x4 *>(red + (1 * NWV + wave) * C)[idx] = sb[u];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 64 * NWV) {
        float a = 0.0f, b = 0.0f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) {
            a += red[(0 * NWV + w) * C + c];
            b += red[(1 * NWV + w) * C + c];
        }
        dg_part[(size_t)blockIdx.x * C + c] = a;
        db_part[(size_t)blockIdx.x * C + c] = b;
    }
    if (RES) {
        __syncthreads();            // the buffer is read out: second round for the LayerScale sums
#pragma unroll
        for (int u = 0; u < VPL; ++u) {
            const int idx = lane + 64 * u;
            if (idx < nvec) {
                reinterpret_cast<f32x4 *>(red + (0 * NWV + wave) * C)[idx] = sa[u];
                reinterpret_cast<f32x4 *>(red + (1 * NWV + wave) * C)[idx] = sq[u];
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 64 * NWV) {
            float a = 0.0f, b = 0.0f;
#pragma unroll
            for (int w = 0; w < NWV; ++w) {
                a += red[(0 * NWV + w) * C + c];
                b += red[(1 * NWV + w) * C + c];
            }
            rf.dzb_part[(size_t)blockIdx.x * C + c] = a;
            rf.dlam_part[(size_t)blockIdx.x * C + c] = b;
        }
    }
}

// ---- second stage of the column reductions / split-K slab sums ---------------------------------------------------------
// job j: out[n] = sum_{p < P} part[p * stride + n], n < N (fixed order).  grid.x = sum over jobs of ceil(N / 1024).
__global__ void __launch_bounds__(256) reduce_jobs_kernel(const ReduceJobs jobs)
{
    __shared__ f32x4 red[16][16];
    int blk = blockIdx.x, j = 0;
    while (j + 1 < jobs.n && blk >= jobs.first_block[j + 1]) ++j;
    blk -= jobs.first_block[j];
    if (jobs.P[j] > 16) {
        // TALL job (a bias / gamma / lambda gradient: hundreds of partial rows, a few thousand columns): a workgroup owns 64
        // columns, 16 row groups of 16 column quads - each thread walks P / 16 rows, then a fixed-order LDS sum.  (One thread
        // per column quad walking all 512 partial rows was a serial chain of ~128 memory round trips: 80 us per layer.)
        const int cq = threadIdx.x & 15, rg = threadIdx.x >> 4;
        const int64_t n = (int64_t)blk * 64 + cq * 4;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (n < jobs.N[j]) {
            const float *p = jobs.part[j] + n;
            const int64_t st = jobs.stride[j];
            // eight rows' loads in flight, added in row order (the bits of the rolled loop, a quarter of its round trips)
            const int P = jobs.P[j];
            for (int q = rg; q < P; q += 128) {
                f32x4 a[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    a[u] = q + 16 * u < P ? *reinterpret_cast<const f32x4 *>(p + (int64_t)(q + 16 * u) * st) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (q + 16 * u < P)
#pragma unroll
                        for (int e = 0; e < 4; ++e) s[e] += a[u][e];
            }
        }
        red[rg][cq] = s;
        __syncthreads();
        if (rg == 0 && n < jobs.N[j]) {
            f32x4 t = red[0][cq];
#pragma unroll
            for (int g = 1; g < 16; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] += red[g][cq][e];
            *reinterpret_cast<f32x4 *>(jobs.out[j] + n) = t;
        }
        return;
    }
    const int64_t n = ((int64_t)blk * 256 + threadIdx.x) * 4;      // four consecutive outputs per thread (16-byte accesses)
    if (n >= jobs.N[j]) return;
    const float *p = jobs.part[j] + n;
    const int64_t st = jobs.stride[j];
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    const int P = jobs.P[j];
    int q = 0;
    for (; q + 3 < P; q += 4) {            // four 16-byte loads in flight per thread
        const f32x4 a = *reinterpret_cast<const f32x4 *>(p + (int64_t)q * st), b = *reinterpret_cast<const f32x4 *>(p + (int64_t)(q + 1) * st);
        const f32x4 c = *reinterpret_cast<const f32x4 *>(p + (int64_t)(q + 2) * st), d = *reinterpret_cast<const f32x4 *>(p + (int64_t)(q + 3) * st);
#pragma unroll
        for (int e = 0; e < 4; ++e) { s0[e] += a[e]; s1[e] += b[e]; s2[e] += c[e]; s3[e] += d[e]; }
    }
    if (q + 2 < P) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(p + (int64_t)q * st), b = *reinterpret_cast<const f32x4 *>(p + (int64_t)(q + 1) * st);
        const f32x4 c = *reinterpret_cast<const f32x4 *>(p + (int64_t)(q + 2) * st);
#pragma unroll
        for (int e = 0; e < 4; ++e) { s0[e] += a[e]; s1[e] += b[e]; s2[e] += c[e]; }
    } else if (q + 1 < P) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(p + (int64_t)q * st), b = *reinterpret_cast<const f32x4 *>(p + (int64_t)(q + 1) * st);
#pragma unroll
        for (int e = 0; e < 4; ++e) { s0[e] += a[e]; s1[e] += b[e]; }
    } else if (q < P) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(p + (int64_t)q * st);
#pragma unroll
        for (int e = 0; e < 4; ++e) s0[e] += a[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) s0[e] = (s0[e] + s1[e]) + (s2[e] + s3[e]);
    *reinterpret_cast<f32x4 *>(jobs.out[j] + n) = s0;
}

// ---- small element-wise helpers ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) add_inplace_f32(float *__restrict__ a, const float *__restrict__ b, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        f32x4 x = reinterpret_cast<f32x4 *>(a)[i];
        const f32x4 y = reinterpret_cast<const f32x4 *>(b)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] += y[e];
        reinterpret_cast<f32x4 *>(a)[i] = x;
    }
}

// rowscale[j][m] = drop[j][m / T]  (per-sample stochastic-depth factor broadcast to the sample's token rows)
__global__ void __launch_bounds__(256) expand_rowscale(const float *__restrict__ drop, float *__restrict__ rowscale, int B, int T,
                                                       int nvec)
{
    const int m = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (j < nvec && m < B * T) rowscale[(size_t)j * B * T + m] = drop[(size_t)j * B + m / T];
}

// embeddings backward: dpos[t][c] = sum_b dh0[b][t][c]  (TF:168-172: the table is broadcast over the batch)
__global__ void __launch_bounds__(256) sum_over_batch(const float *__restrict__ dh0, float *__restrict__ dpos, int B, size_t tc4)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= tc4) return;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < B; ++b) {
        const f32x4 v = reinterpret_cast<const f32x4 *>(dh0)[(size_t)b * tc4 + i];
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += v[e];
    }
    reinterpret_cast<f32x4 *>(dpos)[i] = s;
}

// dcls[c] = dpos[0][c] ; dpatch_b[c] = sum_{t >= 1} dpos[t][c]
__global__ void __launch_bounds__(256) embed_small_grads(const float *__restrict__ dpos, float *__restrict__ dcls,
                                                         float *__restrict__ dpb, int T, int C)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    dcls[c] = dpos[c];
    float s = 0.0f;
    for (int t = 1; t < T; ++t) s += dpos[(size_t)t * C + c];
    dpb[c] = s;
}

// im2col of the image batch, transposed and rounded to bf16: out[k][m], k = (ch, dy, dx), m = (b, gy, gx) -
// the W operand of the patch-embedding wgrad (dW[c][k] = sum_m dE[m][c] patch[m][k]).  One workgroup per (b, gy, ch).
__global__ void __launch_bounds__(256) patches_transposed(const float *__restrict__ x, bf16_t *__restrict__ out, int in_ch,
                                                          int img_h, int img_w, int p, int gw, int gh, int Mp)
{
    const int ch = blockIdx.x % in_ch, gy = (blockIdx.x / in_ch) % gh, b = blockIdx.x / (in_ch * gh);
    const float *src = x + (((size_t)b * in_ch + ch) * img_h + (size_t)gy * p) * img_w;      // p rows of img_w floats
    for (int i = threadIdx.x; i < p * img_w; i += 256) {
        const int dy = i / img_w, xx = i - dy * img_w, gx = xx / p, dx = xx - gx * p;
        const int k = (ch * p + dy) * p + dx;
        const size_t m = ((size_t)b * gh + gy) * gw + gx;
        out[(size_t)k * Mp + m] = (bf16_t)src[i];
    }
}

// im2col of the image batch rounded to bf16, ROW-major: out[m][k], m = (b, gy, gx), k = (ch, dy, dx) - the reduction-major W
// operand of the patch-embedding wgrad.  One workgroup per (b, gy, ch): p image rows of img_w floats -> gw x (p*p) values.
// planes > 1 (split-fp32 builds): the row holds that many bf16 planes of the pixel side by side (row stride planes * Kp)
__global__ void __launch_bounds__(256) patches_rows(const float *__restrict__ x, bf16_t *__restrict__ out, int in_ch, int img_h,
                                                    int img_w, int p, int gw, int gh, int Kp, int planes)
{
    const int ch = blockIdx.x % in_ch, gy = (blockIdx.x / in_ch) % gh, b = blockIdx.x / (in_ch * gh);
    const float *src = x + (((size_t)b * in_ch + ch) * img_h + (size_t)gy * p) * img_w;
    for (int i = threadIdx.x; i < p * img_w; i += 256) {
        const int dy = i / img_w, xx = i - dy * img_w, gx = xx / p, dx = xx - gx * p;
        const size_t m = ((size_t)b * gh + gy) * gw + gx;
        float v = src[i];
        bf16_t *d = out + m * ((size_t)planes * Kp) + (ch * p + dy) * p + dx;
        for (int sp = 0; sp < planes; ++sp) {
            const bf16_t q = (bf16_t)v;
            d[(size_t)sp * Kp] = q;
            v -= (float)q;
        }
    }
}

// The same im2col rows straight from the detector's RAGGED image list ("input transform fused into the patch-embed
// load"): every value is the normalised, bilinearly resized pixel of image_blend.h - the statement ldit_preprocess_f32/_f16 evaluates,
// hence its bits - rounded to bf16 (or split into planes) on the spot.  The fp32 out_h x out_w batch is never written or read back
// (38.5 MB each way at ViT-B bs=64) and one launch disappears.  One workgroup per (image, patch row, channel) as above.
struct PatchImgArgs {
    ImageList l;
    bf16_t *out;
    int in_ch, out_h, out_w, p, gw, gh, Kp, planes;
};

template <typename IN>
__global__ void __launch_bounds__(256) patches_rows_images(const PatchImgArgs a)
{
    const int ch = blockIdx.x % a.in_ch, gy = (blockIdx.x / a.in_ch) % a.gh, i = blockIdx.x / (a.in_ch * a.gh);
    const int h = a.l.h[i], w = a.l.w[i], b = a.l.first + i;
    const IN *src = static_cast<const IN *>(a.l.img[i]) + (size_t)ch * h * w;
    for (int t = threadIdx.x; t < a.p * a.out_w; t += 256) {
        const int dy = t / a.out_w, xx = t - dy * a.out_w, gx = xx / a.p, dx = xx - gx * a.p;
        const BlendRow br = blend_row(gy * a.p + dy, h, a.out_h);
        float v = blend_pixel(src + (size_t)br.y0 * w, src + (size_t)br.y1 * w, xx, w, a.out_w, br, a.l.mean, a.l.inv_std);
        const size_t m = ((size_t)b * a.gh + gy) * a.gw + gx;
        bf16_t *d = a.out + m * ((size_t)a.planes * a.Kp) + (ch * a.p + dy) * a.p + dx;
        for (int sp = 0; sp < a.planes; ++sp) {
            const bf16_t q = (bf16_t)v;
            d[(size_t)sp * a.Kp] = q;
            v -= (float)q;
        }
    }
}

__global__ void __launch_bounds__(256) zero_pad_columns(bf16_t *__restrict__ buf, int rows, int M, int Mp)
{
    const int r = blockIdx.x, pad = Mp - M;
    if (r >= rows) return;
    for (int i = threadIdx.x; i < pad; i += 256) buf[(size_t)r * Mp + M + i] = (bf16_t)0.0f;
}

// ---- fused AdamW over the flat parameter block (torch.optim.AdamW semantics; ref trainer.py:62-68: lr 1e-4, wd 0) -----
// p *= 1 - lr wd ; m = b1 m + (1 - b1) g ; v = b2 v + (1 - b2) g^2 ; p -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps),
// g = grad * grad_scale (1 / world size after the all-reduce sum, times any loss-scale inverse).
__global__ void __launch_bounds__(256) adamw_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                    float *__restrict__ v, size_t n4, float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float bc2_sqrt, float grad_scale, bf16_t *__restrict__ mirror)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 P = reinterpret_cast<f32x4 *>(p)[i], Mo = reinterpret_cast<f32x4 *>(m)[i], Vo = reinterpret_cast<f32x4 *>(v)[i];
    const f32x4 G = reinterpret_cast<const f32x4 *>(g)[i];
    const float step = lr / bc1, decay = 1.0f - lr * wd;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float gr = G[e] * grad_scale;
        const float mo = b1 * Mo[e] + (1.0f - b1) * gr;
        const float vo = b2 * Vo[e] + (1.0f - b2) * gr * gr;
        const float denom = sqrtf(vo) / bc2_sqrt + eps;
        P[e] = P[e] * decay - step * (mo / denom);
        Mo[e] = mo;
        Vo[e] = vo;
    }
    reinterpret_cast<f32x4 *>(p)[i] = P;
    if (mirror) {           // bf16 operand copy of the updated parameters, refreshed in the same pass (no separate re-pack)
        const bf16x4 o = {(bf16_t)P[0], (bf16_t)P[1], (bf16_t)P[2], (bf16_t)P[3]};
        reinterpret_cast<bf16x4 *>(mirror)[i] = o;
    }
    reinterpret_cast<f32x4 *>(m)[i] = Mo;
    reinterpret_cast<f32x4 *>(v)[i] = Vo;
}

}  // namespace

#define LAUNCH_CHECKED(...)            \
    do {                               \
        hipLaunchKernelGGL(__VA_ARGS__); \
        LDIT_HIP_CHECK(hipGetLastError()); \
    } while (0)

int launch_transpose_bf16(const void *src, bool src_f32, void *dst, int M, int N, int ld_src, int Mp, int skip_tokens,
                          float *colsum_part, hipStream_t stream, void *rowmajor_copy)
{
    if (M <= 0 || N <= 0) return fail(LDIT_EINVAL, "transpose: empty problem");
    if (Mp % 64 || Mp < M) return fail(LDIT_EINVAL, "transpose: padded row count %d must be a multiple of 64 and >= %d", Mp, M);
    if (!src || !dst || (reinterpret_cast<uintptr_t>(dst) & 15u)) return fail(LDIT_EINVAL, "transpose: null or misaligned operand");
    const dim3 grid((unsigned)((N + 63) / 64), (unsigned)(Mp / 64));
    if (src_f32)
        LAUNCH_CHECKED((transpose_tile<true>), grid, dim3(256), 0, stream, src, static_cast<bf16_t *>(dst), M, N, ld_src, Mp, skip_tokens, colsum_part,
                       static_cast<bf16_t *>(rowmajor_copy));
    else
        LAUNCH_CHECKED((transpose_tile<false>), grid, dim3(256), 0, stream, src, static_cast<bf16_t *>(dst), M, N, ld_src, Mp, skip_tokens, colsum_part,
                       static_cast<bf16_t *>(nullptr));
    return LDIT_OK;
}

int launch_resid_bwd(const float *dh, const void *z, const float *lam, const float *rowscale, void *dz, void *dzT, int M, int C,
                     int Mp, float *dlam_part, float *db_part, hipStream_t stream)
{
    if (M <= 0 || C <= 0 || C % 64) return fail(LDIT_EUNSUPPORTED, "resid_bwd: C=%d must be a positive multiple of 64", C);
    if (Mp % 64 || Mp < M) return fail(LDIT_EINVAL, "resid_bwd: bad padded row count");
    LAUNCH_CHECKED(resid_bwd_tile, dim3((unsigned)(C / 64), (unsigned)(Mp / 64)), dim3(256), 0, stream, dh,
                   static_cast<const bf16_t *>(z), lam, rowscale, static_cast<bf16_t *>(dz), static_cast<bf16_t *>(dzT), M, C, Mp,
                   dlam_part, db_part);
    return LDIT_OK;
}

int launch_colsum_bf16(const void *src, int M, int N, int ld, float *part, hipStream_t stream)
{
    if (M <= 0 || N <= 0 || !src || !part) return fail(LDIT_EINVAL, "colsum: bad argument");
    if ((N & 7) || (ld & 7) || (reinterpret_cast<uintptr_t>(src) & 15u)) return fail(LDIT_EINVAL, "colsum: N, ld multiples of 8, 16-byte aligned source");
    LAUNCH_CHECKED(colsum_tile, dim3((unsigned)((N + 511) / 512), (unsigned)((M + 63) / 64)), dim3(256), 0, stream,
                   static_cast<const bf16_t *>(src), M, N, ld, part);
    return LDIT_OK;
}

int launch_rows_to_bf16(const float *src, void *dst, int M, int N, int skip_tokens, hipStream_t stream)
{
    if (M <= 0 || N <= 0 || (N & 3) || !src || !dst) return fail(LDIT_EINVAL, "rows_to_bf16: bad argument");
    const size_t total = (size_t)M * (N / 4);
    LAUNCH_CHECKED(rows_to_bf16, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src, static_cast<bf16_t *>(dst), M, N / 4,
                   skip_tokens);
    return LDIT_OK;
}

// two workgroups of eight waves per CU: a wave walks ~3 rows at bs=64, each one memory round trip - the latency is hidden by
// the waves beside it, not inside it - and 512 partial rows per gradient vector (1024 four-wave workgroups hid it as well but
// doubled the second-stage reduction)
int layernorm_bwd_blocks(int64_t rows) { return (int)(rows < 8 * 512 ? (rows + 7) / 8 : 512); }
// the fused LayerNorm + LayerScale backward runs four-wave workgroups, three per CU (C <= 1024; wider rows keep the layout above)
int layernorm_bwd_resid_blocks(int64_t rows) { return (int)(rows < 4 * 768 ? (rows + 3) / 4 : 768); }

template <bool RES>
static int launch_layernorm_bwd_t(const float *dy, const float *x, const float *g, float *dh, int64_t rows, int C, float eps,
                                  float *dg_part, float *db_part, const ResidFused &rf, hipStream_t stream)
{
    if (rows <= 0 || C <= 0) return fail(LDIT_EINVAL, "layernorm_bwd: empty problem");
    if ((C & 3) || C > 4096) return fail(LDIT_EUNSUPPORTED, "layernorm_bwd: C=%d must be a multiple of 4, at most 4096", C);
    if (!dy || !x || !g || !dh || !dg_part || !db_part) return fail(LDIT_EINVAL, "layernorm_bwd: null operand");
    if (RES && (!rf.z || !rf.lam || !rf.dz || !rf.dlam_part || !rf.dzb_part || (reinterpret_cast<uintptr_t>(rf.z) & 7u) ||
                (reinterpret_cast<uintptr_t>(rf.dz) & 7u) || !aligned16(rf.lam)))
        return fail(LDIT_EINVAL, "layernorm_bwd: fused LayerScale operands null or misaligned");
    const int blocks = (RES && C <= 1024) ? layernorm_bwd_resid_blocks(rows) : layernorm_bwd_blocks(rows);
    const size_t lds = (size_t)2 * ((C <= 1024 && !RES) ? 8 : 4) * C * sizeof(float);
    if (RES && C <= 1024) {
        // fused form: 145 - 180 VGPRs = three / two waves per SIMD, so FOUR-wave workgroups (three / two per CU) keep the CU full
        if (C <= 256) LAUNCH_CHECKED((layernorm_bwd_rows<1, 4, RES>), dim3(blocks), dim3(256), lds, stream, dy, x, g, dh, rows, C, eps, dg_part, db_part, rf);
        else if (C <= 768) LAUNCH_CHECKED((layernorm_bwd_rows<3, 4, RES>), dim3(blocks), dim3(256), lds, stream, dy, x, g, dh, rows, C, eps, dg_part, db_part, rf);
        else LAUNCH_CHECKED((layernorm_bwd_rows<4, 4, RES>), dim3(blocks), dim3(256), lds, stream, dy, x, g, dh, rows, C, eps, dg_part, db_part, rf);
    } else if (C <= 256) LAUNCH_CHECKED((layernorm_bwd_rows<1, 8, RES>), dim3(blocks), dim3(512), lds, stream, dy, x, g, dh, rows, C, eps, dg_part, db_part, rf);
    else if (C <= 768) LAUNCH_CHECKED((layernorm_bwd_rows<3, 8, RES>), dim3(blocks), dim3(512), lds, stream, dy, x, g, dh, rows, C, eps, dg_part, db_part, rf);
    else if (C <= 1024) LAUNCH_CHECKED((layernorm_bwd_rows<4, 8, RES>), dim3(blocks), dim3(512), lds, stream, dy, x, g, dh, rows, C, eps, dg_part, db_part, rf);
    else {
        LDIT_DYN_LDS((layernorm_bwd_rows<16, 4, RES>), 2 * 4 * 4096 * 4);
        LAUNCH_CHECKED((layernorm_bwd_rows<16, 4, RES>), dim3(blocks), dim3(256), lds, stream, dy, x, g, dh, rows, C, eps, dg_part, db_part, rf);
    }
    return LDIT_OK;
}

int launch_layernorm_bwd(const float *dy, const float *x, const float *g, float *dh, int64_t rows, int C, float eps,
                         float *dg_part, float *db_part, hipStream_t stream, const float *add)
{
    ResidFused rf{};
    rf.add = add;
    return launch_layernorm_bwd_t<false>(dy, x, g, dh, rows, C, eps, dg_part, db_part, rf, stream);
}

// LayerNorm backward + the LayerScale / residual backward of the branch below it (dz, and the partial sums of dz and of dh z rs per
// workgroup: layernorm_bwd_blocks(rows) partial rows each) in one pass over the rows
int launch_layernorm_bwd_resid(const float *dy, const float *x, const float *g, float *dh, int64_t rows, int C, float eps,
                               float *dg_part, float *db_part, const void *z, const float *lam, const float *rowscale, void *dz,
                               float *dlam_part, float *dzb_part, hipStream_t stream)
{
    ResidFused rf{};
    rf.z = static_cast<const bf16_t *>(z); rf.lam = lam; rf.rowscale = rowscale; rf.dz = static_cast<bf16_t *>(dz);
    rf.dlam_part = dlam_part; rf.dzb_part = dzb_part;
    return launch_layernorm_bwd_t<true>(dy, x, g, dh, rows, C, eps, dg_part, db_part, rf, stream);
}

int launch_reduce_jobs(ReduceJobs &jobs, hipStream_t stream)
{
    if (jobs.n <= 0) return LDIT_OK;
    int blocks = 0;
    for (int j = 0; j < jobs.n; ++j) {
        if ((jobs.N[j] & 3) || (jobs.stride[j] & 3) || (reinterpret_cast<uintptr_t>(jobs.part[j]) & 15u) || (reinterpret_cast<uintptr_t>(jobs.out[j]) & 15u))
            return fail(LDIT_EINVAL, "reduce: lengths and strides must be multiples of 4 floats, pointers 16-byte aligned");
        jobs.first_block[j] = blocks;
        blocks += jobs.P[j] > 16 ? (int)((jobs.N[j] + 63) / 64) : (int)((jobs.N[j] / 4 + 255) / 256);
    }
    LAUNCH_CHECKED(reduce_jobs_kernel, dim3(blocks), dim3(256), 0, stream, jobs);
    jobs.n = 0;
    return LDIT_OK;
}

int launch_add_inplace(float *a, const float *b, size_t n, hipStream_t stream)
{
    if (n % 4) return fail(LDIT_EINVAL, "add: length must be a multiple of 4");
    if (n == 0) return LDIT_OK;
    LAUNCH_CHECKED(add_inplace_f32, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, a, b, n / 4);
    return LDIT_OK;
}

int launch_expand_rowscale(const float *drop, float *rowscale, int B, int T, int nvec, hipStream_t stream)
{
    LAUNCH_CHECKED(expand_rowscale, dim3((unsigned)((B * T + 255) / 256), (unsigned)nvec), dim3(256), 0, stream, drop, rowscale, B, T, nvec);
    return LDIT_OK;
}

int launch_embed_bwd_small(const float *dh0, float *dpos, float *dcls, float *dpb, int B, int T, int C, hipStream_t stream)
{
    const size_t tc4 = (size_t)T * C / 4;
    LAUNCH_CHECKED(sum_over_batch, dim3((unsigned)((tc4 + 255) / 256)), dim3(256), 0, stream, dh0, dpos, B, tc4);
    LAUNCH_CHECKED(embed_small_grads, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, stream, dpos, dcls, dpb, T, C);
    return LDIT_OK;
}

int launch_patches_transposed(const float *x, void *out, int B, int in_ch, int img_h, int img_w, int p, int Mp, hipStream_t stream)
{
    const int gh = img_h / p, gw = img_w / p, M = B * gh * gw;
    LAUNCH_CHECKED(patches_transposed, dim3((unsigned)(B * gh * in_ch)), dim3(256), 0, stream, x, static_cast<bf16_t *>(out), in_ch,
                   img_h, img_w, p, gw, gh, Mp);
    if (Mp > M) LAUNCH_CHECKED(zero_pad_columns, dim3((unsigned)(in_ch * p * p)), dim3(256), 0, stream, static_cast<bf16_t *>(out), in_ch * p * p, M, Mp);
    return LDIT_OK;
}

int launch_patches_rows(const float *x, void *out, int B, int in_ch, int img_h, int img_w, int p, hipStream_t stream, int planes)
{
    const int gh = img_h / p, gw = img_w / p;
    LAUNCH_CHECKED(patches_rows, dim3((unsigned)(B * gh * in_ch)), dim3(256), 0, stream, x, static_cast<bf16_t *>(out), in_ch, img_h,
                   img_w, p, gw, gh, in_ch * p * p, planes);
    return LDIT_OK;
}

int launch_patches_rows_images(const void *const *images, bool half_in, const int *heights, const int *widths, int B, int in_ch,
                               float mean, float std, int out_h, int out_w, int p, void *out, hipStream_t stream, int planes)
{
    const int gh = out_h / p, gw = out_w / p;
    for (int first = 0; first < B; first += PRE_MAX) {
        PatchImgArgs a{};
        if (int rc = fill_image_list(a.l, images, heights, widths, B, first, mean, std)) return rc;
        a.out = static_cast<bf16_t *>(out); a.in_ch = in_ch; a.out_h = out_h; a.out_w = out_w; a.p = p; a.gw = gw; a.gh = gh;
        a.Kp = in_ch * p * p; a.planes = planes;
        if (half_in) LAUNCH_CHECKED(patches_rows_images<_Float16>, dim3((unsigned)(a.l.n * gh * in_ch)), dim3(256), 0, stream, a);
        else LAUNCH_CHECKED(patches_rows_images<float>, dim3((unsigned)(a.l.n * gh * in_ch)), dim3(256), 0, stream, a);
    }
    return LDIT_OK;
}

int launch_adamw(float *p, const float *g, float *m, float *v, size_t n, float lr, float b1, float b2, float eps, float wd,
                 int step, float grad_scale, void *mirror, hipStream_t stream)
{
    if (n == 0) return LDIT_OK;
    if (n % 4) return fail(LDIT_EINVAL, "adamw: length must be a multiple of 4");
    if (!p || !g || !m || !v || !aligned16(p) || !aligned16(g) || !aligned16(m) || !aligned16(v)) return fail(LDIT_EINVAL, "adamw: null or misaligned operand");
    if (step < 1) return fail(LDIT_EINVAL, "adamw: step counts from 1");
    const double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
    LAUNCH_CHECKED(adamw_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, p, g, m, v, n / 4, lr, b1, b2, eps, wd,
                   (float)bc1, (float)sqrt(bc2), grad_scale, static_cast<bf16_t *>(mirror));
    return LDIT_OK;
}

}  // namespace ldit
