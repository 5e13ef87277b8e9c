// Backward of the FPN stage behind DiTWithFPN (ref src/layoutdit/modeling/dit_backbone.py:87-90 under
// loss.backward(), ref src/layoutdit/training/trainer.py:169-178): the HBM-bound pieces.  The MFMA pieces reuse existing
// kernels (layoutdit_amd/modeling/dit_fpn.py sequences them):
//   dgrad of a 3x3 convolution  = the SAME implicit-im2col fp32 MFMA GEMM on the spatially flipped, in/out-swapped weight
//   wgrad of a 3x3 convolution  = nine reduction-major bf16 GEMMs (gemm_bf16_tr.hip, wgrad form), one per tap, on ZERO-PADDED
//                                 NHWC copies of dY and of the convolution's input: with both maps stored [B, H+2, W+2, C] the
//                                 tap (ky, kx) is a constant row offset (ky-1)(W+2) + (kx-1) of the input operand - the border
//                                 rows of dY are zero, so every term that the offset drags across a row or image end vanishes
//   laterals                    = ldit_linear_f32 (dgrad, on the transposed weight) / the same bf16 wgrad form
// Here:  (1) the adjoint of fpn_merge_nhwc (misc.hip): bilinear rescale of the lateral tokens + nearest top-down add,
//        (2) fp32 NHWC -> zero-padded bf16 NHWC copies for (wgrad),
//        (3) column sums (bias gradients), two stages, fixed order - no atomics anywhere, every result bit-reproducible.
#include "ldit_common.h"

namespace ldit {
namespace {

typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));

// weight of source index g in the bilinear sample of output index o (the forward's own index arithmetic, misc.hip)
__device__ __forceinline__ float axis_w(int o, int g, int G, float inv_scale)
{
    float s = ((float)o + 0.5f) * inv_scale - 0.5f;
    s = s < 0.f ? 0.f : s;
    int i0 = (int)s;
    i0 = i0 > G - 1 ? G - 1 : i0;
    const int i1 = i0 + (i0 < G - 1);
    const float l = s - (float)i0;
    return (i0 == g ? 1.f - l : 0.f) + (i1 == g ? l : 0.f);
}

// d_lat[b, 1 + gy Gw + gx, :] = sum over the pixels of d_inner [B, Oh, Ow, Ch] whose bilinear footprint holds token (gy, gx);
// CLS row = 0.  One thread per (token, 4 channels): gather, fixed order.
__global__ void __launch_bounds__(256) fpn_merge_bwd_lat(const float *__restrict__ din, float *__restrict__ dlat, int B, int Gh, int Gw,
                                                         int Ch, int Oh, int Ow, float scale)
{
    const int c4n = Ch >> 2;
    const size_t total = (size_t)B * Gh * Gw * c4n, idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c4 = (int)(idx % c4n);
    size_t t = idx / c4n;
    const int gx = (int)(t % Gw);
    t /= Gw;
    const int gy = (int)(t % Gh), b = (int)(t / Gh);
    const float inv = 1.0f / scale;
    int ylo = (int)floorf(scale * (float)(gy - 1)) - 1, yhi = (int)ceilf(scale * (float)(gy + 2)) + 1;
    int xlo = (int)floorf(scale * (float)(gx - 1)) - 1, xhi = (int)ceilf(scale * (float)(gx + 2)) + 1;
    ylo = ylo < 0 ? 0 : ylo; xlo = xlo < 0 ? 0 : xlo;
    yhi = yhi > Oh - 1 ? Oh - 1 : yhi; xhi = xhi > Ow - 1 ? Ow - 1 : xhi;
    const f32x4 *src = reinterpret_cast<const f32x4 *>(din) + (size_t)b * Oh * Ow * c4n + c4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int oy = ylo; oy <= yhi; ++oy) {
        const float wy = axis_w(oy, gy, Gh, inv);
        if (wy == 0.0f) continue;
        f32x4 row = {0.f, 0.f, 0.f, 0.f};
        for (int ox = xlo; ox <= xhi; ++ox) {
            const float wx = axis_w(ox, gx, Gw, inv);
            if (wx == 0.0f) continue;
            const f32x4 v = src[(size_t)(oy * Ow + ox) * c4n];
#pragma unroll
            for (int e = 0; e < 4; ++e) row[e] += wx * v[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += wy * row[e];
    }
    f32x4 *dst = reinterpret_cast<f32x4 *>(dlat) + (size_t)b * (Gh * Gw + 1) * c4n;
    dst[(size_t)(1 + gy * Gw + gx) * c4n + c4] = acc;
    if (gy == 0 && gx == 0) dst[c4] = f32x4{0.f, 0.f, 0.f, 0.f};            // CLS row
}

// dtop[b, ty, tx, :] += sum over the pixels (oy, ox) of d_inner with nearest source (ty, tx): src = min(floor(o in/out), in-1)
__global__ void __launch_bounds__(256) fpn_merge_bwd_top(const float *__restrict__ din, float *__restrict__ dtop, int B, int Ch, int Oh,
                                                         int Ow, int Th, int Tw)
{
    const int c4n = Ch >> 2;
    const size_t total = (size_t)B * Th * Tw * c4n, idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c4 = (int)(idx % c4n);
    size_t t = idx / c4n;
    const int tx = (int)(t % Tw);
    t /= Tw;
    const int ty = (int)(t % Th), b = (int)(t / Th);
    const float ry = (float)Th / (float)Oh, rx = (float)Tw / (float)Ow;
    auto src_of = [](int o, float r, int n) { int s = (int)floorf((float)o * r); return s > n - 1 ? n - 1 : s; };
    int ylo = (int)floorf((float)ty / ry) - 2, yhi = (int)ceilf((float)(ty + 1) / ry) + 2;
    int xlo = (int)floorf((float)tx / rx) - 2, xhi = (int)ceilf((float)(tx + 1) / rx) + 2;
    ylo = ylo < 0 ? 0 : ylo; xlo = xlo < 0 ? 0 : xlo;
    yhi = yhi > Oh - 1 ? Oh - 1 : yhi; xhi = xhi > Ow - 1 ? Ow - 1 : xhi;
    if (ty == Th - 1) yhi = Oh - 1;                  // the clamp sends everything past the end to the last source row / column
    if (tx == Tw - 1) xhi = Ow - 1;
    const f32x4 *src = reinterpret_cast<const f32x4 *>(din) + (size_t)b * Oh * Ow * c4n + c4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int oy = ylo; oy <= yhi; ++oy) {
        if (src_of(oy, ry, Th) != ty) continue;
        for (int ox = xlo; ox <= xhi; ++ox) {
            if (src_of(ox, rx, Tw) != tx) continue;
            const f32x4 v = src[(size_t)(oy * Ow + ox) * c4n];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += v[e];
        }
    }
    f32x4 *d = reinterpret_cast<f32x4 *>(dtop) + idx;
    f32x4 cur = *d;
#pragma unroll
    for (int e = 0; e < 4; ++e) cur[e] += acc[e];
    *d = cur;
}

// dst bf16 [B, H+2, W+2, C] = zero-padded copy of src fp32 [B, H, W, C]; every element of dst is written
__global__ void __launch_bounds__(256) pad_nhwc_bf16(const float *__restrict__ src, __bf16 *__restrict__ dst, int B, int H, int W, int C)
{
    const int c4n = C >> 2;
    const size_t total = (size_t)B * (H + 2) * (W + 2) * c4n, idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c4 = (int)(idx % c4n);
    size_t t = idx / c4n;
    const int px = (int)(t % (W + 2));
    t /= (W + 2);
    const int py = (int)(t % (H + 2)), b = (int)(t / (H + 2));
    bf16x4_t o = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
    if (py >= 1 && py <= H && px >= 1 && px <= W) {
        const f32x4 v = reinterpret_cast<const f32x4 *>(src)[((size_t)(b * H + py - 1) * W + px - 1) * c4n + c4];
        o = bf16x4_t{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    }
    reinterpret_cast<bf16x4_t *>(dst)[idx] = o;
}

constexpr int CS_ROWS = 512;

// ABSMAX = false: column sums; true: column maxima of |x| (fp8 calibration: per-channel activation ranges)
template <bool ABSMAX> __device__ __forceinline__ float cs_op(float acc, float v) { return ABSMAX ? fmaxf(acc, fabsf(v)) : acc + v; }

// stage 1: part[blockIdx.x][n] = sum of rows [blockIdx.x * CS_ROWS, +CS_ROWS) of x[:, n]; a thread owns a column
template <bool ABSMAX>
__global__ void __launch_bounds__(256) colsum_stage1(const float *__restrict__ x, float *__restrict__ part, long M, int N, long ld)
{
    const int n = blockIdx.y * 256 + threadIdx.x;
    if (n >= N) return;
    const long m0 = (long)blockIdx.x * CS_ROWS, m1 = m0 + CS_ROWS < M ? m0 + CS_ROWS : M;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    long m = m0;
    for (; m + 3 < m1; m += 4) {             // four rows in flight; the summation order is a fixed function of (M, row block)
        s0 = cs_op<ABSMAX>(s0, x[m * ld + n]);
        s1 = cs_op<ABSMAX>(s1, x[(m + 1) * ld + n]);
        s2 = cs_op<ABSMAX>(s2, x[(m + 2) * ld + n]);
        s3 = cs_op<ABSMAX>(s3, x[(m + 3) * ld + n]);
    }
    for (; m < m1; ++m) s0 = cs_op<ABSMAX>(s0, x[m * ld + n]);
    part[(long)blockIdx.x * N + n] = ABSMAX ? fmaxf(fmaxf(s0, s1), fmaxf(s2, s3)) : (s0 + s1) + (s2 + s3);
}

template <bool ABSMAX>
__global__ void __launch_bounds__(256) colsum_stage2(const float *__restrict__ part, float *__restrict__ out, int P, int N)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int p = 0; p < P; ++p) s = ABSMAX ? fmaxf(s, part[(long)p * N + n]) : s + part[(long)p * N + n];
    out[n] = s;
}

}  // namespace

int launch_fpn_merge_bwd(const float *din, float *dlat, float *dtop, int B, int Gh, int Gw, int Ch, float scale, int top_h, int top_w,
                         hipStream_t stream)
{
    if (B <= 0 || Gh <= 0 || Gw <= 0 || Ch <= 0 || (Ch & 3)) return fail(LDIT_EINVAL, "fpn_merge_bwd: bad geometry");
    if (!(scale == 4.0f || scale == 2.0f || scale == 1.0f || scale == 0.5f)) return fail(LDIT_EUNSUPPORTED, "fpn_merge_bwd: scale %g not in {4,2,1,0.5}", (double)scale);
    if (!din || !aligned16(din) || (dlat && !aligned16(dlat)) || (dtop && !aligned16(dtop))) return fail(LDIT_EINVAL, "fpn_merge_bwd: null or misaligned operand");
    const int Oh = (int)((float)Gh * scale), Ow = (int)((float)Gw * scale);
    if (Oh <= 0 || Ow <= 0) return fail(LDIT_EINVAL, "fpn_merge_bwd: map collapses to zero size");
    if (dtop && (top_h <= 0 || top_w <= 0)) return fail(LDIT_EINVAL, "fpn_merge_bwd: bad top size");
    if ((size_t)B * Oh * Ow * (Ch >> 2) >= (1ull << 31)) return fail(LDIT_EUNSUPPORTED, "fpn_merge_bwd: map exceeds 2^31 vectors");
    if (dlat) {
        const size_t total = (size_t)B * Gh * Gw * (Ch >> 2);
        hipLaunchKernelGGL(fpn_merge_bwd_lat, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, din, dlat, B, Gh, Gw, Ch, Oh, Ow,
                           scale);
        LDIT_HIP_CHECK(hipGetLastError());
    }
    if (dtop) {
        const size_t total = (size_t)B * top_h * top_w * (Ch >> 2);
        hipLaunchKernelGGL(fpn_merge_bwd_top, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, din, dtop, B, Ch, Oh, Ow,
                           top_h, top_w);
        LDIT_HIP_CHECK(hipGetLastError());
    }
    return LDIT_OK;
}

int launch_pad_nhwc_bf16(const float *src, void *dst, int B, int H, int W, int C, hipStream_t stream)
{
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return fail(LDIT_EINVAL, "pad_nhwc_bf16: bad geometry");
    if (!src || !dst || !aligned16(src) || (reinterpret_cast<uintptr_t>(dst) & 7u)) return fail(LDIT_EINVAL, "pad_nhwc_bf16: null or misaligned operand");
    const size_t total = (size_t)B * (H + 2) * (W + 2) * (C >> 2);
    if (total >= (1ull << 31)) return fail(LDIT_EUNSUPPORTED, "pad_nhwc_bf16: map exceeds 2^31 vectors");
    hipLaunchKernelGGL(pad_nhwc_bf16, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src, static_cast<__bf16 *>(dst), B, H, W, C);
    LDIT_HIP_CHECK(hipGetLastError());
    return LDIT_OK;
}

size_t colsum_scratch_bytes(int64_t M, int64_t N) { return (size_t)((M + CS_ROWS - 1) / CS_ROWS) * (size_t)N * sizeof(float); }

int launch_colsum_f32(const float *x, int64_t M, int N, int64_t ld, float *out, float *scratch, size_t scratch_bytes, bool absmax,
                      hipStream_t stream)
{
    if (M <= 0 || N <= 0 || ld < N) return fail(LDIT_EINVAL, "colsum: bad geometry");
    if (!x || !out || !scratch) return fail(LDIT_EINVAL, "colsum: null operand");
    if (scratch_bytes < colsum_scratch_bytes(M, N)) return fail(LDIT_EWORKSPACE, "colsum: scratch %zu bytes < required %zu", scratch_bytes, colsum_scratch_bytes(M, N));
    const long P = (M + CS_ROWS - 1) / CS_ROWS;
    if (P > 65535L * 32768L) return fail(LDIT_EUNSUPPORTED, "colsum: too many rows");
    const dim3 g1((unsigned)P, (unsigned)((N + 255) / 256)), g2((unsigned)((N + 255) / 256));
    if (absmax) {
        hipLaunchKernelGGL(colsum_stage1<true>, g1, dim3(256), 0, stream, x, scratch, (long)M, N, (long)ld);
        hipLaunchKernelGGL(colsum_stage2<true>, g2, dim3(256), 0, stream, scratch, out, (int)P, N);
    } else {
        hipLaunchKernelGGL(colsum_stage1<false>, g1, dim3(256), 0, stream, x, scratch, (long)M, N, (long)ld);
        hipLaunchKernelGGL(colsum_stage2<false>, g2, dim3(256), 0, stream, scratch, out, (int)P, N);
    }
    LDIT_HIP_CHECK(hipGetLastError());
    return LDIT_OK;
}

}  // namespace ldit
