// Small HBM-bound kernels around the GEMMs: CLS rows of the embedding, the fused q/k/v bias, and DiTBackbone's tap
// post-processing (CLS drop + [C,Gh,Gw] view + bilinear rescale, ref src/layoutdit/modeling/dit_backbone.py:50-61).
#include "ldit_common.h"

namespace ldit {

namespace {

// out[b, 0, :] = cls + pos[0]     (TF:models/beit/modeling_beit.py:168-172)
__global__ void __launch_bounds__(256) cls_rows(const float *__restrict__ cls, const float *__restrict__ pos,
                                                float *__restrict__ out, int B, int tokens, int C)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C;
    out[(size_t)b * tokens * C + c] = cls[c] + pos[c];
}

// dst[0:C] = bq ; dst[C:2C] = 0 (key projection has no bias, TF:306) ; dst[2C:3C] = bv
__global__ void __launch_bounds__(256) qkv_bias(const float *__restrict__ bq, const float *__restrict__ bv,
                                                float *__restrict__ dst, int C)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 3 * C) return;
    dst[i] = i < C ? bq[i] : (i < 2 * C ? 0.0f : bv[i - 2 * C]);
}

// One thread per 4 consecutive channels of one output pixel: the token-major tap is read as float4 along C
// (coalesced), the NCHW output is written per channel plane.  The four bilinear taps follow
// F.interpolate(scale_factor=s, mode="bilinear", align_corners=False): src = (dst + 0.5) / s - 0.5, clamped at 0,
// neighbour index clamped at size-1.
__global__ void __launch_bounds__(256) tap_to_map(const float *__restrict__ tap, float *__restrict__ out, int B, int Gh,
                                                  int Gw, int C, int Oh, int Ow, float inv_scale)
{
    const int c4n = C >> 2;
    const size_t total = (size_t)B * Oh * Ow * c4n;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c4 = (int)(idx % c4n);
    size_t t = idx / c4n;
    const int ox = (int)(t % Ow); t /= Ow;
    const int oy = (int)(t % Oh);
    const int b = (int)(t / Oh);
    float sy = ((float)oy + 0.5f) * inv_scale - 0.5f; sy = sy < 0.f ? 0.f : sy;
    float sx = ((float)ox + 0.5f) * inv_scale - 0.5f; sx = sx < 0.f ? 0.f : sx;
    int y0 = (int)sy; y0 = y0 > Gh - 1 ? Gh - 1 : y0;
    int x0 = (int)sx; x0 = x0 > Gw - 1 ? Gw - 1 : x0;
    const int y1 = y0 + (y0 < Gh - 1), x1 = x0 + (x0 < Gw - 1);
    const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const size_t N = (size_t)Gh * Gw + 1;
    const f32x4 *base = reinterpret_cast<const f32x4 *>(tap + ((size_t)b * N + 1) * C) + c4;
    const f32x4 v00 = base[(size_t)(y0 * Gw + x0) * c4n], v01 = base[(size_t)(y0 * Gw + x1) * c4n];
    const f32x4 v10 = base[(size_t)(y1 * Gw + x0) * c4n], v11 = base[(size_t)(y1 * Gw + x1) * c4n];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float v = hy * (hx * v00[e] + lx * v01[e]) + ly * (hx * v10[e] + lx * v11[e]);
        out[(((size_t)b * C + c4 * 4 + e) * Oh + oy) * Ow + ox] = v;
    }
}

}  // namespace

int launch_cls_rows(const float *cls, const float *pos, float *out, int B, int tokens, int C, hipStream_t stream)
{
    hipLaunchKernelGGL(cls_rows, dim3((B * C + 255) / 256), dim3(256), 0, stream, cls, pos, out, B, tokens, C);
    LDIT_HIP_CHECK(hipGetLastError());
    return LDIT_OK;
}

int launch_pack_qkv_bias(const float *bq, const float *bv, float *dst, int C, hipStream_t stream)
{
    hipLaunchKernelGGL(qkv_bias, dim3((3 * C + 255) / 256), dim3(256), 0, stream, bq, bv, dst, C);
    LDIT_HIP_CHECK(hipGetLastError());
    return LDIT_OK;
}

int launch_tap_to_map(const float *tap, float *out, int B, int Gh, int Gw, int C, float scale, hipStream_t stream)
{
    if (B <= 0 || Gh <= 0 || Gw <= 0 || C <= 0) return fail(LDIT_EINVAL, "tap_to_map: empty problem");
    if (C & 3) return fail(LDIT_EUNSUPPORTED, "tap_to_map: C=%d must be a multiple of 4", C);
    if (!(scale == 4.0f || scale == 2.0f || scale == 1.0f || scale == 0.5f))
        return fail(LDIT_EUNSUPPORTED, "tap_to_map: scale %g not in {4,2,1,0.5}", (double)scale);
    const int Oh = (int)((float)Gh * scale), Ow = (int)((float)Gw * scale);
    if (Oh <= 0 || Ow <= 0) return fail(LDIT_EINVAL, "tap_to_map: output collapses to zero size");
    const size_t total = (size_t)B * Oh * Ow * (C >> 2);
    hipLaunchKernelGGL(tap_to_map, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, tap, out, B, Gh, Gw, C, Oh,
                       Ow, 1.0f / scale);
    LDIT_HIP_CHECK(hipGetLastError());
    return LDIT_OK;
}

}  // namespace ldit
