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

// Tiled version: one workgroup = (image, CB channels).  The [tokens][CB] slab of the tap is read token-major
// (CB*4 contiguous bytes per token), parked in LDS with a CB+1 stride (distinct tokens -> distinct banks), and every
// thread then produces VEC consecutive output pixels of one channel plane, so a wave writes 64*VEC*4 contiguous bytes:
// the NCHW planes are written at full line width instead of 4 B per channel plane (1183 us -> HBM-bound for the x4 map).
template <int CB, int VEC>
__global__ void __launch_bounds__(256) tap_to_map_tiled(const float *__restrict__ tap, float *__restrict__ out, int Gh,
                                                        int Gw, int C, int Oh, int Ow, float inv_scale)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float *slab = reinterpret_cast<float *>(smem_raw);
    constexpr int LD = CB + 1;
    const int P = Gh * Gw, cblocks = C / CB;
    const int b = blockIdx.x / cblocks, c0 = (blockIdx.x % cblocks) * CB;
    const float *src = tap + ((size_t)b * (P + 1) + 1) * C + c0;
    for (int u = threadIdx.x; u < P * (CB / 4); u += 256) {
        const int tok = u / (CB / 4), c4 = (u % (CB / 4)) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(src + (size_t)tok * C + c4);
        float *d = slab + tok * LD + c4;
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
    __syncthreads();
    const int owv = Ow / VEC, per_ch = Oh * owv, items = CB * per_ch;
    for (int it = threadIdx.x; it < items; it += 256) {
        const int c = it / per_ch, rem = it - c * per_ch, oy = rem / owv, ox0 = (rem - oy * owv) * VEC;
        float sy = ((float)oy + 0.5f) * inv_scale - 0.5f; sy = sy < 0.f ? 0.f : sy;
        int y0 = (int)sy; y0 = y0 > Gh - 1 ? Gh - 1 : y0;
        const int y1 = y0 + (y0 < Gh - 1);
        const float ly = sy - (float)y0, hy = 1.f - ly;
        const float *r0 = slab + (y0 * Gw) * LD + c, *r1 = slab + (y1 * Gw) * LD + c;
        float o[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float sx = ((float)(ox0 + e) + 0.5f) * inv_scale - 0.5f; sx = sx < 0.f ? 0.f : sx;
            int x0 = (int)sx; x0 = x0 > Gw - 1 ? Gw - 1 : x0;
            const int x1 = x0 + (x0 < Gw - 1);
            const float lx = sx - (float)x0, hx = 1.f - lx;
            o[e] = hy * (hx * r0[x0 * LD] + lx * r0[x1 * LD]) + ly * (hx * r1[x0 * LD] + lx * r1[x1 * LD]);
        }
        float *dst = out + (((size_t)b * C + c0 + c) * Oh + oy) * Ow + ox0;
        if (VEC == 4) *reinterpret_cast<f32x4 *>(dst) = f32x4{o[0], o[1], o[2], o[3]};
        else
#pragma unroll
            for (int e = 0; e < VEC; ++e) dst[e] = o[e];
    }
}

// Input transform: up to PRE_MAX images per launch, descriptors in the kernel arguments (no device-side table).
// One thread per 4 consecutive output pixels of one (image, channel, row): reads are two input rows (L2-friendly),
// writes are 16 B per lane, contiguous across the wave.  normalise-then-resize == resize-then-normalise for a
// per-channel affine map, so the normalisation is applied to the interpolated value.
constexpr int PRE_MAX = 48;
struct PreArgs {
    const float *img[PRE_MAX];
    int h[PRE_MAX], w[PRE_MAX];
    float *out;
    int n, in_ch, out_h, out_w, first;
    float mean, inv_std;
};

__global__ void __launch_bounds__(256) preprocess_images(const PreArgs a)
{
    const int owv = (a.out_w + 3) / 4;
    const int per_img = a.in_ch * a.out_h * owv;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (idx >= per_img) return;
    const int ch = idx / (a.out_h * owv), rem = idx - ch * (a.out_h * owv), oy = rem / owv, ox0 = (rem - oy * owv) * 4;
    const int h = a.h[i], w = a.w[i];
    const float *src = a.img[i] + (size_t)ch * h * w;
    const float sch = (float)h / (float)a.out_h, scw = (float)w / (float)a.out_w;   // F.interpolate(size=...) scale
    float sy = ((float)oy + 0.5f) * sch - 0.5f; sy = sy < 0.f ? 0.f : sy;
    int y0 = (int)sy; y0 = y0 > h - 1 ? h - 1 : y0;
    const int y1 = y0 + (y0 < h - 1);
    const float ly = sy - (float)y0, hy = 1.f - ly;
    const float *r0 = src + (size_t)y0 * w, *r1 = src + (size_t)y1 * w;
    float *dst = a.out + (((size_t)(a.first + i) * a.in_ch + ch) * a.out_h + oy) * a.out_w + ox0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (ox0 + e >= a.out_w) break;
        float sx = ((float)(ox0 + e) + 0.5f) * scw - 0.5f; sx = sx < 0.f ? 0.f : sx;
        int x0 = (int)sx; x0 = x0 > w - 1 ? w - 1 : x0;
        const int x1 = x0 + (x0 < w - 1);
        const float lx = sx - (float)x0, hx = 1.f - lx;
        const float v = hy * (hx * r0[x0] + lx * r0[x1]) + ly * (hx * r1[x0] + lx * r1[x1]);
        dst[e] = (v - a.mean) * a.inv_std;
    }
}

template <int CB>
int launch_tiled_maps(const float *tap, float *out, int B, int Gh, int Gw, int C, int Oh, int Ow, float scale,
                      hipStream_t stream)
{
    const int lds = Gh * Gw * (CB + 1) * 4;
    const dim3 grid((unsigned)(B * (C / CB))), block(256);
    const bool vec = (Ow % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15u) == 0);
    if (vec) {
        auto kern = tap_to_map_tiled<CB, 4>;
        static bool attr_set = false;
        if (!attr_set) { LDIT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 65536)); attr_set = true; }
        hipLaunchKernelGGL(kern, grid, block, lds, stream, tap, out, Gh, Gw, C, Oh, Ow, 1.0f / scale);
    } else {
        auto kern = tap_to_map_tiled<CB, 1>;
        static bool attr_set = false;
        if (!attr_set) { LDIT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 65536)); attr_set = true; }
        hipLaunchKernelGGL(kern, grid, block, lds, stream, tap, out, Gh, Gw, C, Oh, Ow, 1.0f / scale);
    }
    LDIT_HIP_CHECK(hipGetLastError());
    return LDIT_OK;
}

}  // namespace

int launch_preprocess(const float *const *images, const int *heights, const int *widths, int B, int in_ch, float mean,
                      float std, int out_h, int out_w, float *out, hipStream_t stream)
{
    for (int first = 0; first < B; first += PRE_MAX) {
        PreArgs a{};
        a.n = (B - first) < PRE_MAX ? (B - first) : PRE_MAX;
        for (int i = 0; i < a.n; ++i) {
            if (!images[first + i] || heights[first + i] <= 0 || widths[first + i] <= 0)
                return fail(LDIT_EINVAL, "preprocess: image %d is null or empty", first + i);
            a.img[i] = images[first + i]; a.h[i] = heights[first + i]; a.w[i] = widths[first + i];
        }
        a.out = out; a.in_ch = in_ch; a.out_h = out_h; a.out_w = out_w; a.first = first; a.mean = mean; a.inv_std = 1.0f / std;
        const int per_img = in_ch * out_h * ((out_w + 3) / 4);
        hipLaunchKernelGGL(preprocess_images, dim3((per_img + 255) / 256, a.n), dim3(256), 0, stream, a);
        LDIT_HIP_CHECK(hipGetLastError());
    }
    return LDIT_OK;
}

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
    // channel block: the largest of 64/32/16 that divides C and whose [tokens][CB+1] slab fits 64 KB of LDS
    const int P = Gh * Gw;
    if (C % 64 == 0 && P * 65 * 4 <= 65536) return launch_tiled_maps<64>(tap, out, B, Gh, Gw, C, Oh, Ow, scale, stream);
    if (C % 32 == 0 && P * 33 * 4 <= 65536) return launch_tiled_maps<32>(tap, out, B, Gh, Gw, C, Oh, Ow, scale, stream);
    if (C % 16 == 0 && P * 17 * 4 <= 65536) return launch_tiled_maps<16>(tap, out, B, Gh, Gw, C, Oh, Ow, scale, stream);
    const size_t total = (size_t)B * Oh * Ow * (C >> 2);
    hipLaunchKernelGGL(tap_to_map, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, tap, out, B, Gh, Gw, C, Oh,
                       Ow, 1.0f / scale);
    LDIT_HIP_CHECK(hipGetLastError());
    return LDIT_OK;
}

}  // namespace ldit
