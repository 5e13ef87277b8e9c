// Small HBM-bound kernels around the GEMMs: CLS rows of the embedding, the fused q/k/v bias, and DiTBackbone's tap
// post-processing (CLS drop + [C,Gh,Gw] view + bilinear rescale, ref src/layoutdit/modeling/dit_backbone.py:50-61).
#include "ldit_common.h"
#include "image_blend.h"

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
// (qmul: pack-time factor on the query bias, see ldit_pack_weights)
__global__ void __launch_bounds__(256) qkv_bias(const float *__restrict__ bq, const float *__restrict__ bv,
                                                float *__restrict__ dst, int C, float qmul)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 3 * C) return;
    dst[i] = i < C ? bq[i] * qmul : (i < 2 * C ? 0.0f : bv[i - 2 * C]);
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
// per-channel affine map, so the normalisation is applied to the interpolated value (image_blend.h: the one statement every
// producer of these pixels shares).
struct PreArgs {
    ImageList l;
    float *out;
    int in_ch, out_h, out_w;
};

// IN = float or _Float16 (the reference's trainer hands fp16 images to the detector, ref trainer.py:153-155); arithmetic and
// output stay fp32.
template <typename IN>
__global__ void __launch_bounds__(256) preprocess_images(const PreArgs a)
{
    const int owv = (a.out_w + 3) / 4;
    const int per_img = a.in_ch * a.out_h * owv;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (idx >= per_img) return;
    const int ch = idx / (a.out_h * owv), rem = idx - ch * (a.out_h * owv), oy = rem / owv, ox0 = (rem - oy * owv) * 4;
    const int h = a.l.h[i], w = a.l.w[i];
    const IN *src = static_cast<const IN *>(a.l.img[i]) + (size_t)ch * h * w;
    const BlendRow br = blend_row(oy, h, a.out_h);
    const IN *r0 = src + (size_t)br.y0 * w, *r1 = src + (size_t)br.y1 * w;
    float *dst = a.out + (((size_t)(a.l.first + i) * a.in_ch + ch) * a.out_h + oy) * a.out_w + ox0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (ox0 + e >= a.out_w) break;
        dst[e] = blend_pixel(r0, r1, ox0 + e, w, a.out_w, br, a.l.mean, a.l.inv_std);
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
        LDIT_DYN_LDS(kern, 65536);
        hipLaunchKernelGGL(kern, grid, block, lds, stream, tap, out, Gh, Gw, C, Oh, Ow, 1.0f / scale);
    } else {
        auto kern = tap_to_map_tiled<CB, 1>;
        LDIT_DYN_LDS(kern, 65536);
        hipLaunchKernelGGL(kern, grid, block, lds, stream, tap, out, Gh, Gw, C, Oh, Ow, 1.0f / scale);
    }
    LDIT_HIP_CHECK(hipGetLastError());
    return LDIT_OK;
}

// fp16 <-> fp32 batch conversion (the encoder computes on fp32 pixels and returns fp32 taps; an fp16 caller gets both ends
// converted by the library, not by a host-framework cast)
__global__ void __launch_bounds__(256) widen_f16(const _Float16 *__restrict__ src, float *__restrict__ dst, size_t n)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        const h4 v = *reinterpret_cast<const h4 *>(src + i);
        *reinterpret_cast<f32x4 *>(dst + i) = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    } else {
        for (size_t k = i; k < n; ++k) dst[k] = (float)src[k];
    }
}

__global__ void __launch_bounds__(256) narrow_f16(const float *__restrict__ src, _Float16 *__restrict__ dst, size_t n)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        const f32x4 v = *reinterpret_cast<const f32x4 *>(src + i);
        *reinterpret_cast<h4 *>(dst + i) = h4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    } else {
        for (size_t k = i; k < n; ++k) dst[k] = (_Float16)src[k];
    }
}

}  // namespace

int launch_cast_f16(const void *src, void *dst, size_t n, bool widen, hipStream_t stream)
{
    if (n == 0) return LDIT_OK;
    const dim3 grid((unsigned)((n / 4 + 255) / 256 + 1));
    if (widen) hipLaunchKernelGGL(widen_f16, grid, dim3(256), 0, stream, static_cast<const _Float16 *>(src), static_cast<float *>(dst), n);
    else hipLaunchKernelGGL(narrow_f16, grid, dim3(256), 0, stream, static_cast<const float *>(src), static_cast<_Float16 *>(dst), n);
    LDIT_HIP_CHECK(hipGetLastError());
    return LDIT_OK;
}

int fill_image_list(ImageList &l, const void *const *images, const int *heights, const int *widths, int B, int first, float mean, float std)
{
    l.n = (B - first) < PRE_MAX ? (B - first) : PRE_MAX;
    for (int i = 0; i < l.n; ++i) {
        if (!images[first + i] || heights[first + i] <= 0 || widths[first + i] <= 0)
            return fail(LDIT_EINVAL, "image %d is null or empty", first + i);
        l.img[i] = images[first + i]; l.h[i] = heights[first + i]; l.w[i] = widths[first + i];
    }
    l.first = first; l.mean = mean; l.inv_std = 1.0f / std;
    return LDIT_OK;
}

int launch_preprocess(const void *const *images, bool half_in, const int *heights, const int *widths, int B, int in_ch, float mean,
                      float std, int out_h, int out_w, float *out, hipStream_t stream)
{
    for (int first = 0; first < B; first += PRE_MAX) {
        PreArgs a{};
        if (int rc = fill_image_list(a.l, images, heights, widths, B, first, mean, std)) return rc;
        a.out = out; a.in_ch = in_ch; a.out_h = out_h; a.out_w = out_w;
        const int per_img = in_ch * out_h * ((out_w + 3) / 4);
        if (half_in) hipLaunchKernelGGL(preprocess_images<_Float16>, dim3((per_img + 255) / 256, a.l.n), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL(preprocess_images<float>, dim3((per_img + 255) / 256, a.l.n), dim3(256), 0, stream, a);
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

int launch_pack_qkv_bias(const float *bq, const float *bv, float *dst, int C, hipStream_t stream, float qmul)
{
    hipLaunchKernelGGL(qkv_bias, dim3((3 * C + 255) / 256), dim3(256), 0, stream, bq, bv, dst, C, qmul);
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

// ---- FPN (ref src/layoutdit/modeling/dit_backbone.py:78-90: torchvision FeaturePyramidNetwork([C]*4, 256, LastLevelMaxPool)) ---
// One level of the top-down pathway, NHWC:   inner[b, y, x, :] = bilinear_s(lat)[b, y, x, :] + nearest(top)[b, y, x, :]
// `lat` = the level's 1x1 lateral convolution applied to the TOKENS of its tap, [B, 1 + Gh*Gw, Ch] (CLS row unused): a 1x1
// convolution and a bilinear resize commute (both linear, the resize's weights sum to one, so the bias passes too), hence
// the lateral runs on 196 tokens instead of on the 56 x 56 map the reference feeds it - 16x fewer FLOPs at p2 and no
// 768-channel map is ever materialised.  Resize semantics = F.interpolate(scale_factor=s, mode="bilinear",
// align_corners=False) of ref dit_backbone.py:55-59; `top` (optional, [B, top_h, top_w, Ch]) is resampled to this level
// with F.interpolate(size=..., mode="nearest") semantics: src = min(floor(dst * in / out), in - 1).
namespace ldit {
namespace {

__global__ void __launch_bounds__(256) fpn_merge_nhwc(const float *__restrict__ lat, const float *__restrict__ top,
                                                      float *__restrict__ out, int B, int Gh, int Gw, int Ch, int Oh, int Ow,
                                                      float inv_scale, int top_h, int top_w)
{
    const int c4n = Ch >> 2;
    const size_t total = (size_t)B * Oh * Ow * c4n, i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i % c4n);
    size_t r = i / c4n;
    const int ox = (int)(r % Ow);
    r /= Ow;
    const int oy = (int)(r % Oh), b = (int)(r / Oh);
    float sy = ((float)oy + 0.5f) * inv_scale - 0.5f, sx = ((float)ox + 0.5f) * inv_scale - 0.5f;
    sy = sy < 0.f ? 0.f : sy;
    sx = sx < 0.f ? 0.f : sx;
    int y0 = (int)sy, x0 = (int)sx;
    y0 = y0 > Gh - 1 ? Gh - 1 : y0;
    x0 = x0 > Gw - 1 ? Gw - 1 : x0;
    const int y1 = y0 + (y0 < Gh - 1), x1 = x0 + (x0 < Gw - 1);
    const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const f32x4 *tok = reinterpret_cast<const f32x4 *>(lat + ((size_t)b * (Gh * Gw + 1) + 1) * Ch) + c4;
    const f32x4 v00 = tok[(size_t)(y0 * Gw + x0) * c4n], v01 = tok[(size_t)(y0 * Gw + x1) * c4n];
    const f32x4 v10 = tok[(size_t)(y1 * Gw + x0) * c4n], v11 = tok[(size_t)(y1 * Gw + x1) * c4n];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = hy * (hx * v00[e] + lx * v01[e]) + ly * (hx * v10[e] + lx * v11[e]);
    if (top) {
        int ty = (int)floorf((float)oy * ((float)top_h / (float)Oh)), tx = (int)floorf((float)ox * ((float)top_w / (float)Ow));
        ty = ty > top_h - 1 ? top_h - 1 : ty;
        tx = tx > top_w - 1 ? top_w - 1 : tx;
        const f32x4 t = reinterpret_cast<const f32x4 *>(top)[((size_t)(b * top_h + ty) * top_w + tx) * c4n + c4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += t[e];
    }
    reinterpret_cast<f32x4 *>(out)[i] = o;
}

}  // namespace

int launch_fpn_merge(const float *lat, const float *top, float *out, int B, int Gh, int Gw, int Ch, float scale, int top_h,
                     int top_w, hipStream_t stream)
{
    if (B <= 0 || Gh <= 0 || Gw <= 0 || Ch <= 0 || (Ch & 3)) return fail(LDIT_EINVAL, "fpn_merge: bad geometry");
    if (!(scale == 4.0f || scale == 2.0f || scale == 1.0f || scale == 0.5f)) return fail(LDIT_EUNSUPPORTED, "fpn_merge: scale %g not in {4,2,1,0.5}", (double)scale);
    if (!lat || !out || !aligned16(lat) || !aligned16(out) || (top && !aligned16(top))) return fail(LDIT_EINVAL, "fpn_merge: null or misaligned operand");
    const int Oh = (int)((float)Gh * scale), Ow = (int)((float)Gw * scale);
    if (Oh <= 0 || Ow <= 0) return fail(LDIT_EINVAL, "fpn_merge: output collapses to zero size");
    if (top && (top_h <= 0 || top_w <= 0)) return fail(LDIT_EINVAL, "fpn_merge: bad top size");
    const size_t total = (size_t)B * Oh * Ow * (Ch >> 2);
    if (total >= (1ull << 31)) return fail(LDIT_EUNSUPPORTED, "fpn_merge: output exceeds 2^31 vectors");
    hipLaunchKernelGGL(fpn_merge_nhwc, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, lat, top, out, B, Gh, Gw, Ch, Oh, Ow,
                       1.0f / scale, top_h, top_w);
    LDIT_HIP_CHECK(hipGetLastError());
    return LDIT_OK;
}

}  // namespace ldit

// ---- adjoint of tap_to_map (training through DiTBackbone.forward, ref dit_backbone.py:50-61) -----------------------------
// dtap[b, 1 + gy Gw + gx, c] = sum over the output pixels whose bilinear footprint contains token (gy, gx) of weight x
// dmap[b, c, oy, ox]; the CLS row receives zero (it is sliced away by the forward).  Gather form: every gradient element
// is written by exactly one thread, in a fixed order - no atomics.  The weights are recomputed with the forward's own index
// arithmetic, so clamping at the borders is the exact transpose of the forward.
namespace ldit {
namespace {

__device__ __forceinline__ float axis_weight(int o, int g, int G, float inv_scale)
{
    float s = ((float)o + 0.5f) * inv_scale - 0.5f;
    s = s < 0.f ? 0.f : s;
    int i0 = (int)s;
    i0 = i0 > G - 1 ? G - 1 : i0;
    const int i1 = i0 + (i0 < G - 1);
    const float l = s - (float)i0;
    return (i0 == g ? 1.f - l : 0.f) + (i1 == g ? l : 0.f);
}

__global__ void __launch_bounds__(256) tap_to_map_bwd(const float *__restrict__ dmap, float *__restrict__ dtap, int B, int Gh, int Gw,
                                                      int C, int Oh, int Ow, float scale)
{
    const size_t total = (size_t)B * C * Gh * Gw, idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int gx = (int)(idx % Gw);
    size_t t = idx / Gw;
    const int gy = (int)(t % Gh);
    t /= Gh;
    const int c = (int)(t % C), b = (int)(t / C);
    const float inv = 1.0f / scale;
    int ylo = (int)floorf(scale * (float)(gy - 1)) - 1, yhi = (int)ceilf(scale * (float)(gy + 2)) + 1;
    int xlo = (int)floorf(scale * (float)(gx - 1)) - 1, xhi = (int)ceilf(scale * (float)(gx + 2)) + 1;
    ylo = ylo < 0 ? 0 : ylo; xlo = xlo < 0 ? 0 : xlo;
    yhi = yhi > Oh - 1 ? Oh - 1 : yhi; xhi = xhi > Ow - 1 ? Ow - 1 : xhi;
    const float *plane = dmap + ((size_t)b * C + c) * Oh * Ow;
    float acc = 0.0f;
    for (int oy = ylo; oy <= yhi; ++oy) {
        const float wy = axis_weight(oy, gy, Gh, inv);
        if (wy == 0.0f) continue;
        float row = 0.0f;
        for (int ox = xlo; ox <= xhi; ++ox) row += axis_weight(ox, gx, Gw, inv) * plane[(size_t)oy * Ow + ox];
        acc += wy * row;
    }
    dtap[((size_t)b * (Gh * Gw + 1) + 1 + gy * Gw + gx) * C + c] = acc;
    if (gy == 0 && gx == 0) dtap[(size_t)b * (Gh * Gw + 1) * C + c] = 0.0f;      // CLS row
}

}  // namespace

int launch_tap_to_map_bwd(const float *dmap, float *dtap, int B, int Gh, int Gw, int C, float scale, hipStream_t stream)
{
    if (B <= 0 || Gh <= 0 || Gw <= 0 || C <= 0) return fail(LDIT_EINVAL, "tap_to_map_bwd: empty problem");
    if (!(scale == 4.0f || scale == 2.0f || scale == 1.0f || scale == 0.5f)) return fail(LDIT_EUNSUPPORTED, "tap_to_map_bwd: scale %g not in {4,2,1,0.5}", (double)scale);
    if (!dmap || !dtap) return fail(LDIT_EINVAL, "tap_to_map_bwd: null operand");
    const int Oh = (int)((float)Gh * scale), Ow = (int)((float)Gw * scale);
    if (Oh <= 0 || Ow <= 0) return fail(LDIT_EINVAL, "tap_to_map_bwd: map collapses to zero size");
    const size_t total = (size_t)B * C * Gh * Gw;
    hipLaunchKernelGGL(tap_to_map_bwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, dmap, dtap, B, Gh, Gw, C, Oh, Ow, scale);
    LDIT_HIP_CHECK(hipGetLastError());
    return LDIT_OK;
}

}  // namespace ldit
