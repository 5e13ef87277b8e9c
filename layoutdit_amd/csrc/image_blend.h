// The detector's input transform (ref src/layoutdit/modeling/model.py:50-54: torchvision GeneralizedRCNNTransform with fixed_size,
// image_mean = image_std = 0.5) as ONE arithmetic statement shared by every kernel that evaluates it - the stand-alone batch
// producer (misc.hip: preprocess_images) and the fused producer of the bf16 patch matrix (train_ops.hip: patches_rows_images) - so
// that both give a pixel the same bits: bilinear resize (align_corners = False, no antialias: F.interpolate(size=...)) of the
// [0, 1] image, then (v - mean) / std.  Every multiply-add is spelled out (no contraction left to the compiler).
#pragma once
#include "ldit_common.h"

namespace ldit {

constexpr int PRE_MAX = 48;          // images per launch: their descriptors travel in the kernel arguments (no device-side table)
struct ImageList {
    const void *img[PRE_MAX];        // device pointers, [in_ch, h, w] planar, fp32 or fp16
    int h[PRE_MAX], w[PRE_MAX];
    int n, first;                    // images in this launch; index of the first one in the batch
    float mean, inv_std;
};

struct BlendRow {                    // the two source rows of one output row and their weights
    int y0, y1;
    float ly, hy;
};

__device__ __forceinline__ BlendRow blend_row(int oy, int h, int out_h)
{
    const float sch = (float)h / (float)out_h;
    float sy = __builtin_fmaf((float)oy + 0.5f, sch, -0.5f);
    sy = sy < 0.f ? 0.f : sy;
    int y0 = (int)sy;
    y0 = y0 > h - 1 ? h - 1 : y0;
    BlendRow r;
    r.y0 = y0; r.y1 = y0 + (y0 < h - 1);
    r.ly = sy - (float)y0; r.hy = 1.f - r.ly;
    return r;
}

// output pixel ox of a row whose source rows are r0 / r1 (w pixels each), normalised
template <typename IN>
__device__ __forceinline__ float blend_pixel(const IN *__restrict__ r0, const IN *__restrict__ r1, int ox, int w, int out_w,
                                             const BlendRow &br, float mean, float inv_std)
{
    const float scw = (float)w / (float)out_w;
    float sx = __builtin_fmaf((float)ox + 0.5f, scw, -0.5f);
    sx = sx < 0.f ? 0.f : sx;
    int x0 = (int)sx;
    x0 = x0 > w - 1 ? w - 1 : x0;
    const int x1 = x0 + (x0 < w - 1);
    const float lx = sx - (float)x0, hx = 1.f - lx;
    const float top = __builtin_fmaf(lx, (float)r0[x1], hx * (float)r0[x0]);
    const float bot = __builtin_fmaf(lx, (float)r1[x1], hx * (float)r1[x0]);
    const float v = __builtin_fmaf(br.ly, bot, br.hy * top);
    return (v - mean) * inv_std;
}

}  // namespace ldit
