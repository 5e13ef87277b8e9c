// Row LayerNorm (nn.LayerNorm over the last axis; TF:models/beit/modeling_beit.py:390-391,426,438, eps = 1e-12).
//
// HBM-bound: one 64-lane wave owns one row, holds it in registers (float4 per lane per 1 KiB of row), and makes the
// two reductions (mean, then variance about the mean - never E[x^2]-E[x]^2, which cancels on BEiT's large-magnitude
// channels) with DPP/permute wave reductions; no LDS, no second read of the row.  4 rows per 256-thread block.
#include "ldit_common.h"

namespace ldit {

namespace {

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// VPL = float4 vectors per lane; handles C <= 256 * VPL
typedef __bf16 ln_bf16x4 __attribute__((ext_vector_type(4)));

// OUT = 1: the normalised row is rounded to bf16 (operand of the bf16 GEMMs); OUT = 2: quantised to fp8 e4m3 codes of
// y / *qscale (operand of the fp8 GEMMs, saturating); OUT = 3 / 4: written as 2 / 3 bf16 planes y ~= p0 + p1 (+ p2) side by
// side in a row of 2 C / 3 C elements (operand of the split-fp32 GEMMs); statistics and affine stay fp32.
template <int VPL, int OUT, int RPW>
__global__ void __launch_bounds__(256) layernorm_rows(const float *__restrict__ X, const float *__restrict__ g,
                                                      const float *__restrict__ b, void *__restrict__ Yv, int64_t rows,
                                                      int C, float eps, const float *__restrict__ qscale)
{
    // RPW rows per wave, their loads all issued before the first reduction (round 3: with one 4-KB row per wave the kernel
    // sat at 4.8 TB/s - four 16-B loads in flight per lane; two rows double the bytes in flight per CU)
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= rows) return;
    const int nvec = C >> 2;
    f32x4 v[RPW][VPL];
    float s[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int64_t row = row0 + r < rows ? row0 + r : rows - 1;          // a ragged last wave re-reads its last row
        const f32x4 *x4 = reinterpret_cast<const f32x4 *>(X + row * C);
#pragma unroll
        for (int u = 0; u < VPL; ++u) {
            const int idx = lane + 64 * u;
            v[r][u] = idx < nvec ? x4[idx] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const float inv_c = 1.0f / (float)C;
    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(g);
    const f32x4 *b4 = reinterpret_cast<const f32x4 *>(b);
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        s[r] = 0.0f;
#pragma unroll
        for (int u = 0; u < VPL; ++u) s[r] += (v[r][u][0] + v[r][u][1]) + (v[r][u][2] + v[r][u][3]);
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int64_t row = row0 + r;
        if (row >= rows) break;                                             // wave-uniform
        const float mu = wave_sum(s[r]) * inv_c;
        float q = 0.0f;
#pragma unroll
        for (int u = 0; u < VPL; ++u) {
            const int idx = lane + 64 * u;
            if (idx < nvec) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = v[r][u][e] - mu;
                    v[r][u][e] = d;
                    q += d * d;
                }
            }
        }
        const float var = wave_sum(q) * inv_c;
        const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
        for (int u = 0; u < VPL; ++u) {
            const int idx = lane + 64 * u;
            if (idx < nvec) {
                const f32x4 gg = g4[idx], bb = b4[idx];
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = v[r][u][e] * rstd * gg[e] + bb[e];
                if (OUT >= 3) {
                    constexpr int S = OUT == 3 ? 2 : 3;
                    __bf16 *yr = static_cast<__bf16 *>(Yv) + row * (int64_t)(S * C);
#pragma unroll
                    for (int sp = 0; sp < S; ++sp) {
                        const ln_bf16x4 pk = {(__bf16)o[0], (__bf16)o[1], (__bf16)o[2], (__bf16)o[3]};
                        reinterpret_cast<ln_bf16x4 *>(yr + sp * C)[idx] = pk;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] -= (float)pk[e];
                    }
                } else if (OUT == 2) {
                    const float qi = 1.0f / qscale[0];
                    reinterpret_cast<unsigned *>(static_cast<unsigned char *>(Yv) + row * C)[idx] =
                        pack_fp8x4(o[0] * qi, o[1] * qi, o[2] * qi, o[3] * qi);
                } else if (OUT == 1) {
                    const ln_bf16x4 pk = {(__bf16)o[0], (__bf16)o[1], (__bf16)o[2], (__bf16)o[3]};
                    reinterpret_cast<ln_bf16x4 *>(static_cast<__bf16 *>(Yv) + row * C)[idx] = pk;
                } else {
                    reinterpret_cast<f32x4 *>(static_cast<float *>(Yv) + row * C)[idx] = o;
                }
            }
        }
    }
}

}  // namespace

template <int OUT>
static int launch_ln(const float *X, const float *g, const float *b, void *Y, int64_t rows, int C, float eps,
                     const float *qscale, hipStream_t stream)
{
    if (rows <= 0 || C <= 0) return fail(LDIT_EINVAL, "layernorm: empty problem");
    if (!X || !g || !b || !Y) return fail(LDIT_EINVAL, "layernorm: null operand");
    if (C & 3) return fail(LDIT_EUNSUPPORTED, "layernorm: C=%d must be a multiple of 4", C);
    if (C > 4096) return fail(LDIT_EUNSUPPORTED, "layernorm: C=%d exceeds 4096", C);
    if (!aligned16(X) || (reinterpret_cast<uintptr_t>(Y) & (OUT == 2 ? 3u : 7u)) || !aligned16(g) || !aligned16(b)) return fail(LDIT_EINVAL, "layernorm: operands must be 16-byte aligned");
    const dim3 block(256);
    const dim3 grid1((unsigned)((rows + 3) / 4)), grid2((unsigned)((rows + 7) / 8));
    // two rows per wave once there are enough rows to fill the machine twice over (serving sizes keep one row per wave)
    const bool two = rows >= 2 * 256 * 16;
    if (C <= 256) hipLaunchKernelGGL((layernorm_rows<1, OUT, 1>), grid1, block, 0, stream, X, g, b, Y, rows, C, eps, qscale);
    else if (C <= 1024 && two) hipLaunchKernelGGL((layernorm_rows<4, OUT, 2>), grid2, block, 0, stream, X, g, b, Y, rows, C, eps, qscale);
    else if (C <= 1024) hipLaunchKernelGGL((layernorm_rows<4, OUT, 1>), grid1, block, 0, stream, X, g, b, Y, rows, C, eps, qscale);
    else hipLaunchKernelGGL((layernorm_rows<16, OUT, 1>), grid1, block, 0, stream, X, g, b, Y, rows, C, eps, qscale);
    LDIT_HIP_CHECK(hipGetLastError());
    return LDIT_OK;
}

int launch_layernorm(const float *X, const float *g, const float *b, float *Y, int64_t rows, int C, float eps,
                     hipStream_t stream)
{
    return launch_ln<0>(X, g, b, Y, rows, C, eps, nullptr, stream);
}

int launch_layernorm_bf16out(const float *X, const float *g, const float *b, void *Y, int64_t rows, int C, float eps,
                             hipStream_t stream)
{
    return launch_ln<1>(X, g, b, Y, rows, C, eps, nullptr, stream);
}

int launch_layernorm_splitout(const float *X, const float *g, const float *b, void *Y, int64_t rows, int C, float eps, int planes,
                              hipStream_t stream)
{
    if (planes == 2) return launch_ln<3>(X, g, b, Y, rows, C, eps, nullptr, stream);
    if (planes == 3) return launch_ln<4>(X, g, b, Y, rows, C, eps, nullptr, stream);
    return fail(LDIT_EINVAL, "layernorm: %d output planes (2 or 3)", planes);
}

int launch_layernorm_fp8out(const float *X, const float *g, const float *b, void *Y, int64_t rows, int C, float eps,
                            const float *qscale, hipStream_t stream)
{
    if (!qscale) return fail(LDIT_EINVAL, "layernorm: fp8 output needs a scale");
    return launch_ln<2>(X, g, b, Y, rows, C, eps, qscale, stream);
}

}  // namespace ldit
