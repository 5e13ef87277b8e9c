// Microbenchmark: sustained fp32 MFMA rate and in-kernel clock for the two gfx950 fp32 shapes on random operands.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_clock.hip -o gpurun_out/mfma_clock && ./gpurun_out/mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ void __launch_bounds__(256) loop(const float *in, float *out, int iters, unsigned long long *clk)
{
    const int lane = threadIdx.x & 63;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(threadIdx.x * 8 + i) & 4095]; b[i] = in[(threadIdx.x * 8 + i + 2048) & 4095]; }
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[8];
        for (int t = 0; t < 8; ++t) for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(k + t) & 7], b[k], acc[t], 0, 0, 0);
        }
        for (int t = 0; t < 8; ++t) for (int e = 0; e < 16; ++e) s += acc[t][e];
    } else {
        f32x4 acc[32];
        for (int t = 0; t < 32; ++t) for (int e = 0; e < 4; ++e) acc[t][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int t = 0; t < 32; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(k + t) & 7], b[k + (t & 4)], acc[t], 0, 0, 0);
        }
        for (int t = 0; t < 32; ++t) for (int e = 0; e < 4; ++e) s += acc[t][e];
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main()
{
    const int blocks = 256 * 1, iters = 40000;
    std::vector<float> h(4096);
    for (auto &v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *in, *out; unsigned long long *clk;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
    hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    for (int shape : {32, 16, 32, 16}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        // warm
        if (shape == 32) hipLaunchKernelGGL(loop<32>, dim3(blocks), dim3(256), 0, 0, in, out, 1000, clk);
        else hipLaunchKernelGGL(loop<16>, dim3(blocks), dim3(256), 0, 0, in, out, 1000, clk);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        if (shape == 32) hipLaunchKernelGGL(loop<32>, dim3(blocks), dim3(256), 0, 0, in, out, iters, clk);
        else hipLaunchKernelGGL(loop<16>, dim3(blocks), dim3(256), 0, 0, in, out, iters, clk);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> c(blocks * 2);
        hipMemcpy(c.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
        double flops = (shape == 32 ? 64.0 * 4096 : 128.0 * 2048) * iters * 4.0 * blocks;  // MFMAs/iter * flop * waves
        double ghz = (double)c[0] / ((double)c[1] / 100.0) / 1e3;
        printf("shape %dx%d: %.2f ms  %.1f TFLOP/s  in-kernel clock %.3f GHz\n", shape, shape, ms, flops / ms / 1e9, ghz);
    }
    return 0;
}
