// Microbenchmark: cycles per MFMA in a dependent accumulator chain vs independent chains, fp32-input MFMAs (one wave).
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma_chain scripts/ubench/mfma_chain.hip && gpurun_out/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ void k16(float a, float b, unsigned long long *out, float *sink, int waves_per_simd)
{
    f32x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0, 0, 0, 0};
    float av = a + threadIdx.x, bv = b - threadIdx.x;
    __builtin_amdgcn_s_barrier();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < 256; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][3];
    asm volatile("s_nop 0" ::"v"(s));
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = s;
}

template <int CHAINS>
__global__ void k32(float a, float b, unsigned long long *out, float *sink)
{
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int e = 0; e < 16; ++e) acc[c][e] = 0;
    float av = a + threadIdx.x, bv = b - threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < 256; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][15];
    asm volatile("s_nop 0" ::"v"(s));
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = s;
}

int main()
{
    unsigned long long *out, h;
    float *sink;
    hipMalloc(&out, 8);
    hipMalloc(&sink, 4096);
    auto report = [&](const char *name, int n) {
        hipDeviceSynchronize();
        hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
        printf("%-44s %8.1f cycles per MFMA\n", name, (double)h / n);
    };
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k16<1>, dim3(1), dim3(64), 0, 0, 1.0f, 2.0f, out, sink, 1); report("16x16x4 f32, 1 chain, 1 wave", 2048);
        hipLaunchKernelGGL(k16<2>, dim3(1), dim3(64), 0, 0, 1.0f, 2.0f, out, sink, 1); report("16x16x4 f32, 2 chains, 1 wave", 4096);
        hipLaunchKernelGGL(k16<4>, dim3(1), dim3(64), 0, 0, 1.0f, 2.0f, out, sink, 1); report("16x16x4 f32, 4 chains, 1 wave", 8192);
        hipLaunchKernelGGL(k16<1>, dim3(1), dim3(256), 0, 0, 1.0f, 2.0f, out, sink, 1); report("16x16x4 f32, 1 chain, 4 waves (1/SIMD)", 2048);
        hipLaunchKernelGGL(k16<1>, dim3(1), dim3(512), 0, 0, 1.0f, 2.0f, out, sink, 1); report("16x16x4 f32, 1 chain, 8 waves (2/SIMD)", 2048);
        hipLaunchKernelGGL(k16<1>, dim3(1), dim3(1024), 0, 0, 1.0f, 2.0f, out, sink, 1); report("16x16x4 f32, 1 chain, 16 waves (4/SIMD)", 2048);
        hipLaunchKernelGGL(k32<1>, dim3(1), dim3(64), 0, 0, 1.0f, 2.0f, out, sink); report("32x32x2 f32, 1 chain, 1 wave", 2048);
        hipLaunchKernelGGL(k32<2>, dim3(1), dim3(64), 0, 0, 1.0f, 2.0f, out, sink); report("32x32x2 f32, 2 chains, 1 wave", 4096);
        hipLaunchKernelGGL(k32<1>, dim3(1), dim3(256), 0, 0, 1.0f, 2.0f, out, sink); report("32x32x2 f32, 1 chain, 4 waves (1/SIMD)", 2048);
    }
    return 0;
}
