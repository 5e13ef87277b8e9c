// Microbenchmark: what an LDS-DMA piece (global_load_lds_dwordx4, 1 KB per wave instruction) costs the MFMA stream that issues it.
// One workgroup per CU; every wave runs blocks of 16 independent v_mfma_f32_32x32x16_bf16 with D DMA pieces dealt out between
// them (source: an L2-resident buffer), D = 0, 1, 2, 4, 8, 16; reported: cycles per MFMA seen by wave 0.
//   hipcc --offload-arch=gfx950 -O3 -o layoutdit_amd/csrc/build/dma_issue scripts/ubench/dma_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int D, int NW, int WAITEVERY = 4, bool ASYM = false>
__global__ void __launch_bounds__(64 * NW) k(const float *src, unsigned long long *out, float *sink, int iters)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c)
        for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane - e); }
    const float *g = src + ((size_t)blockIdx.x * 4096 + wave * 2048 + lane * 4) % (1 << 20);
    char *dst = smem + wave * 16384;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
            if (D > 0 && (m % (16 / (D > 16 ? 16 : D))) == 0 && (!ASYM || wave < NW / 2)) {
#pragma unroll
                for (int r = 0; r < (D > 16 ? D / 16 : 1); ++r)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + ((it * 16 + m) & 63) * 256),
                                                     (__attribute__((address_space(3))) void *)(dst + ((m + r) & 15) * 1024), 16, 0, 0);
            }
        }
        if (D > 0 && (it % WAITEVERY) == WAITEVERY - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                   // wave 0 times the whole workgroup
    float s = 0;
    for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][15];
    asm volatile("s_nop 0" ::"v"(s));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    sink[blockIdx.x * 64 * NW + threadIdx.x] = s + smem[threadIdx.x];
}

template <int D, int NW, int WAITEVERY = 4, bool ASYM = false>
void run(const float *src, unsigned long long *out, float *sink, const char *what)
{
    const int iters = 256;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<D, NW, WAITEVERY, ASYM>), hipFuncAttributeMaxDynamicSharedMemorySize, NW * 16384);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<D, NW, WAITEVERY, ASYM>), dim3(256), dim3(64 * NW), NW * 16384, 0, src, out, sink, iters);
    hipDeviceSynchronize();
    unsigned long long h;
    hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
    printf("%-28s D = %2d pieces per 16 MFMAs%s, vmcnt(0) every %2d blocks: %6.1f cycles per MFMA\n", what, D, ASYM ? " on waves 0..NW/2-1 only" : "", WAITEVERY, (double)h / (iters * 16));
}

int main()
{
    float *src, *sink;
    unsigned long long *out;
    hipMalloc(&src, 8 << 20);          // offsets below stay under 1 Mi + 16 Ki + 256 floats: inside the first 5 MB
    hipMemset(src, 0, 8 << 20);
    hipMalloc(&sink, 256 * 512 * 4);
    hipMalloc(&out, 8);
    run<0, 4>(src, out, sink, "4 waves (1 per SIMD)"); run<1, 4>(src, out, sink, "4 waves (1 per SIMD)"); run<2, 4>(src, out, sink, "4 waves (1 per SIMD)");
    run<4, 4>(src, out, sink, "4 waves (1 per SIMD)"); run<8, 4>(src, out, sink, "4 waves (1 per SIMD)"); run<16, 4>(src, out, sink, "4 waves (1 per SIMD)");
    run<0, 8>(src, out, sink, "8 waves (2 per SIMD)"); run<2, 8>(src, out, sink, "8 waves (2 per SIMD)"); run<4, 8>(src, out, sink, "8 waves (2 per SIMD)");
    run<8, 8>(src, out, sink, "8 waves (2 per SIMD)");
    run<3, 8>(src, out, sink, "8 waves (2 per SIMD)");
    run<4, 8, 1>(src, out, sink, "8 waves (2 per SIMD)"); run<4, 8, 2>(src, out, sink, "8 waves (2 per SIMD)"); run<4, 8, 16>(src, out, sink, "8 waves (2 per SIMD)");
    run<2, 8, 1>(src, out, sink, "8 waves (2 per SIMD)"); run<8, 4, 1>(src, out, sink, "4 waves (1 per SIMD)"); run<8, 4, 16>(src, out, sink, "4 waves (1 per SIMD)");
    run<8, 8, 4, true>(src, out, sink, "8 waves (2 per SIMD)"); run<16, 8, 4, true>(src, out, sink, "8 waves (2 per SIMD)"); run<4, 8, 4, true>(src, out, sink, "8 waves (2 per SIMD)");
    return 0;
}
