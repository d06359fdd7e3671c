// mfma_peak.hip -- what the gfx950 matrix pipe sustains on this box (clock under load included):
//   mode 0: MFMA only, 9 independent 32x32x16 f16 accumulators per wave
//   mode 1: the same + the quad wgrad kernel's LDS transpose reads (88 ds_read_b64_tr_b16 per 72 MFMAs)
//   mode 2: the same as 0 with only `waves` waves per SIMD resident (occupancy sensitivity)
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip ; run: ./mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 fp16x4_t;

__device__ __forceinline__ uint2 tr_read(const char* lds_addr) {
    auto p = reinterpret_cast<__attribute__((address_space(3))) fp16x4_t*>((__attribute__((address_space(3))) char*)lds_addr);
    const fp16x4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4f16(p);
    return __builtin_bit_cast(uint2, r);
}

template <int MODE>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 40960; i += blockDim.x) {
        unsigned h = i * 2654435761u; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        reinterpret_cast<unsigned*>(smem)[i] = (MODE & 16) ? ((h & 0x8fff8fffu) | 0x30003000u) : __float_as_uint(0.001f * (i & 255));
    }
    __syncthreads();
    float16v acc[9];
    for (int t = 0; t < 9; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    uint4 fa = make_uint4(lane, 1, 2, 3), fb[3] = {make_uint4(4, lane, 6, 7), make_uint4(1, 2, lane, 4), make_uint4(9, 8, 7, lane)};
    if (MODE & 16) {   // operands with realistic bit activity: f16 values in (-2, 2) from a hash
        auto rnd = [&](unsigned k) { unsigned h = (threadIdx.x * 2654435761u) ^ (k * 40503u + blockIdx.x); h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
                                     return (h & 0x8fff8fffu) | 0x30003000u; };
        fa = make_uint4(rnd(1), rnd(2), rnd(3), rnd(4));
        for (int d = 0; d < 3; ++d) fb[d] = make_uint4(rnd(5 + d), rnd(9 + d), rnd(13 + d), rnd(17 + d));
    }
    // the wgrad kernels' lane map: 8 pixels x 64 B per half-wave, conflict free
    const int a16 = lane & 15;
    const char* base = smem + (((lane >> 5) << 3) + (a16 >> 2)) * 64 + ((((lane >> 4) & 1) << 4) + ((a16 & 3) << 2)) * 2;
    unsigned sink = 0;
#define RD(x) ((MODE & 8) ? *reinterpret_cast<const uint2*>(x) : tr_read(x))
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (MODE & 1) {
                asm volatile("" ::: "memory");
                const char* p = base + s * 2176;
                const uint2 a0 = RD(p), a1 = RD(p + 256);
                if (MODE & 2) sink ^= a0.x ^ a1.y; else fa = make_uint4(a0.x, a0.y, a1.x, a1.y);
#pragma unroll
                for (int d = 0; d < ((MODE & 4) ? 1 : 3); ++d) {
                    const uint2 l = RD(p + 4096 + d * 64), h = RD(p + 4096 + d * 64 + 256);
                    if (MODE & 2) sink ^= l.x ^ h.y; else fb[d] = make_uint4(l.x, l.y, h.x, h.y);
                }
                if (s < 3 && !(MODE & 4)) {   // 11 reads per 9 MFMA ~ 88 per 72
                    const uint2 l = RD(p + 8192), h = RD(p + 8192 + 256), m = RD(p + 8192 + 512);
                    sink ^= l.x & h.y & m.x & 1;
                }
            }
#pragma unroll
            for (int t = 0; t < 9; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, fa), __builtin_bit_cast(half8, fb[t % 3]), acc[t], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int t = 0; t < 9; ++t)
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)sink;
}

template <int MODE>
static void run(const char* name, int threads, int blocks, int iters) {
    float* out;
    hipMalloc(&out, (size_t)blocks * threads * 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 160 * 1024, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = 2.0 * 32 * 32 * 16 * 72.0 * iters * (threads / 64) * blocks;
        printf("%-28s threads %4d blocks %4d iters %6d  %8.3f ms  %8.1f TFLOP/s\n", name, threads, blocks, iters, ms, flop / ms * 1e-9);
    }
    hipFree(out);
}

int main() {
    run<0>("mfma only, 2 waves/SIMD", 512, 256, 4000);
    run<0>("mfma only, 1 wave/SIMD", 256, 256, 4000);
    run<1>("mfma + tr reads, 2 w/SIMD", 512, 256, 4000);
    run<1>("mfma + tr reads, 1 w/SIMD", 256, 256, 4000);
    run<16>("mfma only, random data, 2 w", 512, 256, 4000);
    run<16>("mfma only, random, long", 512, 256, 40000);
    run<17>("mfma + tr, random data, 2 w", 512, 256, 4000);
    run<3>("reads not consumed, 2 w", 512, 256, 4000);
    run<5>("4 reads per 9 mfma, 2 w", 512, 256, 4000);
    run<9>("plain b64 reads, 2 w", 512, 256, 4000);
    return 0;
}
