// sustained.hip -- what the memory pipe and the matrix pipe deliver TOGETHER once the chip sits at its power cap.
// hbm_rate.hip / mfma_peak.hip time sub-millisecond bursts; a training step holds its load for seconds.  Here every arm
// is launched back to back for ~1.5 s and only the last third is timed:
//   dma        LDS-DMA streaming, 4 waves per workgroup, 8 idle waves
//   dma+mfma   the same while 8 waves per workgroup issue v_mfma_f32_32x32x16_f16 on random f16 operands
//   mfma       the MFMA waves alone (the streaming waves exit at once)
// Reported: TB/s of the stream and executed PFLOP/s of the matrix waves in the timed window.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void glds(const char* sbase, unsigned voff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds), "v"(voff), "s"(sbase) : "memory", "m0");
}

// mode bit 0: stream, bit 1: MFMA.  MFMA waves run until the streaming waves are done (or `iters` rounds without a stream)
__global__ __launch_bounds__(768) void k_mix(const char* __restrict__ src, size_t bytes, unsigned long long* mfma_count, int mode, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    volatile unsigned* flag = reinterpret_cast<volatile unsigned*>(smem + 65536);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0) *flag = 0;
    __syncthreads();
    if (wave < 4) {
        if (mode & 1) {
            const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + wave * 16384;
            const size_t blk = 64 * 1024, nblk = bytes / blk;
            for (size_t b = blockIdx.x; b < nblk; b += gridDim.x) {
                const char* base = src + b * blk;
#pragma unroll
                for (int i = 0; i < 16; ++i) glds(base, (unsigned)((i * 4 + wave) * 1024 + lane * 16), lds0 + (i & 7) * 1024);
                __builtin_amdgcn_s_waitcnt(0x0F70);
            }
        }
        if (lane == 0) atomicAdd((unsigned*)flag, 1u);
    } else if (mode & 2) {
        unsigned h = threadIdx.x * 2654435761u + blockIdx.x; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        uint4 fa = make_uint4((h & 0x8fff8fffu) | 0x30003000u, (h * 3 & 0x8fff8fffu) | 0x30003000u, (h * 5 & 0x8fff8fffu) | 0x30003000u, (h * 7 & 0x8fff8fffu) | 0x30003000u);
        uint4 fb = make_uint4((h * 11 & 0x8fff8fffu) | 0x30003000u, (h * 13 & 0x8fff8fffu) | 0x30003000u, (h * 17 & 0x8fff8fffu) | 0x30003000u, (h * 19 & 0x8fff8fffu) | 0x30003000u);
        float16v acc[4];
        for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        unsigned rounds = 0;
        while ((mode & 1) ? (*flag < 4u) : (rounds < (unsigned)iters)) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, fa), __builtin_bit_cast(half8, fb), acc[u & 3], 0, 0, 0);
            ++rounds;
        }
        float s = 0.f;
        for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
        if (lane == 0) atomicAdd(mfma_count, (unsigned long long)rounds * 16ull + (s == 1.2345f ? 1ull : 0ull));
    }
}

int main() {
    const size_t bytes = (size_t)2 << 30;
    char* a; unsigned long long* cnt;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&cnt, 8)); CK(hipMemset(a, 1, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mix), hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 64));
    struct Arm { const char* name; int mode; } arms[] = {{"dma (8 idle waves)", 1}, {"dma + 8 MFMA waves (random f16)", 3}, {"8 MFMA waves alone", 2}};
    for (const Arm& arm : arms) {
        for (int burst : {1, 0}) {
            // one launch of the stream arms moves 2 GB (~0.35 ms at 6 TB/s); the MFMA-only arm runs 2000 rounds of 16 per wave
            const int warm = burst ? 2 : 3000, timed = burst ? 1 : 1500;
            for (int i = 0; i < warm; ++i) hipLaunchKernelGGL(k_mix, dim3(256), dim3(768), 65536 + 64, 0, a, bytes, cnt, arm.mode, 2000);
            CK(hipMemsetAsync(cnt, 0, 8, 0));
            CK(hipEventRecord(e0));
            for (int i = 0; i < timed; ++i) hipLaunchKernelGGL(k_mix, dim3(256), dim3(768), 65536 + 64, 0, a, bytes, cnt, arm.mode, 2000);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long n = 0; CK(hipMemcpy(&n, cnt, 8, hipMemcpyDeviceToHost));
            const double tbs = (arm.mode & 1) ? (double)bytes * timed / ms * 1e-9 : 0.0;
            const double pf = (double)n * 2.0 * 32 * 32 * 16 / (ms * 1e-3) * 1e-15;
            printf("%-36s %-9s window %8.1f ms   stream %6.2f TB/s   matrix %6.3f PFLOP/s\n", arm.name, burst ? "burst" : "sustained", ms, tbs, pf);
            CK(hipDeviceSynchronize());
        }
    }
    return 0;
}
