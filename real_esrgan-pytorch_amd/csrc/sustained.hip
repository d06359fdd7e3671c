// sustained.hip -- what the memory pipe and the matrix pipe deliver TOGETHER on THIS board once it sits at its power cap
// (measurement aid behind resr_debug_sustained, include/resr_debug.h; the standalone form is tools/micro/sustained.hip).
// bench.py prices its dominant kernel against this frontier (`roofline.vs_sustained`); the box-to-box spread of one build is
// +-5 %, so the frontier has to come from the box the bench runs on, in the same process.
//   mode 1  LDS-DMA stream, 4 waves per workgroup
//   mode 2  8 waves per workgroup issuing v_mfma_f32_32x32x16_f16 on random f16 operands
//   mode 3  both at once
// Launched back to back for `seconds`; the last third is timed with events.
#include "common.h"

namespace resr {

namespace {

__device__ __forceinline__ void sus_glds(const char* sbase, unsigned voff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds), "v"(voff), "s"(sbase) : "memory", "m0");
}

// MFMA waves run until the streaming waves are done (mode 3) or for `iters` rounds of 16 (mode 2)
__global__ __launch_bounds__(768) void sustained_kernel(const char* __restrict__ src, size_t bytes, unsigned long long* mfma_count, int mode, int iters) {
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
                for (int i = 0; i < 16; ++i) sus_glds(base, (unsigned)((i * 4 + wave) * 1024 + lane * 16), lds0 + (i & 7) * 1024);
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

}  // namespace

// src: `bytes` (a multiple of 64 KB, >= 64 MB) of readable device memory; counter: 8 bytes of device memory.  Synchronises.
int sustained_run(int mode, double seconds, const void* src, size_t bytes, void* counter, double* tbs, double* pflops, hipStream_t st) {
    if (mode < 1 || mode > 3 || !src || !counter || bytes < (64u << 20) || seconds <= 0 || seconds > 30) return fail(RESR_ERR_ARG, "sustained: bad argument");
    bytes &= ~(size_t)65535;
    const size_t lds = 65536 + 64;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&sustained_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return fail(RESR_ERR_LAUNCH, "sustained: hipFuncSetAttribute");
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fail(RESR_ERR_LAUNCH, "sustained: hipEventCreate");
    auto launch = [&]() { hipLaunchKernelGGL(sustained_kernel, dim3(256), dim3(768), lds, st, (const char*)src, bytes, (unsigned long long*)counter, mode, 2000); };
    // calibrate the launch time, then warm for two thirds of `seconds` and time the last third
    float ms = 0.f;
    hipEventRecord(e0, st);
    for (int i = 0; i < 4; ++i) launch();
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    const double per = ms / 4.0 > 1e-3 ? ms / 4.0 : 1e-3;
    const int warm = (int)(seconds * 1e3 * 2.0 / 3.0 / per) + 1, timed = (int)(seconds * 1e3 / 3.0 / per) + 1;
    for (int i = 0; i < warm; ++i) launch();
    hipMemsetAsync(counter, 0, 8, st);
    hipEventRecord(e0, st);
    for (int i = 0; i < timed; ++i) launch();
    hipEventRecord(e1, st);
    if (hipEventSynchronize(e1) != hipSuccess) return fail(RESR_ERR_LAUNCH, "sustained: hipEventSynchronize");
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long n = 0;
    hipMemcpy(&n, counter, 8, hipMemcpyDeviceToHost);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    if (tbs) *tbs = (mode & 1) ? (double)bytes * timed / ms * 1e-9 : 0.0;
    if (pflops) *pflops = (double)n * 2.0 * 32 * 32 * 16 / (ms * 1e-3) * 1e-15;
    RESR_CHECK_LAUNCH("sustained_kernel");
    return RESR_OK;
}

}  // namespace resr
