// dma_cache.hip -- what the LDS-DMA path (global_load_lds_dwordx4, 1 KB per instruction) delivers per CU when the data is NOT
// streamed from HBM: 256 persistent workgroups of 8 waves; the workgroups of XCD x (blockIdx & 7) walk a region of F bytes of
// their own again and again, D instructions in flight per wave.  F <= ~3 MB stays in the XCD's 4 MB L2, 8 F <= 256 MB in the
// Infinity Cache, beyond that HBM.  "shared": every workgroup re-reads the same 18 KB (a chunk's packed weights).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void glds(const char* sbase, unsigned voff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds), "v"(voff), "s"(sbase) : "memory", "m0");
}

__device__ __forceinline__ void glds_v(const char* src, unsigned lds) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(lds), "v"(src) : "memory", "m0");
}
__device__ uint4 g_zero16 = {0, 0, 0, 0};

template <int D>
__global__ __launch_bounds__(512) void k_walk(const char* __restrict__ src, size_t region, int iters, int shared, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + wave * 16384;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per = gridDim.x >> 3;
    const char* base = src + (size_t)xcd * region;
    const size_t nblk = shared ? 1 : region / 65536;        // 64 KB blocks: 8 waves x 8 instructions
    size_t b = shared ? 0 : (size_t)j % nblk;
    for (int it = 0; it < iters; ++it) {
        const char* p = base + b * 65536;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned off = shared ? (unsigned)(((i * 8 + wave) % 18) * 1024 + lane * 16) : (unsigned)((i * 8 + wave) * 1024 + lane * 16);
            glds(p, off, lds0 + (i & 15) * 1024);
            if (D == 4) __builtin_amdgcn_s_waitcnt(3 | 0x0F70);
            if (D == 8) __builtin_amdgcn_s_waitcnt(7 | 0x0F70);
            if (D == 16) __builtin_amdgcn_s_waitcnt(15 | 0x0F70);
            if (D == 32) __builtin_amdgcn_s_waitcnt(15 | 0x0F70 | (1 << 14));
        }
        b += per;
        if (b >= nblk) b -= nblk * (b / nblk);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (smem[threadIdx.x] == 0x5a && iters == -1) out[0] = 1;
}

// The conv kernel's halo request: a (ROWS+2) x 34 pixel x 64 B tile of a [n][64][64] x 64 B chunk plane, slot s = instruction * 64 + lane ->
// pixel s / 4, 16-byte piece (s % 4) ^ ((hx >> 2) & 3) (the XOR swizzle of the consumers' reads), 8 producer waves; next to it 8 (or 4)
// consumer waves on the matrix pipe with the kernel's LDS diet.  Region = the planes of the XCD's images (L2-resident when small).
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
template <int ROWS, int EDGE>
__global__ __launch_bounds__(1024) void k_tile(const char* __restrict__ src, size_t region, int iters, int consumers, int lds_reads, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    volatile unsigned* flag = reinterpret_cast<volatile unsigned*>(smem + 3 * 40960);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0) *flag = 0;
    __syncthreads();
    constexpr int HH = ROWS + 2, NSLOT = HH * 34 * 4, NI = (NSLOT + 63) / 64, NIP = (NI + 7) / 8;
    if (wave < 8) {
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const char* base = src + (size_t)xcd * region;
        const int nplanes = (int)(region / (64 * 64 * 64));          // image-planes of 256 KB in the region
        unsigned voff[NIP];
        bool zero_lane[NIP];
        const char* zp = reinterpret_cast<const char*>(&g_zero16);
        const int ty = (j % (64 / ROWS)), tx = (j / (64 / ROWS)) & 1;
#pragma unroll
        for (int i = 0; i < NIP; ++i) {
            const unsigned sl = (i * 8 + wave) * 64 + lane;
            const unsigned hp = sl / 4, cp = sl % 4, hy = hp / 34, hx = hp % 34;
            int y = ty * ROWS - 1 + (int)hy, x = tx * 32 - 1 + (int)hx;
            zero_lane[i] = EDGE && (y < 0 || y > 63 || x < 0 || x > 63);   // EDGE: out-of-image lanes read the zero page through a per-lane 64-bit address
            y = y < 0 ? 0 : (y > 63 ? 63 : y); x = x < 0 ? 0 : (x > 63 ? 63 : x);
            voff[i] = (unsigned)((y * 64 + x) * 64) + ((cp ^ ((hx >> 2) & 3)) << 4);
        }
        int plane = j % nplanes;
        for (int it = 0; it < iters; ++it) {
            const char* p = base + (size_t)plane * (64 * 64 * 64);
#pragma unroll
            for (int i = 0; i < NIP; ++i)
                if (i * 8 + wave < NI) {
                    if (EDGE) glds_v(zero_lane[i] ? zp : p + voff[i], lds0 + (it % 3) * 40960 + (i * 8 + wave) * 1024);
                    else glds(p, voff[i], lds0 + (it % 3) * 40960 + (i * 8 + wave) * 1024);
                }
            // two stages in flight, like the kernel's three halo buffers
            if (NIP == 3) __builtin_amdgcn_s_waitcnt(3 | 0x0F70); else __builtin_amdgcn_s_waitcnt(5 | 0x0F70);
            plane = plane + 5 >= nplanes ? plane + 5 - nplanes : plane + 5;
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        if (lane == 0) atomicAdd((unsigned*)flag, 1u);
    } else if (wave < 8 + consumers) {
        unsigned h = threadIdx.x * 2654435761u + blockIdx.x; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        uint4 fa = make_uint4((h & 0x8fff8fffu) | 0x30003000u, (h * 3 & 0x8fff8fffu) | 0x30003000u, (h * 5 & 0x8fff8fffu) | 0x30003000u, (h * 7 & 0x8fff8fffu) | 0x30003000u);
        uint4 fb = make_uint4((h * 11 & 0x8fff8fffu) | 0x30003000u, (h * 13 & 0x8fff8fffu) | 0x30003000u, (h * 17 & 0x8fff8fffu) | 0x30003000u, (h * 19 & 0x8fff8fffu) | 0x30003000u);
        float16v acc[4];
        for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        const char* lpp = smem + (wave - 8) * 4096 + lane * 16;
        unsigned sink = 0;
        while (*flag < 8u) {
#pragma unroll
            for (int u = 0; u < 12; ++u) {
                // lds_reads per 6 MFMAs: 7 = the 16-row shape (4 rows + 3 weight fragments), 12 = the 8-row one (3 + 3 per 3)
                for (int q = lds_reads * u / 6; q < lds_reads * (u + 1) / 6; ++q) {
                    typedef unsigned u4 __attribute__((ext_vector_type(4)));
                    const u4 v = *reinterpret_cast<const volatile u4*>(lpp + (q & 7) * 1024);
                    sink ^= v[0];
                }
                acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, fa), __builtin_bit_cast(half8, fb), acc[u & 3], 0, 0, 0);
            }
        }
        float sacc = 0.f;
        for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) sacc += acc[t][r];
        if (sacc == 1.2345f || sink == 0x12345u) out[1] = 1;
    }
}

int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    const size_t bytes = (size_t)4 << 30;
    char* a; unsigned* out;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(a, 1, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](auto kern, const char* dname, size_t region, int shared) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        const int iters = 3000;
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(256), dim3(512), 131072, 0, a, region, iters, shared, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        const double moved = 256.0 * iters * 65536.0;
        if (shared) printf("shared 18 KB          %s in flight per wave  %8.3f ms  %6.1f GB/s per CU  %6.2f TB/s chip\n", dname, ms, moved / 256 / ms * 1e-6, moved / ms * 1e-9);
        else printf("region %7.2f MB/XCD  %s in flight per wave  %8.3f ms  %6.1f GB/s per CU  %6.2f TB/s chip\n", region / 1048576.0, dname, ms, moved / 256 / ms * 1e-6, moved / ms * 1e-9);
    };
    const size_t regions[] = {(size_t)2 << 20, (size_t)3 << 20, (size_t)6 << 20, (size_t)24 << 20, (size_t)512 << 20};
    for (size_t r : regions) {
        run(k_walk<4>, " 4", r, 0);
        run(k_walk<8>, " 8", r, 0);
        run(k_walk<16>, "16", r, 0);
        run(k_walk<32>, "32", r, 0);
    }
    auto tile = [&](auto kern, int rows, size_t region, int consumers, int lds_reads) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 40960 + 64));
        const int iters = 20000;
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(256), dim3(64 * (8 + consumers)), 3 * 40960 + 64, 0, a, region, iters, consumers, lds_reads, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        const double tileb = (rows + 2) * 34 * 64.0, moved = 256.0 * iters * tileb;
        printf("halo tiles %2d+2 rows, region %6.2f MB/XCD, %d consumer waves (%2d LDS reads / 6 MFMAs)  %8.3f ms  %6.1f GB/s per CU  %6.2f TB/s chip  %5.2f us per tile\n",
               rows, region / 1048576.0, consumers, lds_reads, ms, moved / 256 / ms * 1e-6, moved / ms * 1e-9, ms * 1e3 / iters);
    };
    for (size_t r : {(size_t)2 << 20}) {
        tile(k_tile<8, 0>, 8, r, 0, 0);
        tile(k_tile<8, 0>, 8, r, 8, 12);
        printf("  (out-of-image lanes -> zero page, per-lane 64-bit addresses:)\n");
        tile(k_tile<8, 1>, 8, r, 0, 0);
        tile(k_tile<8, 1>, 8, r, 8, 12);
        tile(k_tile<16, 0>, 16, r, 8, 7);
        tile(k_tile<16, 1>, 16, r, 8, 7);
    }
    run(k_walk<4>, " 4", 65536, 1);
    run(k_walk<16>, "16", 65536, 1);
    return 0;
}
