// hbm_rate.hip -- achievable HBM bandwidth on this box for the access shapes the conv kernels use.
//   read:   every workgroup streams disjoint 64 KB blocks with global_load_dwordx4 (sum to defeat DCE)
//   dma:    the same through LDS-DMA (global_load_lds_dwordx4), 1 KB per instruction, 8 in flight per wave
//   tile:   LDS-DMA of 34-row x 2176-byte segments with a row pitch of 16 KB (a halo tile of a 256-wide chunk plane)
//   copy:   read + write (2:1 and 1:1 mixes)
// Buffers are 2 GB so nothing is served by the 256 MB Infinity Cache.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ src, size_t n16, unsigned* out) {
    uint4 s = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 4096;          // 64 KB blocks
    for (size_t b = (size_t)blockIdx.x * 4096; b < n16; b += stride) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint4 v = src[b + i * 256 + threadIdx.x];
            s.x ^= v.x; s.y ^= v.y; s.z ^= v.z; s.w ^= v.w;
        }
    }
    if ((s.x ^ s.y ^ s.z ^ s.w) == 0x12345678u) out[0] = 1;
}

__device__ __forceinline__ void glds(const char* sbase, unsigned voff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds), "v"(voff), "s"(sbase) : "memory", "m0");
}

template <int TILE>
__global__ __launch_bounds__(256) void k_dma(const char* __restrict__ src, size_t bytes, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + wave * 16384;
    const size_t blk = 64 * 1024;                              // per workgroup per iteration
    const size_t nblk = TILE == 1 ? 32768 : bytes / blk;
    for (size_t b = blockIdx.x; b < nblk; b += gridDim.x) {
        const char* base = src + b * blk;
        if (TILE == 1) {
            // block b = a 34-row tile: row r at pitch 16 KB (wraps inside a 64 MB plane), 2176 B per row -> 72.25 KB; waves split rows
            // tile b of a [32 planes][16 images][256][256] x 64 B tensor: plane-major, 32x32-pixel tiles with a 1-pixel halo
            const size_t t = b % 1024, plane_i = (b / 1024) % 32;
            const size_t n = t / 64, ty = (t / 8) % 8, tx = t % 8;
            const long y0 = (long)ty * 32 - 1, x0 = (long)tx * 32 - 1;
            long off = (long)(plane_i << 26) + ((long)(n * 256 + (y0 < 0 ? 0 : y0)) * 256 + (x0 < 0 ? 0 : x0)) * 64;
            if (off + 34 * 16384 > (long)bytes) off = (long)bytes - 34 * 16384;
            const char* pb = src + off;
#pragma unroll
            for (int i = 0; i < 9; ++i) {                      // 34 rows x 2176 B = 73984 B = 72.25 waves of 1 KB; 4 waves x 18 + ...
                const unsigned idx = (unsigned)(i * 4 + wave) * 2u;   // two 1 KB pieces per step
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const unsigned s = (idx + h) * 1024u + lane * 16u;   // linear byte inside the tile image
                    const unsigned row = s / 2176u, col = s - row * 2176u;
                    if (row < 34u) glds(pb, row * 16384u + col, lds0 + ((i * 2 + h) & 7) * 1024);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) glds(base, (unsigned)((i * 4 + wave) * 1024 + lane * 16), lds0 + (i & 7) * 1024);
            if (TILE == 2) {   // + 32 KB that every workgroup re-reads (L2 hits), like a chunk's packed weights
#pragma unroll
                for (int i = 0; i < 8; ++i) glds(src, (unsigned)((i * 4 + wave) * 1024 + lane * 16), lds0 + (i & 7) * 1024);
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    if (smem[threadIdx.x] == 0x5a && bytes == 1) out[0] = 1;
}


// LDS-DMA streaming (waves 0-3) while waves 4-11 keep the matrix pipe busy with random f16 operands: what the
// memory pipe delivers under the power state of a convolution.
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(768) void k_dma_mfma(const char* __restrict__ src, size_t bytes, unsigned* out, int mfma_on) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    volatile unsigned* flag = reinterpret_cast<volatile unsigned*>(smem + 65536);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0) *flag = 0;
    __syncthreads();
    if (wave < 4) {
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + wave * 16384;
        const size_t blk = 64 * 1024, nblk = bytes / blk;
        for (size_t b = blockIdx.x; b < nblk; b += gridDim.x) {
            const char* base = src + b * blk;
#pragma unroll
            for (int i = 0; i < 16; ++i) glds(base, (unsigned)((i * 4 + wave) * 1024 + lane * 16), lds0 + (i & 7) * 1024);
            __builtin_amdgcn_s_waitcnt(0x0F70);
        }
        if (lane == 0) atomicAdd((unsigned*)flag, 1u);
    } else if (mfma_on) {
        unsigned h = threadIdx.x * 2654435761u + blockIdx.x; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        uint4 fa = make_uint4((h & 0x8fff8fffu) | 0x30003000u, (h * 3 & 0x8fff8fffu) | 0x30003000u, (h * 5 & 0x8fff8fffu) | 0x30003000u, (h * 7 & 0x8fff8fffu) | 0x30003000u);
        uint4 fb = make_uint4((h * 11 & 0x8fff8fffu) | 0x30003000u, (h * 13 & 0x8fff8fffu) | 0x30003000u, (h * 17 & 0x8fff8fffu) | 0x30003000u, (h * 19 & 0x8fff8fffu) | 0x30003000u);
        float16v acc[4];
        for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        const char* lp = smem + (wave - 4) * 8192 + lane * 16;   // reads the DMA destination area (no data dependence on it)
        while (*flag < 4u) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (mfma_on & 2) {   // a consumer's LDS diet: ~1.2 KB of ds_read_b128 per MFMA
                    uint4 v; asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) char*)(lp + (u & 7) * 1024)) : "memory");
                    fb.x ^= v.x & 1u;
                    if (u & 1) { uint4 w; asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(w) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) char*)(lp + 512)) : "memory"); fa.y ^= w.y & 1u; }
                }
                acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, fa), __builtin_bit_cast(half8, fb), acc[u & 3], 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
        if (s == 1.2345f) out[1] = 1;
    }
}

__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16, int reads_per_write) {
    const size_t stride = (size_t)gridDim.x * 4096;
    for (size_t b = (size_t)blockIdx.x * 4096; b < n16; b += stride) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            uint4 v = src[b + i * 256 + threadIdx.x];
            if (reads_per_write == 2) {
                const uint4 u = src[(b + n16 / 2 + i * 256 + threadIdx.x) % n16];
                v.x ^= u.x; v.y ^= u.y; v.z ^= u.z; v.w ^= u.w;
            }
            dst[b + i * 256 + threadIdx.x] = v;
        }
    }
}

int main() {
    const size_t bytes = (size_t)2 << 30;
    char *a, *b; unsigned* out;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dma<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dma<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dma<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    auto time = [&](const char* name, double moved, auto launch) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("%-76s %8.3f ms  %7.2f TB/s\n", name, ms, moved / ms * 1e-9);
        }
    };
    for (int wgs : {512, 1024, 2048}) {
        char nm[96];
        snprintf(nm, sizeof nm, "read  global_load_dwordx4, %d wgs", wgs);
        time(nm, (double)bytes, [&] { hipLaunchKernelGGL(k_read, dim3(wgs), dim3(256), 0, 0, (const uint4*)a, bytes / 16, out); });
    }
    for (int wgs : {256, 512, 1024}) {
        char nm[96];
        snprintf(nm, sizeof nm, "read  LDS-DMA 64 KB blocks, %d wgs", wgs);
        time(nm, (double)bytes, [&] { hipLaunchKernelGGL(k_dma<0>, dim3(wgs), dim3(256), 65536, 0, a, bytes, out); });
        snprintf(nm, sizeof nm, "read  LDS-DMA 34x2176 B tiles, %d wgs", wgs);
        time(nm, 32768.0 * 73984.0, [&] { hipLaunchKernelGGL(k_dma<1>, dim3(wgs), dim3(256), 65536, 0, a, bytes, out); });
    }
    time("read  LDS-DMA 64 KB HBM + 32 KB L2-hit, 256 wgs (HBM part)", (double)bytes, [&] { hipLaunchKernelGGL(k_dma<2>, dim3(256), dim3(256), 65536, 0, a, bytes, out); });
    time("read  LDS-DMA 64 KB HBM + 32 KB L2-hit, 512 wgs (HBM part)", (double)bytes, [&] { hipLaunchKernelGGL(k_dma<2>, dim3(512), dim3(256), 65536, 0, a, bytes, out); });
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dma_mfma), hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 64));
    time("read  LDS-DMA 64 KB blocks, 256 wgs, 8 idle waves", (double)bytes, [&] { hipLaunchKernelGGL(k_dma_mfma, dim3(256), dim3(768), 65536 + 64, 0, a, bytes, out, 0); });
    time("read  LDS-DMA 64 KB blocks, 256 wgs, 8 MFMA waves (random data)", (double)bytes, [&] { hipLaunchKernelGGL(k_dma_mfma, dim3(256), dim3(768), 65536 + 64, 0, a, bytes, out, 1); });
    time("read  LDS-DMA 64 KB blocks, 256 wgs, 8 MFMA waves + ds_read_b128 streams", (double)bytes, [&] { hipLaunchKernelGGL(k_dma_mfma, dim3(256), dim3(768), 65536 + 64, 0, a, bytes, out, 3); });
    time("copy  1 read : 1 write, 2048 wgs", 2.0 * bytes, [&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, bytes / 16, 1); });
    time("copy  2 reads : 1 write, 2048 wgs", 3.0 * bytes, [&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, bytes / 16, 2); });
    return 0;
}
