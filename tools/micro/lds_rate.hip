// lds_rate.hip -- LDS read throughput per CU on gfx950 for the read flavours the kernels use (no MFMA):
// ds_read_b64_tr_b16 (transpose read), ds_read_b64, ds_read_b128; 8 waves per CU, conflict-free lane maps.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 fp16x4_t;
__device__ __forceinline__ uint2 tr_read(const char* a) {
    auto p = reinterpret_cast<__attribute__((address_space(3))) fp16x4_t*>((__attribute__((address_space(3))) char*)a);
    return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4f16(p));
}
template <int MODE>
__global__ __launch_bounds__(512, 1) void k(unsigned* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = i;
    __syncthreads();
    const int a16 = lane & 15;
    const char* btr = smem + wave * 4096 + (((lane >> 5) << 3) + (a16 >> 2)) * 64 + ((((lane >> 4) & 1) << 4) + ((a16 & 3) << 2)) * 2;
    const char* b64 = smem + wave * 4096 + lane * 8;
    const char* b128 = smem + wave * 4096 + lane * 16;
    unsigned s = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) { const uint2 v = tr_read(btr + u * 1024); s ^= v.x ^ v.y; }
            if (MODE == 1) { const uint2 v = *reinterpret_cast<const uint2*>(b64 + u * 512); s ^= v.x ^ v.y; }
            if (MODE == 2) { const uint4 v = *reinterpret_cast<const uint4*>(b128 + (u & 3) * 1024); s ^= v.x ^ v.y ^ v.z ^ v.w; }
        }
        asm volatile("" ::: "memory");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
static void run(const char* name, int bytes_per_lane) {
    unsigned* out;
    (void)hipMalloc(&out, 256 * 512 * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 64 * 1024, 0, out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double bytes = 16.0 * iters * 512 * bytes_per_lane;   // per CU
        printf("%-22s %8.3f ms  %7.1f GB/s per CU  (%5.1f B/clk at 2.4 GHz)  %6.2f TB/s chip\n", name, ms, bytes / ms * 1e-6,
               bytes / ms * 1e-6 / 2.4, bytes * 256 / ms * 1e-9);
    }
    (void)hipFree(out);
}
int main() {
    run<0>("ds_read_b64_tr_b16", 8);
    run<1>("ds_read_b64", 8);
    run<2>("ds_read_b128", 16);
    return 0;
}
