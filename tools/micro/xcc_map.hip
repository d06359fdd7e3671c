// Which XCD does workgroup b of a 256-workgroup, one-per-CU launch run on?  (HW_REG_XCC_ID; default stream, a second
// stream, and a second stream while the first is busy.)  hipcc --offload-arch=gfx950 -O2 -o xcc_map xcc_map.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(1024) void probe(unsigned* out, int spin) {
    extern __shared__ char smem[];
    if (threadIdx.x == 0) {
        out[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u;
        smem[0] = 1;
    }
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
}
static void show(const char* tag, unsigned* d, int n) {
    unsigned h[256];
    hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
    int cnt[16] = {0}, rr = 0;
    for (int b = 0; b < n; ++b) { cnt[h[b] & 15]++; if ((h[b] & 7) == ((h[0] + b) & 7)) rr++; }
    printf("%-28s first 16:", tag);
    for (int b = 0; b < 16; ++b) printf(" %u", h[b]);
    printf(" | per XCD:");
    for (int x = 0; x < 8; ++x) printf(" %d", cnt[x]);
    printf(" | round-robin from wg0's XCD: %d/%d\n", rr, n);
}
int main() {
    unsigned* d; hipMalloc(&d, 1024);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, -1);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe, dim3(256), dim3(1024), 150 * 1024, 0, d, 0); hipDeviceSynchronize(); show("default stream", d, 256);
        hipLaunchKernelGGL(probe, dim3(256), dim3(1024), 150 * 1024, s1, d, 0); hipDeviceSynchronize(); show("stream 1", d, 256);
        hipLaunchKernelGGL(probe, dim3(256), dim3(1024), 150 * 1024, s2, d, 0); hipDeviceSynchronize(); show("stream 2 (high priority)", d, 256);
        hipLaunchKernelGGL(probe, dim3(100), dim3(1024), 150 * 1024, s1, d, 0); hipDeviceSynchronize();
        hipLaunchKernelGGL(probe, dim3(256), dim3(1024), 150 * 1024, s1, d, 0); hipDeviceSynchronize(); show("stream 1 after a 100-wg launch", d, 256);
        hipLaunchKernelGGL(probe, dim3(13), dim3(64), 0, 0, d + 512, 2000);   // a small busy kernel on the default stream
        hipLaunchKernelGGL(probe, dim3(256), dim3(1024), 150 * 1024, s1, d, 0); hipDeviceSynchronize(); show("stream 1, default busy", d, 256);
    }
    return 0;
}
