// tr8_probe.hip -- what ds_read_b64_tr_b8 hands each lane (gfx950; the ISA text is not in this image).  LDS holds a byte matrix whose
// byte at (row r, column c) is tagged (r << 4 | c) for r, c < 16; lane l points at row (l & 15) (+ 8-byte column block (l >> 4) & 1) as the
// 16-bit variant's users do (wgrad.hip tr_read); the eight bytes each lane receives are printed.
// build: hipcc --offload-arch=gfx950 -O2 -o tr8_probe tr8_probe.hip ; run: ./tr8_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v2i __attribute__((ext_vector_type(2)));

__global__ void k(unsigned* out, int row_bytes, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned char m[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) {
        const int r = i / row_bytes, c = i % row_bytes;
        m[i] = (unsigned char)(((r & 15) << 4) | (c & 15));
    }
    __syncthreads();
    const int l = threadIdx.x;
    // mode 0: lane l -> row (l & 15), byte column 8 * ((l >> 4) & 1), rows 16 * (l >> 5) further down for the upper half
    // mode 1: lane l -> row (l & 7) + 8 * (l >> 5), byte column 8 * ((l >> 3) & 3)
    const int row = mode == 0 ? (l & 15) + 16 * (l >> 5) : (l & 7) + 8 * (l >> 5);
    const int col = mode == 0 ? 8 * ((l >> 4) & 1) : 8 * ((l >> 3) & 3);
    auto p = reinterpret_cast<__attribute__((address_space(3))) v2i*>((__attribute__((address_space(3))) unsigned char*)(m + row * row_bytes + col));
    const v2i r = __builtin_amdgcn_ds_read_tr8_b64_v2i32(p);
    out[l * 2] = (unsigned)r.x;
    out[l * 2 + 1] = (unsigned)r.y;
}

int main() {
    unsigned* d;
    (void)hipMalloc(&d, 128 * 4);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 64, mode);
        unsigned h[128];
        (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d (row stride 64 B): lane -> eight (row,col) tags\n", mode);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d:", l);
            for (int b = 0; b < 8; ++b) { const unsigned v = (h[l * 2 + b / 4] >> (8 * (b % 4))) & 0xff; printf(" (%x,%x)", v >> 4, v & 15); }
            printf("\n");
        }
    }
    return 0;
}
