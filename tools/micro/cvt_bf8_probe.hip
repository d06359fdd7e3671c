// cvt_bf8_probe.hip -- v_cvt_scalef32_pk_bf8_f16 at unit scale against "round the f16 pattern to its upper byte, nearest even"
// (csrc/pack.hip f16_bits_to_bf8) for all 65536 f16 patterns, both halves of the source, both destination words.
// build: hipcc --offload-arch=gfx950 -O2 -o cvt_bf8_probe cvt_bf8_probe.hip ; run: ./cvt_bf8_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef short short2v __attribute__((ext_vector_type(2)));

__global__ void k(unsigned* out) {
    const unsigned h = blockIdx.x * 256 + threadIdx.x;   // 0 .. 65535
    const unsigned src = h | ((h ^ 0x5555u) << 16);       // low half = h, high half = another pattern
    short2v r = {0, 0};
    r = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(r, __builtin_bit_cast(half2v, src), 1.0f, false);
    r = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(r, __builtin_bit_cast(half2v, src), 1.0f, true);
    out[h] = __builtin_bit_cast(unsigned, r);
}

static unsigned ref8(unsigned h) { return ((h + 0x7fu + ((h >> 8) & 1u)) >> 8) & 0xffu; }

int main() {
    unsigned* d;
    hipMalloc(&d, 65536 * 4);
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d);
    std::vector<unsigned> o(65536);
    hipMemcpy(o.data(), d, 65536 * 4, hipMemcpyDeviceToHost);
    int bad0 = 0, bad1 = 0, bad_words = 0, shown = 0, nan_in = 0;
    for (unsigned h = 0; h < 65536; ++h) {
        const unsigned h2 = (h ^ 0x5555u) & 0xffffu;
        const bool nan0 = (h & 0x7c00u) == 0x7c00u, nan1 = (h2 & 0x7c00u) == 0x7c00u;
        const unsigned b0 = o[h] & 0xff, b1 = (o[h] >> 8) & 0xff, b2 = (o[h] >> 16) & 0xff, b3 = o[h] >> 24;
        if (b0 != b2 || b1 != b3) ++bad_words;
        if (nan0) ++nan_in;
        if (!nan0 && b0 != ref8(h)) { if (shown++ < 12) printf("h=%04x got %02x want %02x\n", h, b0, ref8(h)); ++bad0; }
        if (!nan1 && b1 != ref8(h2)) ++bad1;
    }
    printf("finite-or-inf-free patterns: low half mismatches %d, high half mismatches %d, word-select mismatches %d (inf/nan inputs skipped: %d)\n", bad0, bad1, bad_words, nan_in);
    return 0;
}
