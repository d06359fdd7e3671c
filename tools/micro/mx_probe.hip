// mx_probe.hip -- operand layout and scale semantics of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 x fp8 e4m3), probed
// against a host evaluation: which K elements a lane's 32 bytes are, which K block a lane's scale byte applies to, how the
// e8m0 scale enters.  (The ISA text is not in this image; an MX correction stage for exact16 -- DESIGN.md section 7 -- rests on it.)
// Candidate K maps of lane l (h = l >> 5), byte b (0..31) of its 8 registers:
//   map 0: k = 32 h + b                          (a lane half owns one 32-wide scale block)
//   map 1: k = 16 h + (b & 15) + 32 (b >> 4)     (two 16-wide runs per lane, as two stacked K = 32 instructions)
//   map 2: k = 8 h + (b & 7) + 16 (b >> 3)       (four 8-wide runs, as four stacked K = 16 instructions)
// build: hipcc --offload-arch=gfx950 -O2 -o mx_probe mx_probe.hip ; run: ./mx_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float float16v __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));

__global__ void k(const unsigned char* A, const unsigned char* B, const unsigned char* sa, const unsigned char* sb, float* D, int map) {
    const int l = threadIdx.x, h = l >> 5, i = l & 31;
    unsigned char ab[32], bb[32];
    for (int b = 0; b < 32; ++b) {
        const int kk = map == 0 ? 32 * h + b : map == 1 ? 16 * h + (b & 15) + 32 * (b >> 4) : 8 * h + (b & 7) + 16 * (b >> 3);
        ab[b] = A[i * 64 + kk];          // A[row i][k]
        bb[b] = B[kk * 32 + i];          // B[k][col i]
    }
    v8i av, bv;
    for (int r = 0; r < 8; ++r) {
        av[r] = (int)(ab[4 * r] | (ab[4 * r + 1] << 8) | (ab[4 * r + 2] << 16) | ((unsigned)ab[4 * r + 3] << 24));
        bv[r] = (int)(bb[4 * r] | (bb[4 * r + 1] << 8) | (bb[4 * r + 2] << 16) | ((unsigned)bb[4 * r + 3] << 24));
    }
    // scale operands: this lane's byte = the scale of (its row / column, K block h)
    const int sva = sa[i * 2 + h], svb = sb[i * 2 + h];
    float16v c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 0, 0, 0, sva, 0, svb);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = c[r];
}

static float e4m3(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f;
    if (e == 0) f = ldexpf((float)m, -9);
    else if (e == 15 && m == 7) f = NAN;
    else f = ldexpf(1.f + m / 8.f, e - 7);
    return s ? -f : f;
}

int main() {
    std::vector<unsigned char> A(32 * 64), B(64 * 32), sa(64), sb(64);
    srand(3);
    for (auto& v : A) { v = rand() & 0xff; if ((v & 0x7f) == 0x7f) v ^= 1; }
    for (auto& v : B) { v = rand() & 0xff; if ((v & 0x7f) == 0x7f) v ^= 1; }
    for (auto& v : sa) v = 120 + rand() % 12;
    for (auto& v : sb) v = 122 + rand() % 10;
    unsigned char *dA, *dB, *dsa, *dsb;
    float* dD;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dsa, 64); hipMalloc(&dsb, 64); hipMalloc(&dD, 32 * 32 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    hipMemcpy(dsa, sa.data(), 64, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 64, hipMemcpyHostToDevice);
    for (int map = 0; map < 3; ++map) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD, map);
        std::vector<float> D(32 * 32);
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        // host evaluations: scale block of K element kk = kk / 32 (what the lane-half scale of map 0 means)
        double worst = 0, worst_noscale = 0, ref_max = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double s = 0, s0 = 0;
                for (int kk = 0; kk < 64; ++kk) {
                    const double p = (double)e4m3(A[i * 64 + kk]) * e4m3(B[kk * 32 + j]);
                    s += p * ldexp(1.0, sa[i * 2 + kk / 32] - 127) * ldexp(1.0, sb[j * 2 + kk / 32] - 127);
                    s0 += p;
                }
                worst = fmax(worst, fabs(D[i * 32 + j] - s));
                worst_noscale = fmax(worst_noscale, fabs(D[i * 32 + j] - s0));
                ref_max = fmax(ref_max, fabs(s));
            }
        printf("K map %d: max |D - ref(scaled, block = k / 32)| = %.3e   (ref max %.3e; against the unscaled sum: %.3e)\n", map, worst, ref_max, worst_noscale);
    }
    return 0;
}
