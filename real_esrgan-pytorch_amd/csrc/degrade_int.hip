// degrade_int.hip -- "integer mode" of the blur and resize stages (north_star: blur / resize / JPEG bit-exact in
// integer mode; SURVEY.md §7 defines it: uint8 images, fixed-point taps, integer accumulation, so that the CPU
// restatement (oracle/imgproc_int_ref.py) and these kernels agree bit for bit whatever the summation order).
// Reference float ops they shadow: imgproc.py:1089-1121 (filter2d_torch) and the F.interpolate call sites
// train_realesrnet.py:288,326-329,349-351,366-368.
//
//   filter2d_u8  dst = clamp((sum_taps q[dy][dx] * src[reflect(y+dy-ry)][reflect(x+dx-rx)] + 2^13) >> 14, 0, 255)
//                q = kernel in Q14 (int32, made on the host so that the taps sum to exactly 2^14), int32 accumulate
//                (|acc| <= 255 * sum|q|: a sinc kernel's sum|q| stays far below 2^31 / 255).
//   resize_u8    bilinear / bicubic: per-axis tap tables (clamped source index + Q11 weight, 2 or 4 taps, made on the host in
//                float64 from ATen's coordinate map), horizontal pass in int32, vertical pass in int64,
//                dst = clamp((acc + 2^21) >> 22, 0, 255);  area (adaptive average): integer window sums,
//                dst = (2 * sum + count) / (2 * count)  (round half up).
//   jpeg_u8      DiffJPEG(differentiable=False) (imgproc.py:1462-1494) as an integer round trip: colour matrices and the
//                DCT matrix in Q20, int64 accumulation, quantiser steps rint(table * factor * 2^20) (factor in float64 from
//                the float32 quality), round-half-even division, the inverse transform and colour matrix likewise; formats
//                stage by stage in oracle/imgproc_int_ref.py (jpeg_u8), which this kernel reproduces bit for bit.
// All shifts are arithmetic (floor), so negative accumulators round the same way on both sides.
#include <cmath>
#include <mutex>

#include "common.h"

namespace resr {

__device__ __forceinline__ int reflect_i(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

// block = 32 x 8 threads, tile = 32 x 32 outputs (4 rows per thread).  LDS: the uint8 tile with its halo (one byte per
// pixel: a 52 x 52 tile of a 21 x 21 kernel is 2.7 KB), then the taps.
__global__ __launch_bounds__(256) void filter2d_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                          const int32_t* __restrict__ taps, int c, int h, int w, int kh,
                                                          int kw, int per_sample) {
    extern __shared__ int32_t smi[];
    const int ry = kh / 2, rx = kw / 2;
    const int tw = 32 + 2 * rx, th = 32 + 2 * ry;
    int32_t* q = smi;                                              // kh * kw taps
    uint8_t* tile = reinterpret_cast<uint8_t*>(smi + kh * kw);     // th * tw bytes
    const int plane = blockIdx.z;                                  // n * c + ch
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
    const uint8_t* sp = src + (size_t)plane * h * w;
    const int32_t* kp = taps + (per_sample ? (size_t)(plane / c) * kh * kw : 0);
    for (int i = threadIdx.x; i < tw * th; i += 256) {
        const int ty = i / tw, tx = i - ty * tw;
        const int iy = reflect_i(y0 + ty - ry, h), ix = reflect_i(x0 + tx - rx, w);
        tile[i] = (iy >= 0 && iy < h && ix >= 0 && ix < w) ? sp[(size_t)iy * w + ix] : (uint8_t)0;
    }
    for (int i = threadIdx.x; i < kh * kw; i += 256) q[i] = kp[i];
    __syncthreads();
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    int32_t acc[4] = {0, 0, 0, 0};
    for (int dy = 0; dy < kh; ++dy)
        for (int dx = 0; dx < kw; ++dx) {
            const int32_t wv = q[dy * kw + dx];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += wv * (int32_t)tile[(ly * 4 + j + dy) * tw + lx + dx];
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int y = y0 + ly * 4 + j, x = x0 + lx;
        if (y < h && x < w) {
            const int32_t v = (acc[j] + (1 << 13)) >> 14;
            dst[(size_t)plane * h * w + (size_t)y * w + x] = (uint8_t)min(max(v, 0), 255);
        }
    }
}

int filter2d_u8_dispatch(const uint8_t* src, uint8_t* dst, const int32_t* taps, int n, int c, int h, int w, int kh, int kw,
                         int per_sample, hipStream_t st) {
    if (!src || !dst || !taps || n <= 0 || c <= 0 || h <= 0 || w <= 0) return fail(RESR_ERR_ARG, "filter2d_u8: bad argument");
    if (!(kh & 1) || !(kw & 1) || kh > 63 || kw > 63) return fail(RESR_ERR_ARG, "filter2d_u8: kernel must be odd-sized, <= 63");
    if (kh / 2 >= h || kw / 2 >= w) return fail(RESR_ERR_ARG, "filter2d_u8: reflect padding needs kernel/2 < image size");
    if ((long)n * c > 65535) return fail(RESR_ERR_ARG, "filter2d_u8: n*c > 65535");
    const size_t lds = (size_t)kh * kw * 4 + (size_t)(32 + kh - 1) * (32 + kw - 1);
    hipLaunchKernelGGL(filter2d_u8_kernel, dim3((w + 31) / 32, (h + 31) / 32, n * c), dim3(256), (lds + 3) / 4 * 4, st, src, dst,
                       taps, c, h, w, kh, kw, per_sample);
    RESR_CHECK_LAUNCH("filter2d_u8_kernel");
    return RESR_OK;
}

// taps = 2 (bilinear) or 4 (bicubic); idx_* [out][taps] clamped source indices, w_* [out][taps] Q11 weights summing to 2^11
__global__ __launch_bounds__(256) void resize_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int planes,
                                                        int h, int w, int oh, int ow, int taps,
                                                        const int32_t* __restrict__ idx_y, const int32_t* __restrict__ w_y,
                                                        const int32_t* __restrict__ idx_x, const int32_t* __restrict__ w_x) {
    const long total = (long)planes * oh * ow;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ox = (int)(i % ow);
    const int oy = (int)((i / ow) % oh);
    const int p = (int)(i / ((long)ow * oh));
    const uint8_t* sp = src + (size_t)p * h * w;
    long long acc = 0;
    for (int a = 0; a < taps; ++a) {
        const uint8_t* row = sp + (size_t)idx_y[oy * taps + a] * w;
        int32_t r = 0;
        for (int b = 0; b < taps; ++b) r += w_x[ox * taps + b] * (int32_t)row[idx_x[ox * taps + b]];
        acc += (long long)w_y[oy * taps + a] * r;
    }
    const long long v = (acc + (1LL << 21)) >> 22;
    dst[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

__global__ __launch_bounds__(256) void resize_area_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                             int planes, int h, int w, int oh, int ow) {
    const long total = (long)planes * oh * ow;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ox = (int)(i % ow);
    const int oy = (int)((i / ow) % oh);
    const int p = (int)(i / ((long)ow * oh));
    const uint8_t* sp = src + (size_t)p * h * w;
    // adaptive_avg_pool2d windows: [floor(o*in/out), ceil((o+1)*in/out)) in exact integer arithmetic
    const int y0 = (int)(((long)oy * h) / oh), y1 = (int)((((long)oy + 1) * h + oh - 1) / oh);
    const int x0 = (int)(((long)ox * w) / ow), x1 = (int)((((long)ox + 1) * w + ow - 1) / ow);
    int32_t s = 0;
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) s += sp[(size_t)y * w + x];
    const int32_t cnt = (y1 - y0) * (x1 - x0);
    dst[i] = (uint8_t)((2 * s + cnt) / (2 * cnt));
}

int resize_u8_dispatch(const uint8_t* src, uint8_t* dst, int n, int c, int h, int w, int oh, int ow, int mode,
                       const int32_t* idx_y, const int32_t* w_y, const int32_t* idx_x, const int32_t* w_x, hipStream_t st) {
    if (!src || !dst || n <= 0 || c <= 0 || h <= 0 || w <= 0 || oh <= 0 || ow <= 0 || mode < 0 || mode > 2)
        return fail(RESR_ERR_ARG, "resize_u8: bad argument");
    const long total = (long)n * c * oh * ow;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (mode == 0) {
        if ((long)((h + oh - 1) / oh + 1) * ((w + ow - 1) / ow + 1) > 8000000L) return fail(RESR_ERR_ARG, "resize_u8: area window too large");
        hipLaunchKernelGGL(resize_area_u8_kernel, dim3(blocks), dim3(256), 0, st, src, dst, n * c, h, w, oh, ow);
        RESR_CHECK_LAUNCH("resize_area_u8_kernel");
        return RESR_OK;
    }
    if (!idx_y || !w_y || !idx_x || !w_x) return fail(RESR_ERR_ARG, "resize_u8: bilinear / bicubic need the four tap tables");
    hipLaunchKernelGGL(resize_u8_kernel, dim3(blocks), dim3(256), 0, st, src, dst, n * c, h, w, oh, ow, mode == 1 ? 2 : 4, idx_y, w_y,
                       idx_x, w_x);
    RESR_CHECK_LAUNCH("resize_u8_kernel");
    return RESR_OK;
}

// ---------------------------------------------------------------------------------------------------------
// JPEG round trip, integer mode
// ---------------------------------------------------------------------------------------------------------
struct JpegIntTables {
    long long dct[64];            // C[u][x] = rint(0.5 alpha(u) cos((2x+1) u pi / 16) 2^20)
    double ytab[64], ctab[64];    // the reference's (transposed) tables, imgproc.py:40-49, indexed [u*8+v]
};
__device__ JpegIntTables g_jpeg_int;

static void jpeg_int_tables_host(JpegIntTables& t) {
    static const double ystd[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
                                    14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
                                    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
    static const double cstd[16] = {17, 18, 24, 47, 18, 21, 26, 66, 24, 26, 56, 99, 47, 66, 99, 99};
    for (int u = 0; u < 8; ++u)
        for (int x = 0; x < 8; ++x) {
            const double a = u == 0 ? 1.0 / std::sqrt(2.0) : 1.0;
            t.dct[u * 8 + x] = (long long)std::nearbyint(0.5 * a * std::cos((2 * x + 1) * u * M_PI / 16) * 1048576.0);
        }
    for (int u = 0; u < 8; ++u)
        for (int v = 0; v < 8; ++v) {
            t.ytab[u * 8 + v] = ystd[v * 8 + u];
            t.ctab[u * 8 + v] = (u < 4 && v < 4) ? cstd[v * 4 + u] : 99.0;
        }
}

// one workgroup per 16x16 macroblock: 6 blocks of 8x8 (4 Y, Cb, Cr); everything between the uint8 load and the uint8 store
// is int64 in LDS (4 arrays of 3 KB)
__global__ __launch_bounds__(256) void jpeg_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                      const float* __restrict__ quality, int32_t* __restrict__ coeffs, int h,
                                                      int w, int mbx, int mby) {
    typedef long long i64;
    __shared__ i64 ycc[3][256];      // Q20, level-shifted
    __shared__ i64 a0[6][64], a1[6][64];
    __shared__ i64 dct[64];
    const int b = blockIdx.z, my = blockIdx.y, mx = blockIdx.x;
    const int t = threadIdx.x, ly = t >> 4, lx = t & 15;
    const int y = my * 16 + ly, x = mx * 16 + lx;
    const size_t hw = (size_t)h * w;
    const uint8_t* sp = src + (size_t)b * 3 * hw;
    i64 r = 0, g = 0, bl = 0;                            // zero padding up to a multiple of 16 (imgproc.py:1486-1488)
    if (y < h && x < w) {
        r = sp[(size_t)y * w + x]; g = sp[hw + (size_t)y * w + x]; bl = sp[2 * hw + (size_t)y * w + x];
    }
    ycc[0][t] = 313524 * r + 615514 * g + 119538 * bl - (128LL << 20);
    ycc[1][t] = -176933 * r - 347355 * g + 524288 * bl;
    ycc[2][t] = 524288 * r - 439026 * g - 85262 * bl;
    if (t < 64) dct[t] = g_jpeg_int.dct[t];
    __syncthreads();
    // block split, Q22: luma x 4, chroma = 2x2 sum
    a0[(ly >> 3) * 2 + (lx >> 3)][(ly & 7) * 8 + (lx & 7)] = ycc[0][t] << 2;
    if (t < 128) {
        const int cidx = t >> 6, e = t & 63, cy = e >> 3, cx = e & 7;
        const i64* pc = ycc[1 + cidx];
        a0[4 + cidx][e] = pc[(2 * cy) * 16 + 2 * cx] + pc[(2 * cy) * 16 + 2 * cx + 1] + pc[(2 * cy + 1) * 16 + 2 * cx] +
                          pc[(2 * cy + 1) * 16 + 2 * cx + 1];
    }
    __syncthreads();
    for (int i = t; i < 384; i += 256) {                 // pass 1: T[u][y] = sum_x C[u][x] blk[x][y]  -> Q22
        const int bi = i >> 6, u = (i >> 3) & 7, yy = i & 7;
        i64 s = 0;
#pragma unroll
        for (int xx = 0; xx < 8; ++xx) s += dct[u * 8 + xx] * a0[bi][xx * 8 + yy];
        a1[bi][u * 8 + yy] = (s + (1LL << 19)) >> 20;
    }
    __syncthreads();
    const double qd = (double)quality[b];
    const double factor = qd < 50.0 ? 50.0 / qd : 2.0 - qd / 50.0;        // imgproc.py:1134-1139
    for (int i = t; i < 384; i += 256) {                 // pass 2 (Q42), quantise (half to even), dequantise (Q20)
        const int bi = i >> 6, uv = i & 63, u = uv >> 3, v = uv & 7;
        i64 f = 0;
#pragma unroll
        for (int yy = 0; yy < 8; ++yy) f += dct[v * 8 + yy] * a1[bi][u * 8 + yy];
        const double tab = (bi < 4 ? g_jpeg_int.ytab[uv] : g_jpeg_int.ctab[uv]) * factor;
        i64 step = (i64)rint(tab * 1048576.0);
        if (step < 1) step = 1;
        const i64 d = step << 22, num = 2 * f + d, den = 2 * d;
        i64 q = num / den, rem = num - q * den;
        if (rem < 0) { q -= 1; rem += den; }             // floor division
        if (rem == 0 && (q & 1)) q -= 1;                 // a tie went up: back to the even neighbour
        if (coeffs) {
            // layout: per image [Y blocks (H/8 * W/8) | Cb blocks | Cr blocks] x 64, block order row-major
            const int nyb = mbx * 2 * mby * 2, ncb = mbx * mby;
            size_t off;
            if (bi < 4) off = (size_t)((my * 2 + (bi >> 1)) * (mbx * 2) + mx * 2 + (bi & 1));
            else off = (size_t)nyb + (size_t)(bi - 4) * ncb + (size_t)my * mbx + mx;
            coeffs[((size_t)b * (nyb + 2 * ncb) + off) * 64 + uv] = (int32_t)q;
        }
        a0[bi][uv] = q * step;
    }
    __syncthreads();
    for (int i = t; i < 384; i += 256) {                 // inverse pass 1: S[x][v] = sum_u C[u][x] D[u][v]  -> Q16
        const int bi = i >> 6, xx = (i >> 3) & 7, v = i & 7;
        i64 s = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) s += dct[u * 8 + xx] * a0[bi][u * 8 + v];
        a1[bi][xx * 8 + v] = (s + (1LL << 23)) >> 24;
    }
    __syncthreads();
    for (int i = t; i < 384; i += 256) {                 // inverse pass 2: rec[x][y] = sum_v C[v][y] S[x][v]  -> Q16
        const int bi = i >> 6, xx = (i >> 3) & 7, yy = i & 7;
        i64 s = 0;
#pragma unroll
        for (int v = 0; v < 8; ++v) s += dct[v * 8 + yy] * a1[bi][xx * 8 + v];
        a0[bi][xx * 8 + yy] = (s + (1LL << 19)) >> 20;
    }
    __syncthreads();
    if (y < h && x < w) {
        const i64 Y = (a0[(ly >> 3) * 2 + (lx >> 3)][(ly & 7) * 8 + (lx & 7)] + (128LL << 16)) << 20;      // Q36
        const i64 cb = a0[4][(ly >> 1) * 8 + (lx >> 1)], cr = a0[5][(ly >> 1) * 8 + (lx >> 1)];             // nearest x2
        const i64 R = (Y + 1470104 * cr + (1LL << 35)) >> 36;
        const i64 G = (Y - 360853 * cb - 748826 * cr + (1LL << 35)) >> 36;
        const i64 B = (Y + 1858077 * cb + (1LL << 35)) >> 36;
        uint8_t* dp = dst + (size_t)b * 3 * hw + (size_t)y * w + x;
        dp[0] = (uint8_t)(R < 0 ? 0 : (R > 255 ? 255 : R));
        dp[hw] = (uint8_t)(G < 0 ? 0 : (G > 255 ? 255 : G));
        dp[2 * hw] = (uint8_t)(B < 0 ? 0 : (B > 255 ? 255 : B));
    }
}

int jpeg_u8_dispatch(const uint8_t* src, uint8_t* dst, const float* quality, int32_t* coeffs, int n, int h, int w, hipStream_t st) {
    if (!src || !dst || !quality || n <= 0 || h <= 0 || w <= 0) return fail(RESR_ERR_ARG, "jpeg_u8: bad argument");
    if (n > 65535 || (h + 15) / 16 > 65535) return fail(RESR_ERR_ARG, "jpeg_u8: batch or height beyond the launch grid");
    static std::once_flag once[kMaxDevices];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= kMaxDevices) return fail(RESR_ERR_ARG, "jpeg_u8: device index beyond %d", kMaxDevices);
    hipError_t err = hipSuccess;
    std::call_once(once[dev], [&]() {
        static JpegIntTables host;
        jpeg_int_tables_host(host);
        err = hipMemcpyToSymbol(HIP_SYMBOL(g_jpeg_int), &host, sizeof(JpegIntTables));
    });
    if (err != hipSuccess) return fail(RESR_ERR_LAUNCH, "jpeg_u8: table upload failed: %s", hipGetErrorString(err));
    const int mbx = (w + 15) / 16, mby = (h + 15) / 16;
    hipLaunchKernelGGL(jpeg_u8_kernel, dim3(mbx, mby, n), dim3(256), 0, st, src, dst, quality, coeffs, h, w, mbx, mby);
    RESR_CHECK_LAUNCH("jpeg_u8_kernel");
    return RESR_OK;
}

}  // namespace resr
