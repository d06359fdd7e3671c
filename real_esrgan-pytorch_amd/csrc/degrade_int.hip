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
// All shifts are arithmetic (floor), so negative accumulators round the same way on both sides.
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

}  // namespace resr
