// degrade.hip -- the second-order degradation pipeline as HIP kernels (reference imgproc.py device ops
// and their call sites train_realesrnet.py:268-377).  Images are planar fp32 [N,3,H,W] in [0,1] (the
// module surface); every op is one pass (or a short chain of passes) bounded by HBM bandwidth:
//   filter2d      reflect-padded kh x kw correlation, shared or per-sample kernel (imgproc.py:1089-1121)
//   usm combine   USMSharp.forward epilogues (imgproc.py:1526-1537); the 51x51 Gaussian is applied as its
//                 two separable 51-tap passes (the kernel is an outer product, imgproc.py:1522-1523)
//   resize        F.interpolate area / bilinear / bicubic, align_corners=False, scale_factor= or size=
//   noise         Gaussian (one gray field shared by the batch, imgproc.py:854) and Poisson (per-sample
//                 unique-value count without host sync, imgproc.py:892-905) with Philox4x32-10 sampling
//   jpeg          DiffJPEG(differentiable=False) (imgproc.py:1195-1494), one workgroup per 16x16 macroblock
//   quantize+crop train_realesrnet.py:374-377, imgproc.py:1894-1934
// No host synchronisation anywhere: per-sample scalars live in small device arrays.
#include <math.h>

#include <mutex>

#include <stdlib.h>

#include "common.h"

namespace resr {

// ---------------------------------------------------------------------------------------------------------
// filter2d
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

// block = 32 x 8 threads, tile = 32 x 32 outputs (4 rows per thread); LDS: tile + halo, then the taps
__global__ __launch_bounds__(256) void filter2d_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                       const float* __restrict__ kern, int c, int h, int w, int kh,
                                                       int kw, int per_sample) {
    extern __shared__ float sm[];
    const int ry = kh / 2, rx = kw / 2;
    const int tw = 32 + 2 * rx, th = 32 + 2 * ry;
    float* tile = sm;
    float* taps = sm + tw * th;
    const int plane = blockIdx.z;                       // n * c + ch
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
    const float* sp = src + (size_t)plane * h * w;
    const float* kp = kern + (per_sample ? (size_t)(plane / c) * kh * kw : 0);
    for (int i = threadIdx.x; i < tw * th; i += 256) {
        const int ty = i / tw, tx = i - ty * tw;
        const int iy = reflect(y0 + ty - ry, h), ix = reflect(x0 + tx - rx, w);
        tile[i] = (iy >= 0 && iy < h && ix >= 0 && ix < w) ? sp[(size_t)iy * w + ix] : 0.f;
    }
    for (int i = threadIdx.x; i < kh * kw; i += 256) taps[i] = kp[i];
    __syncthreads();
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int dy = 0; dy < kh; ++dy) {
        for (int dx = 0; dx < kw; ++dx) {
            const float wv = taps[dy * kw + dx];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += wv * tile[(ly * 4 + j + dy) * tw + lx + dx];
        }
    }
    float* dp = dst + (size_t)plane * h * w;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int y = y0 + ly * 4 + j, x = x0 + lx;
        if (y < h && x < w) dp[(size_t)y * w + x] = acc[j];
    }
}

// 21 x 21 taps (every blur / sinc kernel of the degradation is zero-padded to 21 x 21, dataset.py:101-103): the generic
// kernel above reads LDS once per FMA and is LDS-bound at ~28 % of the vector rate.  Here a thread owns 8 consecutive
// outputs of one row of a 64 x 32 tile: per tap row it reads its 28-float window (7 x ds_read_b128) once for 168 FMAs, and the 21
// taps of the row come through the SCALAR cache into SGPRs (they are uniform over the workgroup: one sample's kernel) -- as LDS
// broadcast reads they were half of the kernel's LDS traffic, and the LDS pipe, not the vector pipe, set its rate.
// Same accumulation order (dy, then dx) per output as the generic kernel: the results are the same bits.
__global__ __launch_bounds__(256) void filter2d21_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                         const float* __restrict__ kern, int c, int h, int w, int per_sample) {
    constexpr int K = 21, R = 10, OW = 64, TW = OW + 2 * R, TH = 32 + 2 * R;   // 84 x 52 tile for 64 x 32 outputs
    __shared__ __attribute__((aligned(16))) float tile[TH * TW];
    const int plane = blockIdx.z;                       // n * c + ch
    const int x0 = blockIdx.x * OW, y0 = blockIdx.y * 32;
    const float* sp = src + (size_t)plane * h * w;
    const float* kp = kern + (per_sample ? (size_t)(plane / c) * K * K : 0);     // workgroup-uniform
    for (int i = threadIdx.x; i < TW * TH; i += 256) {
        const int ty = i / TW, tx = i - ty * TW;
        const int iy = reflect(y0 + ty - R, h), ix = reflect(x0 + tx - R, w);
        tile[i] = (iy >= 0 && iy < h && ix >= 0 && ix < w) ? sp[(size_t)iy * w + ix] : 0.f;
    }
    __syncthreads();
    const int cg = threadIdx.x & 7, row = threadIdx.x >> 3;   // 8 column groups of 8 x 32 rows
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int dy = 0; dy < K; ++dy) {
        float win[28];
        const float4* wp = reinterpret_cast<const float4*>(tile + (row + dy) * TW + cg * 8);
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const float4 v = wp[q];
            win[q * 4] = v.x; win[q * 4 + 1] = v.y; win[q * 4 + 2] = v.z; win[q * 4 + 3] = v.w;
        }
        const float* tr = kp + dy * K;
#pragma unroll
        for (int dx = 0; dx < K; ++dx) {
            const float t = tr[dx];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += t * win[j + dx];
        }
    }
    const int y = y0 + row;
    if (y < h) {
        float* dp = dst + ((size_t)plane * h + y) * w + x0 + cg * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (x0 + cg * 8 + j < w) dp[j] = acc[j];
    }
}

// 1 x K / K x 1 taps (the two passes of USMSharp's separable 51-tap Gaussian): a thread owns 4 consecutive outputs along the
// filter axis and reads its K+3 window once for 4K FMAs (the generic kernel reads LDS once per FMA).
template <int K, int AXIS>   // AXIS 0: taps along x, 1: taps along y
__global__ __launch_bounds__(256) void filter1d_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                       const float* __restrict__ kern, int h, int w) {
    constexpr int R = K / 2, WIN = K + 3, WIN4 = (WIN + 3) / 4 * 4;
    constexpr int TW = AXIS == 0 ? 32 + WIN4 - 4 + 4 : 32, TH = AXIS == 0 ? 32 : 32 + K - 1;   // x pass: row padded for whole float4 windows
    __shared__ __attribute__((aligned(16))) float tile[TH * TW];
    __shared__ __attribute__((aligned(16))) float taps[WIN4];
    const int plane = blockIdx.z;
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
    const float* sp = src + (size_t)plane * h * w;
    for (int i = threadIdx.x; i < TW * TH; i += 256) {
        const int ty = i / TW, tx = i - ty * TW;
        const int iy = reflect(y0 + ty - (AXIS == 1 ? R : 0), h), ix = reflect(x0 + tx - (AXIS == 0 ? R : 0), w);
        tile[i] = (iy >= 0 && iy < h && ix >= 0 && ix < w) ? sp[(size_t)iy * w + ix] : 0.f;
    }
    for (int i = threadIdx.x; i < WIN4; i += 256) taps[i] = i < K ? kern[i] : 0.f;
    __syncthreads();
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float win[WIN4], tp[WIN4];
#pragma unroll
    for (int q = 0; q < WIN4 / 4; ++q) {
        const float4 t = reinterpret_cast<const float4*>(taps)[q];
        tp[q * 4] = t.x; tp[q * 4 + 1] = t.y; tp[q * 4 + 2] = t.z; tp[q * 4 + 3] = t.w;
    }
    int ox, oy;
    if (AXIS == 0) {
        const int cg = threadIdx.x & 7, row = threadIdx.x >> 3;
        const float4* wp = reinterpret_cast<const float4*>(tile + row * TW + cg * 4);
#pragma unroll
        for (int q = 0; q < WIN4 / 4; ++q) {
            const float4 v = wp[q];
            win[q * 4] = v.x; win[q * 4 + 1] = v.y; win[q * 4 + 2] = v.z; win[q * 4 + 3] = v.w;
        }
        ox = x0 + cg * 4; oy = y0 + row;
    } else {
        const int col = threadIdx.x & 31, rg = threadIdx.x >> 5;
#pragma unroll
        for (int q = 0; q < WIN; ++q) win[q] = tile[(rg * 4 + q) * TW + col];
        ox = x0 + col; oy = y0 + rg * 4;
    }
#pragma unroll
    for (int t = 0; t < K; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += tp[t] * win[j + t];
    float* dp = dst + (size_t)plane * h * w;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x = AXIS == 0 ? ox + j : ox, y = AXIS == 0 ? oy : oy + j;
        if (y < h && x < w) dp[(size_t)y * w + x] = acc[j];
    }
}

int filter2d_dispatch(const float* src, float* dst, const float* kern, int n, int c, int h, int w, int kh, int kw,
                      int per_sample, hipStream_t st) {
    if (!src || !dst || !kern || n <= 0 || c <= 0 || h <= 0 || w <= 0) return fail(RESR_ERR_ARG, "filter2d: bad argument");
    if (!(kh & 1) || !(kw & 1) || kh > 63 || kw > 63) return fail(RESR_ERR_ARG, "Wrong kernel size.");   // imgproc.py:1106
    if (kh / 2 >= h || kw / 2 >= w) return fail(RESR_ERR_ARG, "filter2d: reflect padding needs pad < image size");
    static const char* generic_env = getenv("RESR_FILTER_GENERIC");   // test knob: every size on the generic kernel
    if (kh == 21 && kw == 21 && !generic_env) {
        hipLaunchKernelGGL(filter2d21_kernel, dim3((w + 63) / 64, (h + 31) / 32, n * c), dim3(256), 0, st, src, dst, kern, c, h, w,
                           per_sample);
        RESR_CHECK_LAUNCH("filter2d21_kernel");
        return RESR_OK;
    }
    if (!per_sample && !generic_env && ((kh == 1 && kw == 51) || (kh == 51 && kw == 1))) {   // USMSharp(50, 0): imgproc.py:1514-1526
        const dim3 grid((w + 31) / 32, (h + 31) / 32, n * c);
        if (kh == 1) hipLaunchKernelGGL((filter1d_kernel<51, 0>), grid, dim3(256), 0, st, src, dst, kern, h, w);
        else hipLaunchKernelGGL((filter1d_kernel<51, 1>), grid, dim3(256), 0, st, src, dst, kern, h, w);
        RESR_CHECK_LAUNCH("filter1d_kernel");
        return RESR_OK;
    }
    const size_t lds = ((size_t)(32 + 2 * (kw / 2)) * (32 + 2 * (kh / 2)) + (size_t)kh * kw) * sizeof(float);
    hipLaunchKernelGGL(filter2d_kernel, dim3((w + 31) / 32, (h + 31) / 32, n * c), dim3(256), lds, st, src, dst, kern, c,
                       h, w, kh, kw, per_sample);
    RESR_CHECK_LAUNCH("filter2d_kernel");
    return RESR_OK;
}

// USMSharp epilogues
__global__ __launch_bounds__(256) void usm_mask_kernel(const float* __restrict__ x, const float* __restrict__ blur,
                                                       float* __restrict__ mask, long count, float threshold) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const float res = x[i] - blur[i];
    mask[i] = (fabsf(res) * 255.f > threshold) ? 1.f : 0.f;
}

__global__ __launch_bounds__(256) void usm_combine_kernel(const float* __restrict__ x, const float* __restrict__ blur,
                                                          const float* __restrict__ soft, float* __restrict__ out,
                                                          long count, float weight) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const float xv = x[i], res = xv - blur[i], s = soft[i];
    const float sharp = fminf(fmaxf(xv + weight * res, 0.f), 1.f);
    out[i] = s * sharp + (1.f - s) * xv;
}

// ---- USMSharp(50, 0).forward as TWO launches (round 5) -----------------------------------------------------------------
// The six passes above (row blur, column blur, mask, row blur, column blur, combine) move 15 fp32 planes = 60 B per value, and
// the column pass of a 32 x 32 tile fetches its 82-row window from HBM again for every tile column that lands on another XCD
// (measured 2.94 x its plane).  Here:
//   launch A  row pass -> column pass -> blur (fp32, kept: combine and backward read it) + the mask |x - blur| * 255 > threshold
//             as ONE BYTE per value;
//   launch B  the same two passes over the mask bytes = soft -> out = soft * clip(x + w (x - blur), 0, 1) + (1 - soft) x
//             (soft stored only when a backward pass will read it).
// 4 + 4 + 1 and 1 + 4 + 4 (+ 4) + 4 = 22 (26) B per value.  A workgroup owns a 64-column STRIP of one plane (or a vertical
// segment of it) and walks it downwards 32 rows at a time: every input row is loaded and row-passed ONCE into a ring of 128 rows in
// LDS, and a tile of 32 output rows is column-passed as soon as the 25 rows below it are in the ring -- 51 + 51 multiply-adds per
// value instead of the 2.56 x 51 + 51 of independent 32-row tiles (the kernel is bound by its vector and LDS work, not by HBM).
// Same taps in the same order as the separate passes (t = 0 .. 50, fused multiply-adds): within an ulp of them
// (tests/test_gpu_degrade.py), the goldens do not move.  Work is dealt to the XCDs in contiguous plane-major ranges (workgroup b ->
// range b % 8): the 50 halo columns a strip shares with its neighbours are then found in THAT XCD's L2 instead of being fetched
// from HBM once per XCD (a locality hint only: nothing depends on where a workgroup really runs).
template <typename TIN, int EPI>   // EPI 0: blur + byte mask (TIN = float);  EPI 1: soft mask + combine (TIN = uint8_t)
__global__ __launch_bounds__(256) void usm51_kernel(const TIN* __restrict__ src, const float* __restrict__ x, float* __restrict__ blur,
                                                    uint8_t* __restrict__ mask, float* __restrict__ soft, float* __restrict__ out,
                                                    const float* __restrict__ kern, int planes, int h, int w, int strips, int segs,
                                                    float threshold, float weight, int keep_soft) {
    constexpr int K = 51, R = 25, SW = 64, CH = 32, IW = SW + 2 * R, IWP = 120, RING = 128;   // 114-column input rows, padded for float4 windows
    __shared__ __attribute__((aligned(16))) float in[CH * IWP];      // 15 KB: the newest 32 input rows
    __shared__ __attribute__((aligned(16))) float ring[RING * SW];   // 32 KB: row-passed rows, slot = (row + 32) & 127
    __shared__ __attribute__((aligned(4))) uint8_t mb[EPI == 0 ? CH * SW : 4];
    const long items = (long)planes * segs * strips, per_xcd = (items + 7) / 8;
    const long item = (long)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if ((long)(blockIdx.x >> 3) >= per_xcd || item >= items) return;
    const int plane = (int)(item / ((long)segs * strips)), rem = (int)(item - (long)plane * segs * strips);
    const int seg = rem / strips, strip = rem - seg * strips;
    const int tiles_y = (h + CH - 1) / CH;
    const int t_begin = (int)((long)tiles_y * seg / segs), t_end = (int)((long)tiles_y * (seg + 1) / segs);   // this segment's output tiles
    const int x0 = strip * SW;
    const TIN* sp = src + (size_t)plane * h * w;
    for (int i = threadIdx.x; i < CH * (IWP - IW); i += 256) in[(i / (IWP - IW)) * IWP + IW + i % (IWP - IW)] = 0.f;
    // the 51 taps are wave-uniform: read through the scalar cache they live in SGPRs (an FMA takes one scalar operand) and leave
    // the vector registers to the windows and to the next chunk's prefetch
    float tp[K];
#pragma unroll
    for (int q = 0; q < K; ++q) tp[q] = kern[q];
    // the next chunk's input elements of this thread (global loads in flight under the column pass of the current tile)
    constexpr int NL = (CH * IW + 255) / 256;
    float pre[NL];
    auto fetch = [&](int k) {
        const int r0 = 32 * k - R;
#pragma unroll
        for (int jj = 0; jj < NL; ++jj) {
            const int i = threadIdx.x + 256 * jj;
            const int rr = i / IW, c = i - rr * IW;
            const int iy = reflect(r0 + rr, h), ix = reflect(x0 + c - R, w);
            // (rows / columns whose reflection still falls outside only feed outputs beyond the image, which are never stored)
            pre[jj] = (i < CH * IW && iy >= 0 && iy < h && ix >= 0 && ix < w) ? (float)sp[(size_t)iy * w + ix] : 0.f;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int jj = 0; jj < NL; ++jj) {
            const int i = threadIdx.x + 256 * jj;
            const int rr = i / IW, c = i - rr * IW;
            if (i < CH * IW) in[rr * IWP + c] = pre[jj];
        }
    };
    fetch(t_begin);
    // Output tile t (rows 32 t .. 32 t + 31) needs input rows 32 t - 25 .. 32 t + 56.  Chunk k = input rows 32 k - 25 .. 32 k + 6,
    // so tile t is complete after chunk t + 2 (rows up to 32 t + 70), and the ring's 128 slots hold tile t's 82 rows next to the
    // 32 rows of the chunk being written (rows 128 apart share a slot: the row a new one replaces is older than 32 t - 25).
    for (int k = t_begin; k < t_end + 2; ++k) {
        const int r0 = 32 * k - R;
        commit();                              // (`in` was last read by the previous chunk's row pass, two barriers ago)
        __syncthreads();
        // row pass of the 32 new rows: item = (row, group of 4 columns); window of 54 (read as 56) values, 4 x 51 FMAs
        for (int i = threadIdx.x; i < CH * (SW / 4); i += 256) {
            const int rr = i / (SW / 4), cg = i - rr * (SW / 4);
            float win[56];
            const float4* wp = reinterpret_cast<const float4*>(in + rr * IWP + cg * 4);
#pragma unroll
            for (int q = 0; q < 14; ++q) {
                const float4 v = wp[q];
                win[q * 4] = v.x; win[q * 4 + 1] = v.y; win[q * 4 + 2] = v.z; win[q * 4 + 3] = v.w;
            }
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < K; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] += tp[t] * win[j + t];
            *reinterpret_cast<float4*>(ring + ((r0 + rr + 32) & (RING - 1)) * SW + cg * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
        if (k + 1 < t_end + 2) fetch(k + 1);
        __syncthreads();
        const int t = k - 2;                    // the tile whose last rows have just arrived
        if (t < t_begin) continue;              // (uniform: the segment's first two chunks only fill the ring)
        const int y0 = t * CH;
        // column pass: item = (group of 4 rows, column); lanes walk the columns (conflict-free), window of 54 ring rows
        for (int i = threadIdx.x; i < (CH / 4) * SW; i += 256) {
            const int rg = i / SW, col = i - rg * SW;
            // the epilogue's global operands first: their round trip runs under the window reads and the 204 multiply-adds
            float xg[4], bg[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gy = y0 + rg * 4 + j;
                const bool ok = gy < h && x0 + col < w;
                const size_t q = ((size_t)plane * h + gy) * w + x0 + col;
                xg[j] = ok ? x[q] : 0.f;
                bg[j] = (EPI == 1 && ok) ? blur[q] : 0.f;
            }
            float win[54];
#pragma unroll
            for (int q = 0; q < 54; ++q) win[q] = ring[((y0 + rg * 4 + q - R + 32) & (RING - 1)) * SW + col];
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tt = 0; tt < K; ++tt)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] += tp[tt] * win[j + tt];
            const int gx = x0 + col;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ly = rg * 4 + j, gy = y0 + ly;
                const bool ok = gy < h && gx < w;
                const size_t q = ((size_t)plane * h + gy) * w + gx;
                if constexpr (EPI == 0) {
                    const float xv = xg[j];
                    mb[ly * SW + col] = (fabsf(xv - acc[j]) * 255.f > threshold) ? 1 : 0;
                    if (ok) blur[q] = acc[j];
                } else if (ok) {
                    const float xv = xg[j], sv = acc[j];
                    const float sharp = fminf(fmaxf(xv + weight * (xv - bg[j]), 0.f), 1.f);
                    if (keep_soft) soft[q] = sv;          // only the backward pass reads it (the degradation path runs without a graph)
                    out[q] = sv * sharp + (1.f - sv) * xv;
                }
            }
        }
        if constexpr (EPI == 0) {   // the tile's mask bytes leave as 4-byte words where the row allows it
            __syncthreads();
            for (int i = threadIdx.x; i < CH * SW / 4; i += 256) {
                const int ly = i / (SW / 4), c4 = (i - ly * (SW / 4)) * 4;
                const int gy = y0 + ly, gx = x0 + c4;
                if (gy >= h || gx >= w) continue;
                uint8_t* mp = mask + ((size_t)plane * h + gy) * w + gx;
                if (gx + 3 < w && (((size_t)mp) & 3) == 0) *reinterpret_cast<uint32_t*>(mp) = *reinterpret_cast<const uint32_t*>(mb + ly * SW + c4);
                else for (int e = 0; e < 4 && gx + e < w; ++e) mp[e] = mb[ly * SW + c4 + e];
            }
        }
        // (the next chunk's loads overwrite `in`, its row pass writes ring slots no pending column pass reads, and the barrier
        // behind the loads orders both against this tile's reads of `ring` / `mb`)
    }
}

int usm_dispatch(const float* src, float* dst, float* tmp, const float* k1d, int ksize, float weight, float threshold,
                 int n, int c, int h, int w, hipStream_t st, int keep_for_backward) {
    if (!src || !dst || !tmp || !k1d) return fail(RESR_ERR_ARG, "usm_sharp: null argument");
    const long count = (long)n * c * h * w;
    const char* six_env = getenv("RESR_USM_SIX_PASSES");   // A/B and test knob (read per call): the separate passes
    if (ksize == 51 && h > 25 && w > 25 && !six_env && (long)n * c < (1L << 24)) {
        uint8_t* mask = reinterpret_cast<uint8_t*>(tmp);   // tmp3 = [mask bytes (+ unused) | blur | soft]
        float* blur = tmp + count;
        float* soft = tmp + 2 * count;
        // workgroups: planes x vertical segments x 64-column strips; a segment pays two extra 32-row chunks of row pass to fill its
        // ring, so segments are only cut until the launch has ~2 workgroups per CU (three fit a CU's LDS)
        const int strips = (w + 63) / 64, tiles_y = (h + 31) / 32;
        int segs = 1;
        while (segs < 8 && (long)n * c * strips * segs < 512 && tiles_y / (segs * 2) >= 4) segs *= 2;
        const long items = (long)n * c * strips * segs;
        const unsigned grid = (unsigned)(((items + 7) / 8) * 8);
        hipLaunchKernelGGL((usm51_kernel<float, 0>), dim3(grid), dim3(256), 0, st, src, src, blur, mask, (float*)nullptr, (float*)nullptr, k1d,
                           n * c, h, w, strips, segs, threshold, weight, 0);
        RESR_CHECK_LAUNCH("usm51_kernel (blur + mask)");
        hipLaunchKernelGGL((usm51_kernel<uint8_t, 1>), dim3(grid), dim3(256), 0, st, (const uint8_t*)mask, src, blur, (uint8_t*)nullptr, soft, dst, k1d,
                           n * c, h, w, strips, segs, threshold, weight, keep_for_backward);
        RESR_CHECK_LAUNCH("usm51_kernel (soft + combine)");
        return RESR_OK;
    }
    float* t0 = tmp;               // row pass
    float* blur = tmp + count;     // blurred x, later soft mask input
    float* t2 = tmp + 2 * count;   // mask / soft
    int rc;
    if ((rc = filter2d_dispatch(src, t0, k1d, n, c, h, w, 1, ksize, 0, st))) return rc;
    if ((rc = filter2d_dispatch(t0, blur, k1d, n, c, h, w, ksize, 1, 0, st))) return rc;
    const unsigned blocks = (unsigned)((count + 255) / 256);
    hipLaunchKernelGGL(usm_mask_kernel, dim3(blocks), dim3(256), 0, st, src, blur, t2, count, threshold);
    RESR_CHECK_LAUNCH("usm_mask_kernel");
    if ((rc = filter2d_dispatch(t2, t0, k1d, n, c, h, w, 1, ksize, 0, st))) return rc;
    if ((rc = filter2d_dispatch(t0, t2, k1d, n, c, h, w, ksize, 1, 0, st))) return rc;
    hipLaunchKernelGGL(usm_combine_kernel, dim3(blocks), dim3(256), 0, st, src, blur, t2, dst, count, weight);
    RESR_CHECK_LAUNCH("usm_combine_kernel");
    return RESR_OK;
}

// ---- USMSharp backward (the reference's GAN step differentiates through usm_sharpener(sr), train_realesrgan.py:476) --
// forward: out = soft*clip(x + w*(x - Bx), 0, 1) + (1-soft)*x with soft piecewise constant in x.
//   a  = soft * 1[0 <= x + w*(x-Bx) <= 1] * g
//   gx = (1-soft)*g + (1+w)*a - w * B^T a          B^T = adjoint of (reflect pad + separable Gaussian)
__global__ __launch_bounds__(256) void usm_bwd_prep_kernel(const float* __restrict__ x, const float* __restrict__ blur,
                                                           const float* __restrict__ soft, const float* __restrict__ g,
                                                           float* __restrict__ a, float* __restrict__ b, long count, float weight) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const float xv = x[i], s = soft[i], gv = g[i];
    const float pre = xv + weight * (xv - blur[i]);
    const float av = (pre >= 0.f && pre <= 1.f) ? s * gv : 0.f;
    a[i] = av;
    b[i] = (1.f - s) * gv + (1.f + weight) * av;
}

// adjoint of a reflect-padded 1-D correlation along x (axis 0) or y (axis 1):
//   z[j] = sum_t k[t] * g0[j - t + r] (g0 = g, zero outside), gx[p] = z[p] + z[-p] (1<=p<=r) + z[2(n-1)-p] (n-1-r<=p<=n-2)
__global__ __launch_bounds__(256) void filter1d_adjoint_kernel(const float* __restrict__ g, float* __restrict__ out,
                                                               const float* __restrict__ k, int planes, int h, int w, int ksize,
                                                               int axis, const float* __restrict__ sub_from, float sub_scale) {
    const long total = (long)planes * h * w;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % w), y = (int)((i / w) % h);
    const long pbase = (i / ((long)w * h)) * (long)h * w;
    const int r = ksize / 2, n = axis == 0 ? w : h, p = axis == 0 ? x : y;
    const long stride = axis == 0 ? 1 : w;
    const float* line = g + pbase + (axis == 0 ? (long)y * w : x);
    auto z = [&](int j) {
        float s = 0.f;
        for (int t = 0; t < ksize; ++t) {
            const int q = j - t + r;
            if (q >= 0 && q < n) s += k[t] * line[(long)q * stride];
        }
        return s;
    };
    float v = z(p);
    if (p >= 1 && p <= r) v += z(-p);
    if (p <= n - 2 && p >= n - 1 - r) v += z(2 * (n - 1) - p);
    out[i] = sub_from ? sub_from[i] - sub_scale * v : v;
}

int usm_bwd_dispatch(const float* x, const float* saved, const float* g, float* gx, float* tmp2, const float* k1d, int ksize,
                     float weight, int n, int c, int h, int w, hipStream_t st) {
    if (!x || !saved || !g || !gx || !tmp2 || !k1d) return fail(RESR_ERR_ARG, "usm_sharp_bwd: null argument");
    const long count = (long)n * c * h * w;
    const float* blur = saved + count;       // layout of resr_usm_sharp's tmp3: [row pass | blur | soft]
    const float* soft = saved + 2 * count;
    float* a = tmp2;
    float* b = tmp2 + count;
    const unsigned blocks = (unsigned)((count + 255) / 256);
    hipLaunchKernelGGL(usm_bwd_prep_kernel, dim3(blocks), dim3(256), 0, st, x, blur, soft, g, a, b, count, weight);
    // B^T a: vertical adjoint then horizontal adjoint (forward applied horizontal then vertical)
    hipLaunchKernelGGL(filter1d_adjoint_kernel, dim3(blocks), dim3(256), 0, st, a, gx, k1d, n * c, h, w, ksize, 1, (const float*)nullptr, 0.f);
    hipLaunchKernelGGL(filter1d_adjoint_kernel, dim3(blocks), dim3(256), 0, st, gx, a, k1d, n * c, h, w, ksize, 0, (const float*)b, weight);
    if (hipMemcpyAsync(gx, a, count * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return fail(RESR_ERR_LAUNCH, "usm_sharp_bwd: copy failed");
    RESR_CHECK_LAUNCH("usm_bwd kernels");
    return RESR_OK;
}

// ---------------------------------------------------------------------------------------------------------
// resize (torch.nn.functional.interpolate, align_corners=False, antialias=False)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

__global__ __launch_bounds__(256) void resize_kernel(const float* __restrict__ src, float* __restrict__ dst, int planes,
                                                     int h, int w, int oh, int ow, int mode, float sh, float sw) {
    const long total = (long)planes * oh * ow;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ox = (int)(i % ow);
    const int oy = (int)((i / ow) % oh);
    const int p = (int)(i / ((long)ow * oh));
    const float* sp = src + (size_t)p * h * w;
    float v;
    if (mode == 0) {            // area == adaptive_avg_pool2d
        const int y0 = (int)floorf((float)(oy * h) / oh), y1 = (int)ceilf((float)((oy + 1) * h) / oh);
        const int x0 = (int)floorf((float)(ox * w) / ow), x1 = (int)ceilf((float)((ox + 1) * w) / ow);
        float s = 0.f;
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) s += sp[(size_t)y * w + x];
        v = s / (float)((y1 - y0) * (x1 - x0));
    } else if (mode == 1) {     // bilinear
        float fy = sh * (oy + 0.5f) - 0.5f, fx = sw * (ox + 0.5f) - 0.5f;
        if (fy < 0.f) fy = 0.f;
        if (fx < 0.f) fx = 0.f;
        const int y0 = (int)fy, x0 = (int)fx;
        const int yp = y0 < h - 1 ? 1 : 0, xp = x0 < w - 1 ? 1 : 0;
        const float ly = fy - y0, lx = fx - x0, hy = 1.f - ly, hx = 1.f - lx;
        const float* r0 = sp + (size_t)y0 * w + x0;
        const float* r1 = r0 + (size_t)yp * w;
        v = hy * (hx * r0[0] + lx * r0[xp]) + ly * (hx * r1[0] + lx * r1[xp]);
    } else {                    // bicubic, A = -0.75
        const float A = -0.75f;
        const float fy = sh * (oy + 0.5f) - 0.5f, fx = sw * (ox + 0.5f) - 0.5f;
        const int iy = (int)floorf(fy), ix = (int)floorf(fx);
        const float ty = fy - iy, tx = fx - ix;
        float cy[4] = {cubic2(ty + 1.f, A), cubic1(ty, A), cubic1(1.f - ty, A), cubic2(2.f - ty, A)};
        float cx[4] = {cubic2(tx + 1.f, A), cubic1(tx, A), cubic1(1.f - tx, A), cubic2(2.f - tx, A)};
        v = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int yy = min(max(iy - 1 + a, 0), h - 1);
            float row = 0.f;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int xx = min(max(ix - 1 + b, 0), w - 1);
                row += cx[b] * sp[(size_t)yy * w + xx];
            }
            v += cy[a] * row;
        }
    }
    dst[i] = v;
}

int resize_dispatch(const float* src, float* dst, int n, int c, int h, int w, int oh, int ow, int mode, double scale_h,
                    double scale_w, hipStream_t st) {
    if (!src || !dst || n <= 0 || c <= 0 || h <= 0 || w <= 0 || oh <= 0 || ow <= 0 || mode < 0 || mode > 2)
        return fail(RESR_ERR_ARG, "resize: bad argument");
    // coordinate scale: 1/scale_factor when the caller gave scale_factor= (torch area_pixel_compute_scale),
    // in/out otherwise
    const float sh = scale_h > 0. ? (float)(1.0 / scale_h) : (float)h / (float)oh;
    const float sw = scale_w > 0. ? (float)(1.0 / scale_w) : (float)w / (float)ow;
    const long total = (long)n * c * oh * ow;
    hipLaunchKernelGGL(resize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, dst, n * c, h, w, oh,
                       ow, mode, sh, sw);
    RESR_CHECK_LAUNCH("resize_kernel");
    return RESR_OK;
}

// ---------------------------------------------------------------------------------------------------------
// RNG: Philox4x32-10, counter = (element index, stream), key = seed
// ---------------------------------------------------------------------------------------------------------
struct Philox {
    uint32_t c[4], k[2];
    __device__ Philox(uint64_t seed, uint64_t idx, uint64_t stream) {
        k[0] = (uint32_t)seed; k[1] = (uint32_t)(seed >> 32);
        c[0] = (uint32_t)idx; c[1] = (uint32_t)(idx >> 32); c[2] = (uint32_t)stream; c[3] = (uint32_t)(stream >> 32);
    }
    __device__ void next(uint32_t out[4]) {
        uint32_t c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], k0 = k[0], k1 = k[1];
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
            const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
            c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
        out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
        if (++c[3] == 0) ++c[2];   // advance the stream word: independent draws for the same element
    }
};

__device__ __forceinline__ float u01(uint32_t x) { return (x >> 8) * (1.0f / 16777216.0f) + (0.5f / 16777216.0f); }  // (0,1)

__global__ __launch_bounds__(256) void randn_kernel(float* __restrict__ dst, long count, uint64_t seed, uint64_t stream) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;     // 4 normals per thread
    if (q * 4 >= count) return;
    Philox ph(seed, (uint64_t)q, stream);
    uint32_t r[4];
    ph.next(r);
    float z[4];
    const float a = sqrtf(-2.f * logf(u01(r[0]))), b = 6.283185307179586f * u01(r[1]);
    const float c = sqrtf(-2.f * logf(u01(r[2]))), d = 6.283185307179586f * u01(r[3]);
    z[0] = a * cosf(b); z[1] = a * sinf(b); z[2] = c * cosf(d); z[3] = c * sinf(d);
    for (int j = 0; j < 4 && q * 4 + j < count; ++j) dst[q * 4 + j] = z[j];
}

int randn_dispatch(float* dst, long count, uint64_t seed, uint64_t stream, hipStream_t st) {
    if (!dst || count <= 0) return fail(RESR_ERR_ARG, "randn_fill: bad argument");
    hipLaunchKernelGGL(randn_kernel, dim3((unsigned)((count / 4 + 256) / 256)), dim3(256), 0, st, dst, count, seed, stream);
    RESR_CHECK_LAUNCH("randn_kernel");
    return RESR_OK;
}

// ---------------------------------------------------------------------------------------------------------
// noise
// ---------------------------------------------------------------------------------------------------------
// clip/rounds combinations of imgproc.py:1050-1055: bit0 = clip, bit1 = rounds
__device__ __forceinline__ float finish_noise(float v, int mode) {
    if (mode == 3) return fminf(fmaxf(rintf(v * 255.f), 0.f), 255.f) / 255.f;
    if (mode == 1) return fminf(fmaxf(v, 0.f), 1.f);
    if (mode == 2) return rintf(v * 255.f) / 255.f;
    return v;
}

__global__ __launch_bounds__(256) void gauss_noise_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                          const float* __restrict__ sigma, const float* __restrict__ gray,
                                                          const float* __restrict__ fg, const float* __restrict__ fc,
                                                          int c, int hw, long count, int clip) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const int b = (int)(i / ((long)c * hw));
    const int px = (int)(i % hw);
    const float s = sigma[b], g = gray[b];
    float noise = fc[i] * s / 255.f;
    if (fg) noise = noise * (1.f - g) + (fg[px] * s / 255.f) * g;     // imgproc.py:861, one gray field for the batch
    dst[i] = finish_noise(src[i] + noise, clip);
}

int gauss_noise_dispatch(const float* src, float* dst, const float* sigma, const float* gray, const float* fg,
                         const float* fc, int n, int c, int h, int w, int clip, hipStream_t st) {
    if (!src || !dst || !sigma || !gray || !fc) return fail(RESR_ERR_ARG, "noise_gaussian: null argument");
    const long count = (long)n * c * h * w;
    hipLaunchKernelGGL(gauss_noise_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, src, dst, sigma, gray,
                       fg, fc, c, h * w, count, clip);
    RESR_CHECK_LAUNCH("gauss_noise_kernel");
    return RESR_OK;
}

__device__ __forceinline__ float q255(float v) { return fminf(fmaxf(rintf(v * 255.f), 0.f), 255.f); }
__device__ __forceinline__ float gray_of(float r, float g, float b) { return 0.2989f * r + 0.587f * g + 0.114f * b; }

// presence bitmaps of the quantised values per sample: flags[b][0][256] colour, flags[b][1][256] gray
__global__ __launch_bounds__(256) void unique_flags_kernel(const float* __restrict__ src, unsigned* __restrict__ flags,
                                                           int hw) {
    __shared__ unsigned present[512];
    const int b = blockIdx.y;
    present[threadIdx.x] = 0;
    present[256 + threadIdx.x] = 0;
    __syncthreads();
    const float* sp = src + (size_t)b * 3 * hw;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < hw; p += gridDim.x * 256) {
        const float r = sp[p], g = sp[hw + p], bl = sp[2 * hw + p];
        present[(int)q255(r)] = 1; present[(int)q255(g)] = 1; present[(int)q255(bl)] = 1;
        present[256 + (int)q255(gray_of(r, g, bl))] = 1;
    }
    __syncthreads();
    if (present[threadIdx.x]) flags[(size_t)b * 512 + threadIdx.x] = 1;
    if (present[256 + threadIdx.x]) flags[(size_t)b * 512 + 256 + threadIdx.x] = 1;
}

// vals[b][0] colour, vals[b][1] gray: 2^ceil(log2(#unique))
__global__ void unique_vals_kernel(const unsigned* __restrict__ flags, float* __restrict__ vals) {
    __shared__ int cnt[2];
    if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int b = blockIdx.x;
    if (flags[(size_t)b * 512 + threadIdx.x]) atomicAdd(&cnt[0], 1);
    if (flags[(size_t)b * 512 + 256 + threadIdx.x]) atomicAdd(&cnt[1], 1);
    __syncthreads();
    if (threadIdx.x < 2) {
        int v = 1;
        while (v < cnt[threadIdx.x]) v <<= 1;
        vals[b * 2 + threadIdx.x] = (float)v;
    }
}

// Poisson sampler: inversion for small means, Hormann's PTRS (transformed rejection) otherwise -- the same
// two regimes torch.poisson uses.
__device__ float poisson_draw(float lam, Philox& ph) {
    uint32_t r[4];
    if (lam <= 0.f) return 0.f;
    if (lam < 10.f) {
        const float el = expf(-lam);
        float prod = 1.f;
        int k = 0;
        for (;;) {
            ph.next(r);
            for (int j = 0; j < 4; ++j) {
                prod *= u01(r[j]);
                if (prod <= el) return (float)k;
                ++k;
            }
        }
    }
    const float slam = sqrtf(lam), loglam = logf(lam);
    const float b = 0.931f + 2.53f * slam, a = -0.059f + 0.02483f * b;
    const float invalpha = 1.1239f + 1.1328f / (b - 3.4f), vr = 0.9277f - 3.6224f / (b - 2.f);
    for (;;) {
        ph.next(r);
        for (int j = 0; j < 4; j += 2) {
            const float U = u01(r[j]) - 0.5f, V = u01(r[j + 1]);
            const float us = 0.5f - fabsf(U);
            const float k = floorf((2.f * a / us + b) * U + lam + 0.43f);
            if (us >= 0.07f && V <= vr) return k;
            if (k < 0.f || (us < 0.013f && V > us)) continue;
            if (logf(V) + logf(invalpha) - logf(a / (us * us) + b) <= -lam + k * loglam - lgammaf(k + 1.f)) return k;
        }
    }
}

__global__ __launch_bounds__(256) void poisson_noise_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                            const float* __restrict__ scale, const float* __restrict__ gray,
                                                            const float* __restrict__ vals, int hw, long npix,
                                                            uint64_t seed, int clip) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;      // one pixel (3 channels) per thread
    if (i >= npix) return;
    const int b = (int)(i / hw);
    const int p = (int)(i % hw);
    const float* sp = src + (size_t)b * 3 * hw + p;
    float* dp = dst + (size_t)b * 3 * hw + p;
    const float ch[3] = {sp[0], sp[hw], sp[2 * hw]};
    const float sc = scale[b], g = gray[b];
    Philox ph(seed, (uint64_t)i, 0);
    float ng = 0.f;
    if (g > 0.f) {                                               // imgproc.py:887-897
        const float vg = vals[b * 2 + 1];
        const float gq = q255(gray_of(ch[0], ch[1], ch[2])) / 255.f;
        ng = poisson_draw(gq * vg, ph) / vg - gq;
    }
    const float v = vals[b * 2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {                                // imgproc.py:899-914
        const float q = q255(ch[k]) / 255.f;
        float noise = poisson_draw(q * v, ph) / v - q;
        noise = noise * (1.f - g) + ng * g;
        dp[k * hw] = finish_noise(ch[k] + noise * sc, clip);
    }
}

int poisson_noise_dispatch(const float* src, float* dst, const float* scale, const float* gray, uint64_t seed,
                           void* workspace, int n, int c, int h, int w, int clip, hipStream_t st) {
    if (!src || !dst || !scale || !gray || !workspace) return fail(RESR_ERR_ARG, "noise_poisson: null argument");
    if (c != 3) return fail(RESR_ERR_ARG, "noise_poisson: needs 3 channels");
    unsigned* flags = (unsigned*)workspace;                       // n*512 uint32, then n*2 float
    float* vals = (float*)(flags + (size_t)n * 512);
    if (hipMemsetAsync(flags, 0, (size_t)n * 512 * sizeof(unsigned), st) != hipSuccess)
        return fail(RESR_ERR_LAUNCH, "noise_poisson: memset failed");
    const int hw = h * w;
    int bx = (hw + 255) / 256;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(unique_flags_kernel, dim3(bx, n), dim3(256), 0, st, src, flags, hw);
    RESR_CHECK_LAUNCH("unique_flags_kernel");
    hipLaunchKernelGGL(unique_vals_kernel, dim3(n), dim3(256), 0, st, flags, vals);
    RESR_CHECK_LAUNCH("unique_vals_kernel");
    const long npix = (long)n * hw;
    hipLaunchKernelGGL(poisson_noise_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, src, dst, scale, gray,
                       vals, hw, npix, seed, clip);
    RESR_CHECK_LAUNCH("poisson_noise_kernel");
    return RESR_OK;
}

// ---------------------------------------------------------------------------------------------------------
// DiffJPEG(differentiable=False)
// ---------------------------------------------------------------------------------------------------------
struct JpegTables {
    float basis[4096];    // [x][y][u][v] = cos((2x+1)u pi/16) cos((2y+1)v pi/16), float32 of the float64 product
    float scale[64];      // 0.25 * alpha(u) alpha(v)
    float alpha[64];      // alpha(u) alpha(v)
    float ytab[64], ctab[64];   // the reference's (transposed) tables, imgproc.py:40-49
};
__device__ JpegTables g_jpeg;

static void jpeg_tables_host(JpegTables& t) {
    static const float ystd[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
                                   14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
                                   49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
    static const float cstd[16] = {17, 18, 24, 47, 18, 21, 26, 66, 24, 26, 56, 99, 47, 66, 99, 99};
    for (int x = 0; x < 8; ++x)
        for (int y = 0; y < 8; ++y)
            for (int u = 0; u < 8; ++u)
                for (int v = 0; v < 8; ++v)
                    t.basis[((x * 8 + y) * 8 + u) * 8 + v] = (float)(cos((2 * x + 1) * u * M_PI / 16) * cos((2 * y + 1) * v * M_PI / 16));
    for (int u = 0; u < 8; ++u)
        for (int v = 0; v < 8; ++v) {
            const double au = u == 0 ? 1.0 / sqrt(2.0) : 1.0, av = v == 0 ? 1.0 / sqrt(2.0) : 1.0;
            t.scale[u * 8 + v] = (float)(au * av * 0.25);
            t.alpha[u * 8 + v] = (float)(au * av);
            t.ytab[u * 8 + v] = ystd[v * 8 + u];                                   // transposed
            t.ctab[u * 8 + v] = (u < 4 && v < 4) ? cstd[v * 4 + u] : 99.f;          // transposed 4x4 corner
        }
}

// one workgroup per 16x16 macroblock: 6 blocks of 8x8 (4 Y, Cb, Cr) -> 384 coefficients
__global__ __launch_bounds__(256) void jpeg_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                   const float* __restrict__ quality, float* __restrict__ coeffs, int h,
                                                   int w, int mbx, int mby, int clamp_in) {
    __shared__ float ycc[3][256];      // Y, Cb, Cr at full resolution
    __shared__ float blk[6][64];       // level-shifted blocks
    __shared__ float coef[6][64];      // dequantised coefficients * alpha
    __shared__ float rec[6][64];
    const int b = blockIdx.z, my = blockIdx.y, mx = blockIdx.x;
    const int t = threadIdx.x, ly = t >> 4, lx = t & 15;
    const int y = my * 16 + ly, x = mx * 16 + lx;
    const size_t hw = (size_t)h * w;
    const float* sp = src + (size_t)b * 3 * hw;
    float r = 0.f, g = 0.f, bl = 0.f;                 // zero padding up to a multiple of 16 (imgproc.py:1488)
    if (y < h && x < w) {
        r = sp[(size_t)y * w + x]; g = sp[hw + (size_t)y * w + x]; bl = sp[2 * hw + (size_t)y * w + x];
        if (clamp_in) {            // torch.clamp(out, 0, 1) of train_realesrnet.py:308,357,362 folded in
            r = fminf(fmaxf(r, 0.f), 1.f); g = fminf(fmaxf(g, 0.f), 1.f); bl = fminf(fmaxf(bl, 0.f), 1.f);
        }
        r *= 255.f; g *= 255.f; bl *= 255.f;
    }
    ycc[0][t] = 0.299f * r + 0.587f * g + 0.114f * bl + 0.f;
    ycc[1][t] = -0.168736f * r + -0.331264f * g + 0.5f * bl + 128.f;
    ycc[2][t] = 0.5f * r + -0.418688f * g + -0.081312f * bl + 128.f;
    __syncthreads();
    // block split: Y block (ly>>3, lx>>3); chroma = 2x2 average
    blk[(ly >> 3) * 2 + (lx >> 3)][(ly & 7) * 8 + (lx & 7)] = ycc[0][t] - 128.f;
    if (t < 128) {
        const int cidx = t >> 6, e = t & 63, cy = e >> 3, cx = e & 7;
        const float* pc = ycc[1 + cidx];
        const float s = (pc[(2 * cy) * 16 + 2 * cx] + pc[(2 * cy) * 16 + 2 * cx + 1] + pc[(2 * cy + 1) * 16 + 2 * cx] +
                         pc[(2 * cy + 1) * 16 + 2 * cx + 1]) * 0.25f;
        blk[4 + cidx][e] = s - 128.f;
    }
    __syncthreads();
    const float qv = quality[b];
    const float factor = qv < 50.f ? (5000.f / qv) / 100.f : (200.f - qv * 2.f) / 100.f;   // imgproc.py:1134-1139
    for (int i = t; i < 384; i += 256) {
        const int bi = i >> 6, uv = i & 63;
        float s = 0.f;
#pragma unroll 8
        for (int xy = 0; xy < 64; ++xy) s += blk[bi][xy] * g_jpeg.basis[xy * 64 + uv];
        const float table = (bi < 4 ? g_jpeg.ytab[uv] : g_jpeg.ctab[uv]) * factor;
        const float q = rintf((g_jpeg.scale[uv] * s) / table);                                 // torch.round: half to even
        if (coeffs) {
            // layout: per image [Y blocks (H/8 * W/8) | Cb blocks | Cr blocks] x 64, block order row-major
            const int nyb = mbx * 2 * mby * 2, ncb = mbx * mby;
            size_t off;
            if (bi < 4) off = (size_t)((my * 2 + (bi >> 1)) * (mbx * 2) + mx * 2 + (bi & 1));
            else off = (size_t)nyb + (size_t)(bi - 4) * ncb + (size_t)my * mbx + mx;
            coeffs[((size_t)b * (nyb + 2 * ncb) + off) * 64 + uv] = q;
        }
        coef[bi][uv] = (q * table) * g_jpeg.alpha[uv];
    }
    __syncthreads();
    for (int i = t; i < 384; i += 256) {
        const int bi = i >> 6, uv = i & 63;      // here (u,v) index the spatial position of the output
        float s = 0.f;
#pragma unroll 8
        for (int xy = 0; xy < 64; ++xy) s += coef[bi][xy] * g_jpeg.basis[uv * 64 + xy];   // basis^T: [u][v][x][y]
        rec[bi][uv] = 0.25f * s + 128.f;
    }
    __syncthreads();
    if (y < h && x < w) {
        const float Y = rec[(ly >> 3) * 2 + (lx >> 3)][(ly & 7) * 8 + (lx & 7)];
        const float cb = rec[4][(ly >> 1) * 8 + (lx >> 1)] - 128.f, cr = rec[5][(ly >> 1) * 8 + (lx >> 1)] - 128.f;
        float* dp = dst + (size_t)b * 3 * hw + (size_t)y * w + x;
        const float R = Y + 1.402f * cr;
        const float G = Y + -0.344136f * cb + -0.714136f * cr;
        const float B = Y + 1.772f * cb;
        dp[0] = fminf(fmaxf(R, 0.f), 255.f) / 255.f;
        dp[hw] = fminf(fmaxf(G, 0.f), 255.f) / 255.f;
        dp[2 * hw] = fminf(fmaxf(B, 0.f), 255.f) / 255.f;
    }
}

int jpeg_dispatch(const float* src, float* dst, const float* quality, float* coeffs, int n, int h, int w, int flags,
                  hipStream_t st) {
    if (!src || !dst || !quality || n <= 0 || h <= 0 || w <= 0) return fail(RESR_ERR_ARG, "jpeg: bad argument");
    static std::once_flag once[16];
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipError_t err = hipSuccess;
    std::call_once(once[dev & 15], [&]() {
        static JpegTables host;
        jpeg_tables_host(host);
        err = hipMemcpyToSymbol(HIP_SYMBOL(g_jpeg), &host, sizeof(JpegTables));
    });
    if (err != hipSuccess) return fail(RESR_ERR_LAUNCH, "jpeg: table upload failed: %s", hipGetErrorString(err));
    const int mbx = (w + 15) / 16, mby = (h + 15) / 16;
    hipLaunchKernelGGL(jpeg_kernel, dim3(mbx, mby, n), dim3(256), 0, st, src, dst, quality, coeffs, h, w, mbx, mby, flags & 1);
    RESR_CHECK_LAUNCH("jpeg_kernel");
    return RESR_OK;
}

// ---------------------------------------------------------------------------------------------------------
// final quantise + crop
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void crop_kernel(const float* __restrict__ src, float* __restrict__ dst, int planes, int h,
                                                   int w, int size, int top, int left, int quant) {
    const long total = (long)planes * size * size;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % size), y = (int)((i / size) % size), p = (int)(i / ((long)size * size));
    float v = src[((size_t)p * h + top + y) * w + left + x];
    if (quant) v = fminf(fmaxf(rintf(v * 255.f), 0.f), 255.f) / 255.f;     // train_realesrnet.py:374
    dst[i] = v;
}

int quantize_crop_dispatch(const float* lr, const float* hr, float* lr_out, float* hr_out, int n, int c, int lr_h, int lr_w,
                           int hr_h, int hr_w, int hr_size, int upscale, int hr_top, int hr_left, hipStream_t st) {
    if (!lr || !hr || !lr_out || upscale <= 0) return fail(RESR_ERR_ARG, "quantize_crop: bad argument");
    if (!hr_out && (hr_top != 0 || hr_left != 0 || hr_size != hr_h || hr_size != hr_w))
        return fail(RESR_ERR_ARG, "quantize_crop: hr_out may only be NULL when the HR window is the whole image (the caller keeps using hr)");
    const int lr_size = hr_size / upscale, lr_top = hr_top / upscale, lr_left = hr_left / upscale;     // imgproc.py:1917-1919
    if (hr_top < 0 || hr_left < 0 || hr_top + hr_size > hr_h || hr_left + hr_size > hr_w || lr_top + lr_size > lr_h ||
        lr_left + lr_size > lr_w)
        return fail(RESR_ERR_ARG, "quantize_crop: window outside the image");
    const long tl = (long)n * c * lr_size * lr_size, th = (long)n * c * hr_size * hr_size;
    hipLaunchKernelGGL(crop_kernel, dim3((unsigned)((tl + 255) / 256)), dim3(256), 0, st, lr, lr_out, n * c, lr_h, lr_w, lr_size,
                       lr_top, lr_left, 1);
    RESR_CHECK_LAUNCH("crop_kernel");
    if (hr_out) {   // (NULL: the window is the whole image and the caller aliases hr instead of copying it)
        hipLaunchKernelGGL(crop_kernel, dim3((unsigned)((th + 255) / 256)), dim3(256), 0, st, hr, hr_out, n * c, hr_h, hr_w, hr_size,
                           hr_top, hr_left, 0);
        RESR_CHECK_LAUNCH("crop_kernel");
    }
    return RESR_OK;
}

}  // namespace resr
