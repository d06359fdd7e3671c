// layout.hip -- HBM-bound layout / elementwise kernels around the generator.
//   nchw_to_nhwc : module surface [N,C,H,W] fp32  ->  pixel-major, channel-padded T, with the
//                  PixelUnshuffle of model.py:220,257 folded in (x2 / x1 generators) and an optional
//                  pass-mask (backward of clamp_, model.py:270)
//   nhwc_to_nchw : the inverse (gradient wrt the input image), PixelShuffle folded in
//   sumpool2x2   : backward of F.interpolate(scale_factor=2, mode="nearest") (model.py:264-265)
//                  fused with the LeakyReLU backward of the producer
//   add_inplace  : skip-connection gradient merge (model.py:262)
//   absmax       : bits of max |g_y| of a backward pass's incoming gradient (common.h: grad_prescale)
// All are one-pass, 16-byte vectorised where the layout allows; each thread owns one pixel.
#include <math.h>

#include "common.h"

namespace resr {

// RESR_F16X2 (lo_off != 0 with T = f16): every tensor of T is a (hi, lo) pair, lo at element offset lo_off; values are
// split / recombined in fp32 (see include/resr.h).
__device__ __forceinline__ void split_f16(float v, half_t& hi, half_t& lo) {
    hi = (half_t)v;
    lo = (half_t)((v - (float)hi) * kLoScale);
}

template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst,
                                                           int n, int c, int h, int w, int r, int c_pad,
                                                           const uint8_t* __restrict__ mask, long lo_off,
                                                           const unsigned* __restrict__ amax, long q_off) {
    // output pixel grid is (h/r) x (w/r); output channel = ch*r*r + i*r + j  (torch pixel_unshuffle).
    // One thread per 16-byte piece of an output pixel: consecutive threads write consecutive 16 bytes.
    // Rows and images come from blockIdx.y / blockIdx.z, (pixel, piece) inside the row from one 32-bit division: decoded from one flat
    // 64-bit index this was four emulated 64-bit divisions per 16 bytes (1 TB/s on the discriminator's 16 x 3 x 256^2 input).
    constexpr int E = 16 / (int)sizeof(T);
    const int ho = h / r, wo = w / r;
    const unsigned pieces = (unsigned)c_pad / E;
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= (unsigned)wo * pieces) return;
    const int xo = (int)(idx / pieces);
    const int piece = (int)(idx - (unsigned)xo * pieces);
    const int creal = c * r * r;
    const float pre = amax ? grad_prescale(*amax, false) : 1.f;   // (exact: a power of two)
    // rows / images from the grid, strided: a grid dimension holds 65535 at most (taller images, larger batches loop)
    for (int b = (int)blockIdx.z; b < n; b += (int)gridDim.z)
    for (int yo = (int)blockIdx.y; yo < ho; yo += (int)gridDim.y) {
        const long p = ((long)b * ho + yo) * wo + xo;
        uint4 out, outl;
        T* o = reinterpret_cast<T*>(&out);
        T* ol = reinterpret_cast<T*>(&outl);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int co = piece * E + e;
            float v = 0.f;
            if (co < creal) {
                const int ch = co / (r * r), ij = co % (r * r), i = ij / r, j = ij % r;
                const size_t q = (((size_t)b * c + ch) * h + (yo * r + i)) * w + (xo * r + j);
                v = src[q] * pre;
                if (mask) v = mask[q] ? v : 0.f;
            }
            if constexpr (sizeof(T) == 2) {
                if (lo_off) split_f16(v, o[e], ol[e]);
                else o[e] = (T)v;
            } else {
                o[e] = (T)v;
            }
        }
        *reinterpret_cast<uint4*>(dst + p * c_pad + piece * E) = out;
        if (sizeof(T) == 2 && lo_off) *reinterpret_cast<uint4*>(dst + lo_off + p * c_pad + piece * E) = outl;
        if constexpr (sizeof(T) == 2) {
            if (lo_off && q_off) {
                // the q tensor of RESR_CONV_MX_PAIRS (include/resr.h): per pixel and 32-channel chunk 64 bytes, byte c = bf8(hi[c]), byte
                // 32 + c = bf8(lo[c]) (e5m2 = the f16 pattern rounded to its upper byte, nearest even); this thread owns 8 channels
                const unsigned short* hs = reinterpret_cast<const unsigned short*>(&out);
                const unsigned short* ls = reinterpret_cast<const unsigned short*>(&outl);
                unsigned long long qh = 0ull, ql = 0ull;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned a = hs[e], b2 = ls[e];
                    qh |= (unsigned long long)(((a + 0x7fu + ((a >> 8) & 1u)) >> 8) & 0xffu) << (8 * e);
                    ql |= (unsigned long long)(((b2 + 0x7fu + ((b2 >> 8) & 1u)) >> 8) & 0xffu) << (8 * e);
                }
                char* rec = reinterpret_cast<char*>(dst + q_off + p * c_pad + (piece >> 2) * 32) + (piece & 3) * 8;
                *reinterpret_cast<unsigned long long*>(rec) = qh;
                *reinterpret_cast<unsigned long long*>(rec + 32) = ql;
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst,
                                                           int n, int c, int h, int w, int r, int src_stride, long lo_off, int wsh,
                                                           const unsigned* __restrict__ amax) {
    // src pixel grid (h/r) x (w/r) with c*r*r channels; dst [n,c,h,w]
    // x from the thread, row from blockIdx.y, (image, channel) from blockIdx.z: no 64-bit divisions per element.  A workgroup
    // covers 2^wsh columns x 256 >> wsh rows (narrow images -- the discriminator's coarse levels -- would leave most of a
    // 256-column workgroup idle); rows and (image, channel) pairs beyond the grid's 65535 loop.
    const int x = (int)(blockIdx.x << wsh) + (int)(threadIdx.x & ((1u << wsh) - 1u));
    if (x >= w) return;
    const int rows_pb = 256 >> wsh, ysub = (int)(threadIdx.x >> wsh);
    const int ho = h / r, wo = w / r;
    const float post = amax ? grad_prescale(*amax, true) : 1.f;
    for (unsigned z = blockIdx.z; z < (unsigned)n * (unsigned)c; z += gridDim.z) {
        const int b = (int)(z / (unsigned)c), ch = (int)(z - (unsigned)b * (unsigned)c);
        for (int y = (int)blockIdx.y * rows_pb + ysub; y < h; y += (int)gridDim.y * rows_pb) {
            const size_t q = (((size_t)b * c + ch) * h + y) * w + x;
            const int co = ch * r * r + (y % r) * r + (x % r);
            const size_t p = ((size_t)b * ho + y / r) * wo + x / r;
            float v = (float)src[p * src_stride + co];
            if (sizeof(T) == 2 && lo_off) v = __builtin_fmaf((float)src[lo_off + p * src_stride + co], kLoInv, v);
            dst[q] = v * post;
        }
    }
}

// bits of max |src| * 2^-t (positive floats order like their bit patterns; a NaN beats every number, so a non-finite gradient stays
// visible to grad_prescale).  *slot is zeroed by the dispatcher; the maximum is order-independent, hence deterministic.
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ src, long count, unsigned* __restrict__ slot, int vec, float down) {
    unsigned m = 0u;
    const long stride = (long)gridDim.x * 256 * 4;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < count; i += stride) {
        if (vec && i + 4 <= count) {
            const uint4 v = *reinterpret_cast<const uint4*>(src + i);
            const unsigned a = v.x & 0x7fffffffu, b = v.y & 0x7fffffffu, c = v.z & 0x7fffffffu, d = v.w & 0x7fffffffu;
            const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
            const unsigned q = ab > cd ? ab : cd;
            m = m > q ? m : q;
        } else {
            for (long k = i; k < count && k < i + 4; ++k) {
                const unsigned a = __float_as_uint(src[k]) & 0x7fffffffu;
                m = m > a ? m : a;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned other = (unsigned)__shfl_xor((int)m, o);
        m = m > other ? m : other;
    }
    __shared__ unsigned wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned a = wm[0] > wm[1] ? wm[0] : wm[1], b = wm[2] > wm[3] ? wm[2] : wm[3];
        // times 2^-t (exact; an inf or NaN stays one): the slot holds the bits of max |src| * 2^-t
        atomicMax(slot, __float_as_uint(__uint_as_float(a > b ? a : b) * down) & 0x7fffffffu);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void sumpool2x2_kernel(const T* __restrict__ src, T* __restrict__ dst,
                                                         const T* __restrict__ mask, int n, int ho, int wo, int c,
                                                         float slope, long src_lo, long dst_lo) {
    constexpr int E = 16 / (int)sizeof(T);
    const unsigned groups = (unsigned)c / E;
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;      // (x, group) inside the output row; row / image from the grid
    if (idx >= (unsigned)wo * groups) return;
    const int x = (int)(idx / groups), g = (int)(idx - (unsigned)x * groups);
    const int wi = wo * 2;
    for (int b = (int)blockIdx.z; b < n; b += (int)gridDim.z)          // (grid dimensions hold 65535 at most: the rest loops)
    for (int y = (int)blockIdx.y; y < ho; y += (int)gridDim.y) {
    const long p = ((long)b * ho + y) * wo + x;
    const T* s = src + (((size_t)b * ho * 2 + y * 2) * wi + x * 2) * c + g * E;
    float acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const uint4 raw = *reinterpret_cast<const uint4*>(s + ((size_t)dy * wi + dx) * c);
            const T* v = reinterpret_cast<const T*>(&raw);
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] += (float)v[e];
            if (sizeof(T) == 2 && src_lo) {
                const uint4 rawl = *reinterpret_cast<const uint4*>(s + src_lo + ((size_t)dy * wi + dx) * c);
                const T* vl = reinterpret_cast<const T*>(&rawl);
#pragma unroll
                for (int e = 0; e < E; ++e) acc[e] = __builtin_fmaf((float)vl[e], kLoInv, acc[e]);
            }
        }
    if (mask) {   // a saved activation of dst's shape (RESR_F16X2: a pair with dst's hi -> lo offset; common.h pair_positive)
        const uint4 raw = *reinterpret_cast<const uint4*>(mask + p * c + g * E);
        const T* v = reinterpret_cast<const T*>(&raw);
        bool pos[E];
        bool anyz = false;
#pragma unroll
        for (int e = 0; e < E; ++e) { pos[e] = (float)v[e] > 0.f; anyz = anyz || (float)v[e] == 0.f; }
        if constexpr (sizeof(T) == 2) {
            if (dst_lo && anyz) {
                const uint4 rawl = *reinterpret_cast<const uint4*>(mask + dst_lo + p * c + g * E);
                const T* vl = reinterpret_cast<const T*>(&rawl);
#pragma unroll
                for (int e = 0; e < E; ++e) pos[e] = pair_positive(v[e], vl[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] *= pos[e] ? 1.f : slope;
    }
    uint4 outv, outl;
    T* o = reinterpret_cast<T*>(&outv);
    T* ol = reinterpret_cast<T*>(&outl);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if constexpr (sizeof(T) == 2) {
            if (dst_lo) split_f16(acc[e], o[e], ol[e]);
            else o[e] = (T)acc[e];
        } else {
            o[e] = (T)acc[e];
        }
    }
    *reinterpret_cast<uint4*>(dst + p * c + g * E) = outv;
    if (sizeof(T) == 2 && dst_lo) *reinterpret_cast<uint4*>(dst + dst_lo + p * c + g * E) = outl;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void add_inplace_kernel(T* __restrict__ dst, const T* __restrict__ src, long count,
                                                          long dst_lo, long src_lo) {
    constexpr int E = 16 / (int)sizeof(T);
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * E;
    if (i >= count) return;
    uint4 a = *reinterpret_cast<const uint4*>(dst + i);
    const uint4 b = *reinterpret_cast<const uint4*>(src + i);
    T* pa = reinterpret_cast<T*>(&a);
    const T* pb = reinterpret_cast<const T*>(&b);
    if constexpr (sizeof(T) == 2) {
        if (dst_lo) {   // hi/lo pairs: add in fp32, split again
            uint4 al = *reinterpret_cast<const uint4*>(dst + dst_lo + i);
            const uint4 bl = *reinterpret_cast<const uint4*>(src + src_lo + i);
            T* pal = reinterpret_cast<T*>(&al);
            const T* pbl = reinterpret_cast<const T*>(&bl);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const float v = ((float)pa[e] + (float)pb[e]) + ((float)pal[e] + (float)pbl[e]) * kLoInv;
                split_f16(v, pa[e], pal[e]);
            }
            *reinterpret_cast<uint4*>(dst + i) = a;
            *reinterpret_cast<uint4*>(dst + dst_lo + i) = al;
            return;
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) pa[e] = (T)((float)pa[e] + (float)pb[e]);
    *reinterpret_cast<uint4*>(dst + i) = a;
}

static unsigned blocks_for(long total) { return (unsigned)((total + 255) / 256); }

// RESR_F16X2: lo_off < 0 selects the C-ABI default -- the lo tensor directly follows the hi tensor
// The sticky half of the slot (common.h): a non-finite result of the previous lifted pass lowers the target; m leaves times 2^back-off.
__global__ void absmax_backoff_kernel(unsigned* __restrict__ slot) {
    unsigned b = slot[1];
    if (b > (unsigned)kPrescaleBackoffMax) b = (unsigned)kPrescaleBackoffMax;
    if (slot[2]) {
        b = b + kPrescaleBackoffStep > (unsigned)kPrescaleBackoffMax ? (unsigned)kPrescaleBackoffMax : b + kPrescaleBackoffStep;
        slot[2] = 0u;
    }
    slot[1] = b;
    const unsigned m = slot[0], e = (m >> 23) & 0xffu;
    if (b && e != 0u && e < 255u) {
        const unsigned e2 = e + b > 254u ? 254u : e + b;
        slot[0] = (m & 0x807fffffu) | (e2 << 23);
    }
}

int absmax_dispatch(const float* src, long count, unsigned* slot, int target_log2, hipStream_t stream) {
    if (!src || !slot || count <= 0 || target_log2 < -64 || target_log2 > 64) return fail(RESR_ERR_ARG, "absmax: bad argument");
    if (hipMemsetAsync(slot, 0, sizeof(unsigned), stream) != hipSuccess) return fail(RESR_ERR_LAUNCH, "absmax: memset");
    const long want = (count / 4 + 255) / 256;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)(want < 1 ? 1 : want > 2048 ? 2048 : want)), dim3(256), 0, stream, src, count, slot,
                       ((size_t)src & 15) ? 0 : 1, ldexpf(1.f, -target_log2));
    RESR_CHECK_LAUNCH("absmax_kernel");
    hipLaunchKernelGGL(absmax_backoff_kernel, dim3(1), dim3(1), 0, stream, slot);
    RESR_CHECK_LAUNCH("absmax_backoff_kernel");
    return RESR_OK;
}

int nchw_to_nhwc_q_dispatch(const float* src, void* dst, int n, int c, int h, int w, int r, int c_pad, int dtype,
                            const uint8_t* mask, hipStream_t stream, long lo_off, const unsigned* amax, long q_off);
int nchw_to_nhwc_scaled_dispatch(const float* src, void* dst, int n, int c, int h, int w, int r, int c_pad, int dtype,
                                 const uint8_t* mask, hipStream_t stream, long lo_off, const unsigned* amax);
int nchw_to_nhwc_dispatch(const float* src, void* dst, int n, int c, int h, int w, int r, int c_pad, int dtype,
                          const uint8_t* mask, hipStream_t stream, long lo_off) {
    return nchw_to_nhwc_scaled_dispatch(src, dst, n, c, h, w, r, c_pad, dtype, mask, stream, lo_off, nullptr);
}

// amax: device pointer to the bits of max |src| (absmax_dispatch) -- src is read times grad_prescale(*amax) -- or nullptr
int nchw_to_nhwc_scaled_dispatch(const float* src, void* dst, int n, int c, int h, int w, int r, int c_pad, int dtype,
                                 const uint8_t* mask, hipStream_t stream, long lo_off, const unsigned* amax) {
    return nchw_to_nhwc_q_dispatch(src, dst, n, c, h, w, r, c_pad, dtype, mask, stream, lo_off, amax, 0L);
}

// q_off != 0 (RESR_F16X2): also write the q tensor (bf8 of hi and lo, RESR_CONV_MX_PAIRS) at that element offset behind dst
int nchw_to_nhwc_q_dispatch(const float* src, void* dst, int n, int c, int h, int w, int r, int c_pad, int dtype,
                            const uint8_t* mask, hipStream_t stream, long lo_off, const unsigned* amax, long q_off) {
    if (q_off != 0 && (dtype != RESR_F16X2 || (c_pad & 31))) return fail(RESR_ERR_ARG, "nchw_to_nhwc: a q tensor goes with RESR_F16X2 and c_pad %% 32 == 0");
    if (!src || !dst || n <= 0 || c <= 0 || h <= 0 || w <= 0 || r <= 0 || (h % r) || (w % r) || c * r * r > c_pad || (c_pad & 7))
        return fail(RESR_ERR_ARG, "nchw_to_nhwc: bad argument (c=%d r=%d c_pad=%d h=%d w=%d)", c, r, c_pad, h, w);
    const dim3 grid(blocks_for((long)(w / r) * (c_pad / (dtype != RESR_F32 ? 8 : 4))), (unsigned)(h / r > 65535 ? 65535 : h / r), (unsigned)(n > 65535 ? 65535 : n));
    if (dtype == RESR_F16X2 && lo_off < 0) lo_off = (long)n * (h / r) * (w / r) * c_pad;
    if (dtype != RESR_F16X2) lo_off = 0;
    if (dtype != RESR_F32)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<half_t>, grid, dim3(256), 0, stream, src, (half_t*)dst, n, c, h, w, r, c_pad, mask, lo_off, amax, q_off);
    else
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid, dim3(256), 0, stream, src, (float*)dst, n, c, h, w, r, c_pad, mask, 0L, amax, 0L);
    RESR_CHECK_LAUNCH("nchw_to_nhwc_kernel");
    return RESR_OK;
}

int nhwc_to_nchw_scaled_dispatch(const void* src, float* dst, int n, int c, int h, int w, int r, int src_stride, int dtype,
                                 hipStream_t stream, long lo_off, const unsigned* amax);
int nhwc_to_nchw_dispatch(const void* src, float* dst, int n, int c, int h, int w, int r, int src_stride, int dtype,
                          hipStream_t stream, long lo_off) {
    return nhwc_to_nchw_scaled_dispatch(src, dst, n, c, h, w, r, src_stride, dtype, stream, lo_off, nullptr);
}

// amax: the result leaves times 1 / grad_prescale(*amax) (the inverse of what nchw_to_nhwc_scaled_dispatch applied), or nullptr
int nhwc_to_nchw_scaled_dispatch(const void* src, float* dst, int n, int c, int h, int w, int r, int src_stride, int dtype,
                                 hipStream_t stream, long lo_off, const unsigned* amax) {
    if (!src || !dst || n <= 0 || c <= 0 || h <= 0 || w <= 0 || r <= 0 || (h % r) || (w % r))
        return fail(RESR_ERR_ARG, "nhwc_to_nchw: bad argument");
    if ((long)n * c > 0x7fffffffL) return fail(RESR_ERR_ARG, "nhwc_to_nchw: n * c beyond 2^31");
    int wsh = 8;                                   // columns per workgroup = 2^wsh >= w (at least a wavefront's half), rows = 256 >> wsh
    while (wsh > 5 && (1 << (wsh - 1)) >= w) --wsh;
    const long rowblocks = ((long)h + (256 >> wsh) - 1) / (256 >> wsh);
    const dim3 grid((unsigned)((w + (1 << wsh) - 1) >> wsh), (unsigned)(rowblocks > 65535 ? 65535 : rowblocks), (unsigned)((long)n * c > 65535 ? 65535 : n * c));
    if (dtype == RESR_F16X2 && lo_off < 0) lo_off = (long)n * (h / r) * (w / r) * src_stride;
    if (dtype != RESR_F16X2) lo_off = 0;
    if (dtype != RESR_F32)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<half_t>, grid, dim3(256), 0, stream, (const half_t*)src, dst, n, c, h, w, r, src_stride, lo_off, wsh, amax);
    else
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, grid, dim3(256), 0, stream, (const float*)src, dst, n, c, h, w, r, src_stride, 0L, wsh, amax);
    RESR_CHECK_LAUNCH("nhwc_to_nchw_kernel");
    return RESR_OK;
}

int sumpool2x2_dispatch(const void* src, void* dst, const void* mask, int n, int ho, int wo, int c, int dtype,
                        float slope, hipStream_t stream, long src_lo, long dst_lo) {
    const int E = dtype != RESR_F32 ? 8 : 4;
    if (dtype == RESR_F16X2 && src_lo < 0) src_lo = (long)n * ho * wo * 4 * c;
    if (dtype == RESR_F16X2 && dst_lo < 0) dst_lo = (long)n * ho * wo * c;
    if (dtype != RESR_F16X2) src_lo = dst_lo = 0;
    if (!src || !dst || n <= 0 || ho <= 0 || wo <= 0 || c <= 0 || (c % E))
        return fail(RESR_ERR_ARG, "sumpool2x2: bad argument");
    const dim3 grid(blocks_for((long)wo * (c / E)), (unsigned)(ho > 65535 ? 65535 : ho), (unsigned)(n > 65535 ? 65535 : n));
    if (dtype != RESR_F32)
        hipLaunchKernelGGL(sumpool2x2_kernel<half_t>, grid, dim3(256), 0, stream, (const half_t*)src, (half_t*)dst, (const half_t*)mask, n, ho, wo, c, slope, src_lo, dst_lo);
    else
        hipLaunchKernelGGL(sumpool2x2_kernel<float>, grid, dim3(256), 0, stream, (const float*)src, (float*)dst, (const float*)mask, n, ho, wo, c, slope, 0L, 0L);
    RESR_CHECK_LAUNCH("sumpool2x2_kernel");
    return RESR_OK;
}

int add_inplace_dispatch(void* dst, const void* src, long count, int dtype, hipStream_t stream, long dst_lo, long src_lo) {
    const int E = dtype != RESR_F32 ? 8 : 4;
    if (dtype == RESR_F16X2 && (dst_lo <= 0 || src_lo <= 0)) return fail(RESR_ERR_ARG, "add_inplace: RESR_F16X2 needs the lo offsets");
    if (dtype != RESR_F16X2) dst_lo = src_lo = 0;
    if (!dst || !src || count <= 0 || (count % E)) return fail(RESR_ERR_ARG, "add_inplace: bad argument");
    const long total = count / E;
    if (dtype != RESR_F32)
        hipLaunchKernelGGL(add_inplace_kernel<half_t>, dim3(blocks_for(total)), dim3(256), 0, stream, (half_t*)dst, (const half_t*)src, count, dst_lo, src_lo);
    else
        hipLaunchKernelGGL(add_inplace_kernel<float>, dim3(blocks_for(total)), dim3(256), 0, stream, (float*)dst, (const float*)src, count, 0L, 0L);
    RESR_CHECK_LAUNCH("add_inplace_kernel");
    return RESR_OK;
}

}  // namespace resr
