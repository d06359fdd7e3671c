// wgrad.hip -- weight gradient of the 3x3 convolution on the gfx950 matrix cores.
//
//   dW[co][ci][tap] = sum_p G[p][co] * X[p + tap][ci]          db[co] = sum_p G[p][co]
// (autograd backward of every F.conv2d call site of the reference model.py wrt weight / bias).
//
// GEMM view: M = cout (A = G^T), N = 32 input channels (B = X shifted by the tap), K = pixels.
// Both operands live pixel-major in HBM and in LDS, so the contraction index is the *row* index of
// the LDS tiles:
//   f16: ds_read_b64_tr_b16 (gfx950 transpose read) delivers, per lane, 4 pixels of one channel --
//        exactly an MFMA fragment; two reads feed one v_mfma_f32_32x32x16_f16.
//   f32: v_mfma_f32_32x32x2_f32 takes one element per lane, so a plain ds_read_b32 (32 consecutive
//        channels of one pixel per half-wave) is already fragment-shaped.
// A workgroup owns one 32-channel chunk of X and all of cout; its 4*MT waves split a
// (4*RPW) x 32 pixel tile by rows (K-split) and by cout tile; every wave keeps all 9 taps' 32x32
// accumulators (144 VGPRs) live while the workgroup walks its share of the pixel tiles, then
// writes its own fp32 slab.  A second launch reduces the slabs deterministically into OIHW fp32.
#include "common.h"

namespace resr {

struct WgradArgs {
    const char* x0;
    const char* x1;
    const char* g;
    float* partial;
    int n, h, w_, hs, ws;
    int cin, cin0;
    int x0_stride_b, x1_stride_b, g_stride_b;
    int cout_pad, flags, splits;
    int tiles_x, tiles_y, ntiles;
};

typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 fp16x4_t;

__device__ __forceinline__ uint2 tr_read(const char* lds_addr) {
    auto p = reinterpret_cast<__attribute__((address_space(3))) fp16x4_t*>(
        (__attribute__((address_space(3))) char*)lds_addr);
    const fp16x4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4f16(p);
    return __builtin_bit_cast(uint2, r);
}

template <typename T, int MT, int RPW>
__global__ __launch_bounds__(256 * MT) void wgrad_kernel(const WgradArgs a) {
    constexpr int NTHR = 256 * MT;
    constexpr int E = 16 / (int)sizeof(T);
    constexpr int SPP = 32 / E;
    constexpr int PB = 32 * (int)sizeof(T);
    constexpr int TH = 4 * RPW, HH = TH + 2, HW = 34;
    constexpr int XSLOT = HH * HW * SPP, GSLOT = MT * TH * 32 * SPP;
    constexpr int NSX = (XSLOT + NTHR - 1) / NTHR, NSG = GSLOT / NTHR;
    constexpr int XBUF = HH * HW * PB, GBUF = MT * TH * 32 * PB, BUF = XBUF + GBUF;
    static_assert(GSLOT % NTHR == 0, "G tile must split evenly");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ck = blockIdx.x, split = blockIdx.y;
    const bool up = (a.flags & RESR_CONV_UPSAMPLE_IN) != 0;
    const int c0 = ck * 32;
    const bool seg1 = c0 >= a.cin0;
    const char* xbase = seg1 ? a.x1 : a.x0;
    const unsigned xstride = seg1 ? a.x1_stride_b : a.x0_stride_b;
    const unsigned xch = (unsigned)(seg1 ? c0 - a.cin0 : c0) * (unsigned)sizeof(T);
    const size_t src_px = (size_t)a.hs * a.ws;

    uint4 sx[NSX], sg[NSG];
    auto stage_load = [&](int tile) {
        const int tx = tile % a.tiles_x;
        const int t2 = tile / a.tiles_x;
        const int ty = t2 % a.tiles_y;
        const int n = t2 / a.tiles_y;
        const int x0 = tx * 32, y0 = ty * TH;
        const char* xb = xbase + (size_t)n * src_px * xstride + xch;
#pragma unroll
        for (int i = 0; i < NSX; ++i) {
            const int s = tid + i * NTHR;
            const int c16 = s % SPP, hp = s / SPP;
            const int hy = hp / HW, hx = hp - hy * HW;
            const int iy = y0 + hy - 1, ix = x0 + hx - 1;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (s < XSLOT && iy >= 0 && iy < a.h && ix >= 0 && ix < a.w_) {
                const int sy = up ? (iy >> 1) : iy, sxx = up ? (ix >> 1) : ix;
                v = *reinterpret_cast<const uint4*>(xb + (size_t)(sy * a.ws + sxx) * xstride + (c16 << 4));
            }
            sx[i] = v;
        }
        const char* gb = a.g + (size_t)n * a.h * a.w_ * a.g_stride_b;
#pragma unroll
        for (int i = 0; i < NSG; ++i) {
            const int s = tid + i * NTHR;
            const int c16 = s % SPP;
            int r = s / SPP;
            const int px = r % 32; r /= 32;
            const int row = r % TH;
            const int m = r / TH;
            const int iy = y0 + row, ix = x0 + px;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (iy < a.h && ix < a.w_)
                v = *reinterpret_cast<const uint4*>(gb + (size_t)(iy * a.w_ + ix) * a.g_stride_b + m * PB + (c16 << 4));
            sg[i] = v;
        }
    };
    auto stage_store = [&](int buf) {
        char* xb = smem + buf * BUF;
        char* gb = xb + XBUF;
#pragma unroll
        for (int i = 0; i < NSX; ++i) {
            const int s = tid + i * NTHR;
            if (s < XSLOT) *reinterpret_cast<uint4*>(xb + (s << 4)) = sx[i];
        }
#pragma unroll
        for (int i = 0; i < NSG; ++i) {
            const int s = tid + i * NTHR;
            *reinterpret_cast<uint4*>(gb + (s << 4)) = sg[i];
        }
    };

    float16v acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int rw = wave & 3, mt = wave >> 2;
    const int kh = lane >> 5;

    int tile = split;
    if (tile < a.ntiles) {
        stage_load(tile);
        stage_store(0);
    }
    __syncthreads();
    int it = 0;
    for (; tile < a.ntiles; tile += a.splits, ++it) {
        const int next = tile + a.splits;
        const bool more = next < a.ntiles;
        if (more) stage_load(next);
        const char* xb = smem + (it & 1) * BUF;
        const char* gb = xb + XBUF + mt * (TH * 32 * PB);
        if constexpr (sizeof(T) == 2) {
            // lane -> (pixel sub-row a>>2, 4-channel group a&3) inside its 16-lane group; groups 0/1 of a
            // half-wave take channels 0-15 / 16-31, half-waves take pixels +0..7 / +8..15 of the k-step
            const int a16 = lane & 15;
            const int chb = ((((lane >> 4) & 1) << 4) + ((a16 & 3) << 2)) * 2;
            const int pxl = (kh << 3) + (a16 >> 2);
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr) {
                const int row = rw * RPW + rr;
#pragma unroll
                for (int kb = 0; kb < 32; kb += 16) {
                    const char* ga = gb + ((row * 32 + kb + pxl) * PB) + chb;
                    uint4 af;
                    {
                        const uint2 lo = tr_read(ga), hi = tr_read(ga + 4 * PB);
                        af = make_uint4(lo.x, lo.y, hi.x, hi.y);
                    }
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
                        const int dy = tap / 3, dx = tap % 3;
                        const char* xa = xb + (((row + dy) * HW + kb + pxl + dx) * PB) + chb;
                        const uint2 lo = tr_read(xa), hi = tr_read(xa + 4 * PB);
                        const uint4 bf = make_uint4(lo.x, lo.y, hi.x, hi.y);
                        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, af),
                                                                          __builtin_bit_cast(half8, bf), acc[tap], 0, 0, 0);
                    }
                }
            }
        } else {
            const int ch = (lane & 31) * 4;
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr) {
                const int row = rw * RPW + rr;
#pragma unroll 4
                for (int kb = 0; kb < 32; kb += 2) {
                    const float av = *reinterpret_cast<const float*>(gb + ((row * 32 + kb + kh) * PB) + ch);
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
                        const int dy = tap / 3, dx = tap % 3;
                        const float bv = *reinterpret_cast<const float*>(xb + (((row + dy) * HW + kb + kh + dx) * PB) + ch);
                        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tap], 0, 0, 0);
                    }
                }
            }
        }
        if (more) stage_store((it + 1) & 1);
        __syncthreads();
    }

    // slab write: partial[((slab*9 + tap)*cout_pad + co)*cin + ci]
    const int slab = split * 4 + rw;
    const int ci = c0 + (lane & 31);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            a.partial[(((size_t)slab * 9 + tap) * a.cout_pad + co) * a.cin + ci] = acc[tap][r];
        }
    }
}

// deterministic slab reduction -> OIHW fp32 gradient
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                           int slabs, int cout_pad, int cin, int cout, int cin_real,
                                                           float scale) {
    const int per = 9 * cout_pad * cin;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= per) return;
    const int ci = e % cin;
    const int co = (e / cin) % cout_pad;
    const int tap = e / (cin * cout_pad);
    if (co >= cout || ci >= cin_real) return;
    float s = 0.f;
    for (int k = 0; k < slabs; ++k) s += partial[(size_t)k * per + e];
    dw[((size_t)co * cin_real + ci) * 9 + tap] = s * scale;
}

// bias gradient: column sums of G.  Stage 1: per-block partial sums; stage 2: reduce over blocks.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ g, float* __restrict__ part, long npix,
                                                     int stride, int cpad) {
    constexpr int E = 16 / (int)sizeof(T);
    __shared__ float red[256 * 8];
    const int groups = cpad / E;          // 4..16
    const int lanes = 256 / groups;       // pixel lanes per block
    const int cg = threadIdx.x % groups, pl = threadIdx.x / groups;
    float acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
    for (long p = (long)blockIdx.x * lanes + pl; p < npix; p += (long)gridDim.x * lanes) {
        const uint4 raw = *reinterpret_cast<const uint4*>(g + p * stride + cg * E);
        const T* v = reinterpret_cast<const T*>(&raw);
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] += (float)v[e];
    }
#pragma unroll
    for (int e = 0; e < E; ++e) red[threadIdx.x * E + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < cpad) {
        const int c = threadIdx.x, g0 = c / E, e0 = c % E;
        float s = 0.f;
        for (int l = 0; l < lanes; ++l) s += red[(l * groups + g0) * E + e0];
        part[(size_t)blockIdx.x * cpad + c] = s;
    }
}

__global__ void colsum_reduce_kernel(const float* __restrict__ part, float* __restrict__ db, int nblocks, int cpad,
                                     int cout, float scale) {
    const int c = threadIdx.x;
    if (c >= cout) return;
    float s = 0.f;
    for (int b = 0; b < nblocks; ++b) s += part[(size_t)b * cpad + c];
    db[c] = s * scale;
}

static const int kColsumBlocks = 512;

template <typename T, int MT, int RPW>
static int launch_wgrad(WgradArgs a, hipStream_t stream) {
    constexpr int PB = 32 * (int)sizeof(T);
    constexpr int TH = 4 * RPW;
    constexpr int BUF = (TH + 2) * 34 * PB + MT * TH * 32 * PB;
    a.tiles_x = (a.w_ + 31) / 32;
    a.tiles_y = (a.h + TH - 1) / TH;
    a.ntiles = a.tiles_x * a.tiles_y * a.n;
    const size_t lds = 2 * BUF;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<T, MT, RPW>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL((wgrad_kernel<T, MT, RPW>), dim3(a.cin / 32, a.splits), dim3(256 * MT), lds, stream, a);
    RESR_CHECK_LAUNCH("wgrad_kernel");
    return RESR_OK;
}

size_t wgrad_partial_bytes(const ResrWgradDesc* d) {
    const size_t slabs = (size_t)d->splits * 4;
    return (slabs * 9 * d->cout_pad * d->cin + (size_t)kColsumBlocks * d->cout_pad) * sizeof(float);
}

int wgrad_dispatch(const ResrWgradDesc* d, const void* x0, const void* x1, const void* g, float* partial, float* dw,
                   float* db, hipStream_t stream) {
    if (!d || !x0 || !g || !partial || !dw) return fail(RESR_ERR_ARG, "wgrad: null argument");
    if (d->cin <= 0 || (d->cin & 31) || (d->cin0 & 31) || d->cin0 <= 0 || d->cin0 > d->cin)
        return fail(RESR_ERR_ARG, "wgrad: cin=%d cin0=%d", d->cin, d->cin0);
    if (d->cin0 < d->cin && !x1) return fail(RESR_ERR_ARG, "wgrad: x1 missing");
    if (d->cout_pad != 32 && d->cout_pad != 64) return fail(RESR_ERR_ARG, "wgrad: cout_pad=%d", d->cout_pad);
    if (d->cout <= 0 || d->cout > d->cout_pad || d->cin_real <= 0 || d->cin_real > d->cin)
        return fail(RESR_ERR_ARG, "wgrad: cout=%d cin_real=%d", d->cout, d->cin_real);
    if (d->splits <= 0 || d->splits > 65535) return fail(RESR_ERR_ARG, "wgrad: splits=%d", d->splits);
    const size_t es = elem_size(d->dtype);
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.x0 = (const char*)x0; a.x1 = (const char*)x1; a.g = (const char*)g; a.partial = partial;
    a.n = d->n; a.h = d->h; a.w_ = d->w;
    const bool up = d->flags & RESR_CONV_UPSAMPLE_IN;
    if (up && ((d->h | d->w) & 1)) return fail(RESR_ERR_ARG, "wgrad: upsampled input needs even h,w");
    a.hs = up ? d->h / 2 : d->h; a.ws = up ? d->w / 2 : d->w;
    a.cin = d->cin; a.cin0 = d->cin0;
    a.x0_stride_b = (int)(d->in0_stride * es); a.x1_stride_b = (int)(d->in1_stride * es);
    a.g_stride_b = (int)(d->g_stride * es);
    a.cout_pad = d->cout_pad; a.flags = d->flags; a.splits = d->splits;
    int rc;
    const int mt = d->cout_pad / 32;
    if (d->dtype == RESR_F16) rc = mt == 1 ? launch_wgrad<half_t, 1, 2>(a, stream) : launch_wgrad<half_t, 2, 2>(a, stream);
    else if (d->dtype == RESR_F32) rc = mt == 1 ? launch_wgrad<float, 1, 1>(a, stream) : launch_wgrad<float, 2, 1>(a, stream);
    else return fail(RESR_ERR_ARG, "wgrad: dtype=%d", d->dtype);
    if (rc) return rc;
    const int slabs = d->splits * 4;
    const int per = 9 * d->cout_pad * d->cin;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((per + 255) / 256), dim3(256), 0, stream, partial, dw, slabs,
                       d->cout_pad, d->cin, d->cout, d->cin_real, d->scale);
    RESR_CHECK_LAUNCH("wgrad_reduce_kernel");
    if (db) {
        float* part = partial + (size_t)slabs * per;
        const long npix = (long)d->n * d->h * d->w;
        if (d->dtype == RESR_F16)
            hipLaunchKernelGGL(colsum_kernel<half_t>, dim3(kColsumBlocks), dim3(256), 0, stream, (const half_t*)g, part, npix, d->g_stride, d->cout_pad);
        else
            hipLaunchKernelGGL(colsum_kernel<float>, dim3(kColsumBlocks), dim3(256), 0, stream, (const float*)g, part, npix, d->g_stride, d->cout_pad);
        RESR_CHECK_LAUNCH("colsum_kernel");
        hipLaunchKernelGGL(colsum_reduce_kernel, dim3(1), dim3(64), 0, stream, part, db, kColsumBlocks, d->cout_pad, d->cout, d->scale);
        RESR_CHECK_LAUNCH("colsum_reduce_kernel");
    }
    return RESR_OK;
}

}  // namespace resr
