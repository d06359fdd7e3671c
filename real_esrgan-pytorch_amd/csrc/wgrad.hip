// wgrad.hip -- weight / bias gradients of the 3x3 convolutions on the gfx950 matrix cores.
//
//   dW[co][ci][tap] = sum_p G[p][co] * X[p + tap][ci]          db[co] = sum_p G[p][co]
// (autograd backward of every F.conv2d call site of the reference model.py wrt weight / bias).
//
// GEMM view: M = 32 output channels (A = G^T), N = 32 input channels (B = X shifted by the tap),
// K = pixels.  Both operands are pixel-major in HBM and in LDS, so the contraction index is the
// *row* index of the LDS tiles:
//   f16: ds_read_b64_tr_b16 (gfx950 transpose read) hands each lane 4 pixels of one channel --
//        exactly an MFMA fragment; two reads feed one v_mfma_f32_32x32x16_f16.
//   f32: v_mfma_f32_32x32x2_f32 takes one element per lane, so a plain ds_read_b32 (32 consecutive
//        channels of one pixel per half-wave) is already fragment-shaped.
//
// Work decomposition.  A *product* is one (conv, 32-channel chunk of X, 32-channel tile of G) pair; a
// launch takes the products of convolutions that share the pixel geometry -- all five convs of a dense
// block go in one launch (26 products), because they read the same workspace and are individually too
// small to fill 256 CUs.  Two kernels compute them:
//   wgrad_kernel<T,RPW>  "pair" kernel, one product per workgroup: grid = (products, pixel splits).  The 4
//                        waves of a workgroup split a (4*RPW) x 32 pixel tile by rows (K-split) and each keeps
//                        all 9 taps' 32x32 accumulators (144 VGPRs) while the workgroup walks its share of the
//                        pixel tiles.  Strict (f32) mode, and f16 when the quad kernel's preconditions fail.
//   wgrad_quad_kernel    f16: four products (2 X chunks x 2 G tiles) per 8-wave workgroup on one staged tile
//                        (see its own comment); the host groups a launch's products into such 2x2 jobs.
// Either way the waves of a workgroup are summed through LDS at the end and one fp32 slab [9][32][32] (+32 bias
// sums) per (product, split) is written; a second launch reduces the slabs in a fixed order (deterministic) into
// the OIHW fp32 gradient arena.
#include <stdlib.h>

#include <map>
#include <mutex>
#include <vector>

#include "common.h"
#include "wgrad.h"

namespace resr {

constexpr int kMaxJobs = kWgradMaxJobs;    // (X chunk, G tile) tap-products per launch (wgrad.h)
constexpr int kMaxReduce = 80;  // algorithmic products per launch (ReduceArgs: 80 x 48 B)
constexpr int kMaxQuads = 40;   // 2x2 jobs per launch of the quad kernel
constexpr int kX2WgradProductsDefault = 3;   // see wgrad_x2_products()
constexpr int kSlab = 9 * 1024 + 32;   // floats per (job, split): 9 taps x 32 co x 32 ci, then 32 bias sums

struct WgradJob {
    const char* x;        // X base + channel offset of the job's 32-channel chunk
    const char* g;        // G base + channel offset of the job's 32-channel tile
    unsigned xstride_b, gstride_b;
    unsigned slab_off;    // float offset of this job's [splits][kSlab] slabs in `partial`
    unsigned want_bias;   // bit 0: also produce sum_p G[p][co] (one job per co tile does); bit 1: ONLY that -- the quad kernel skips the job's tap
                          // products (its dW slabs are zeros): the (x_hi chunk 0, g_lo) job of WgradConv.g_lo_bias_only
    unsigned xsub;        // 0..3: X chunk lies in sub-position (i*2+j) of a space-to-depth image -- only 2x2 of the 9 taps of the
                          // virtual kernel of a 4x4 / stride-2 conv are non-zero there; 4: all taps
    unsigned mx;          // 1: an MX job -- x and g are q tensors (bf8 records); the job yields (x_hi, g_lo) + (x_lo, g_hi) in one slab (quad kernel only)
};

struct WgradArgs {
    WgradJob jobs[kMaxJobs];
    float* partial;
    const char* zero;     // 16 zero bytes in global memory
    int n, h, w_, hs, ws;
    int up, splits, njobs;
    int splits_mx;        // pixel splits of the MX jobs' launch (a multiple of splits)
    int fast_addr;        // 1: every operand < 4 GB and < 2^24 pixels -> 32-bit lane offsets + uniform base (see stage())
    int tiles_x, tiles_y, ntiles;
};

// RESR_F16X2: a product's slabs come in three -- (x_hi, g_hi) at slab_off, (x_hi, g_lo) at slab_b, (x_lo, g_hi) at
// slab_c, the last two carrying the lo tensors' 2^12 -- and dW = A + (B + C) * 2^-12, db = bias(A) + bias(B) * 2^-12.
struct ReduceJob {
    float* dw;            // OIHW fp32 gradient of the conv
    float* db;            // bias gradient or nullptr
    unsigned slab_off, slab_b, slab_c;   // slab_b / slab_c = ~0u outside RESR_F16X2
    short co_base, ci_base;
    int cout, cin_real;
    float scale;
    short want_bias;
    short c_bias;         // the slab_c job (an MX job) also carries the bias sum of g_lo: taken like slab_b's
};

static_assert(sizeof(WgradArgs) <= 4096, "kernel arguments");

struct ReduceArgs {
    ReduceJob jobs[kMaxReduce];
    const float* partial;
    int splits;
    int layer_nck;   // > 0: layer mode (WgradLayer) -- workgroup row b reduces product (b / nck, b % nck) of the convolution jobs[0] describes
    unsigned layer_part_stride;   // layer mode, RESR_F16X2 with three tap-products: floats from a product's (hi, hi) slabs to its (hi, lo) and on to its (lo, hi) slabs; else 0
    const unsigned* unscale;      // pre-scaled backward pass (common.h: grad_prescale): results leave times the inverse factor; else nullptr
    int splits_c;                 // slabs per slab_c region when it differs from `splits` (the MX jobs' launch has its own pixel splits); 0: = splits
};

static_assert(sizeof(ReduceArgs) <= 4096, "kernel arguments");

typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 fp16x4_t;

__device__ __forceinline__ uint2 tr_read(const char* lds_addr) {
    auto p = reinterpret_cast<__attribute__((address_space(3))) fp16x4_t*>(
        (__attribute__((address_space(3))) char*)lds_addr);
    const fp16x4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4f16(p);
    return __builtin_bit_cast(uint2, r);
}

// 16-byte LDS-DMA: the LDS destination is wave-uniform base + lane*16, the global source is per lane
__device__ __forceinline__ void glds16(const char* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ uint4 g_zero16 = {0, 0, 0, 0};

// LDS-DMA with a uniform 64-bit base (SGPR pair) + 32-bit per-lane byte offset (asm: the builtin only takes per-lane
// 64-bit pointers).  Not counted by the compiler: callers wait with s_waitcnt vmcnt(0) before the barrier.
__device__ __forceinline__ void glds16_s(const char* sbase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds_addr), "v"(voff), "s"(sbase) : "memory", "m0");
}

typedef _Float16 half2v __attribute__((ext_vector_type(2)));

// sum of the 8 f16 of a fragment in fp32: 4 x v_dot2_f32_f16 against (1,1)
__device__ __forceinline__ float sum8_f16(const uint4& v) {
    const half2v one = {(_Float16)1.0f, (_Float16)1.0f};
    float s = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, v.x), one, 0.f, false);
    s = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, v.y), one, s, false);
    s = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, v.z), one, s, false);
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, v.w), one, s, false);
}

template <typename T, int RPW>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradArgs a) {
    constexpr int E = 16 / (int)sizeof(T);
    constexpr int SPP = 32 / E;
    constexpr int PB = 32 * (int)sizeof(T);
    constexpr int TH = 4 * RPW, HH = TH + 2, HW = 34;
    constexpr int XSLOT = HH * HW * SPP, GSLOT = TH * 32 * SPP;
    constexpr int NSX = (XSLOT + 255) / 256, NSG = GSLOT / 256;
    constexpr int XSLOT_PAD = (XSLOT + 63) / 64 * 64;       // whole waves of LDS-DMA lanes
    constexpr int XBUF = XSLOT_PAD * 16, GBUF = TH * 32 * PB, BUF = XBUF + GBUF;
    static_assert(GSLOT % 256 == 0, "G tile must split evenly");
    static_assert(2 * BUF >= 4 * 16 * 64 * 4, "LDS must hold one tap of 4 waves for the final reduce");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware work map.  Blocks are dealt round-robin to the 8 XCDs (block b -> XCD b % 8, each with its own
    // L2).  All jobs of one pixel split read the same X / G tiles, so they are placed on ONE XCD, adjacent in
    // dispatch order: the k-th block of XCD x takes job k % njobs of split x + 8 * (k / njobs).
    const int bx = blockIdx.x & 7, bk = blockIdx.x >> 3;
    const int split = bx + 8 * (bk / a.njobs);
    const WgradJob job = a.jobs[bk % a.njobs];
    if (split >= a.splits) return;
    const size_t src_px = (size_t)a.hs * a.ws;

    // Staging is LDS-DMA (global_load_lds, 16 B per lane): no staging VGPRs, no ds_write pass.  The LDS image is
    // lane-linear (slot s at byte 16*s), which is exactly the pixel-major tile the transpose reads want; lanes
    // whose pixel lies outside the image fetch from a 16-byte zero page instead.
    // Tile-independent part of the lane -> slot map (the staging runs on the MFMA waves: every instruction it saves
    // is an issue slot for the matrix pipe).  X: hy<<8 | hx | (piece*16)<<16, ~0u = no slot; G: row<<8 | px | (piece*16)<<16.
    unsigned cx[NSX], cg[NSG];
#pragma unroll
    for (int i = 0; i < NSX; ++i) {
        const unsigned s = i * 256 + tid;
        const unsigned c16 = s % SPP, hp = s / SPP;
        const unsigned hy = (hp * 61681u) >> 21;  // hp / 34, exact below 100000
        cx[i] = s < (unsigned)XSLOT ? (hy << 8 | (hp - hy * HW) | (c16 << 20)) : ~0u;
    }
#pragma unroll
    for (int i = 0; i < NSG; ++i) {
        const unsigned s = i * 256 + tid;
        const unsigned c16 = s % SPP, r = s / SPP;
        cg[i] = (r >> 5) << 8 | (r & 31) | (c16 << 20);
    }
    auto stage_fast = [&](int tile, int buf) {
        const int tx = tile % a.tiles_x;
        const int t2 = tile / a.tiles_x;
        const int ty = t2 % a.tiles_y;
        const int n = t2 / a.tiles_y;
        const int x0 = tx * 32, y0 = ty * TH;
        const unsigned xl = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(smem + buf * BUF) + wave * 1024;
        const unsigned gl = xl + XBUF;
        const unsigned xn = (unsigned)n * a.hs * a.ws, gn = (unsigned)n * a.h * a.w_;
#pragma unroll
        for (int i = 0; i < NSX; ++i) {
            const unsigned c = cx[i];
            const int iy = y0 - 1 + (int)((c >> 8) & 0xff), ix = x0 - 1 + (int)(c & 0xff);
            if (c != ~0u) {
                if ((unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w_) {
                    const unsigned pix = xn + (unsigned)(iy >> a.up) * a.ws + (unsigned)(ix >> a.up);
                    glds16_s(job.x, __umul24(pix, job.xstride_b) + (c >> 16), xl + i * 4096);
                } else {
                    glds16_s(a.zero, 0u, xl + i * 4096);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NSG; ++i) {
            const unsigned c = cg[i];
            const int iy = y0 + (int)((c >> 8) & 0xff), ix = x0 + (int)(c & 0xff);
            if (iy < a.h && ix < a.w_) {
                const unsigned pix = gn + (unsigned)iy * a.w_ + (unsigned)ix;
                glds16_s(job.g, __umul24(pix, job.gstride_b) + (c >> 16), gl + i * 4096);
            } else {
                glds16_s(a.zero, 0u, gl + i * 4096);
            }
        }
    };
    auto stage_generic = [&](int tile, int buf) {
        const int tx = tile % a.tiles_x;
        const int t2 = tile / a.tiles_x;
        const int ty = t2 % a.tiles_y;
        const int n = t2 / a.tiles_y;
        const int x0 = tx * 32, y0 = ty * TH;
        const char* xg = job.x + (size_t)n * src_px * job.xstride_b;
        const char* gg = job.g + (size_t)n * a.h * a.w_ * job.gstride_b;
        char* xl = smem + buf * BUF;
        char* gl = xl + XBUF;
#pragma unroll
        for (int i = 0; i < NSX; ++i) {
            const int sbase = i * 256 + wave * 64;            // wave-uniform
            if (sbase < XSLOT_PAD) {
                const int s = sbase + lane;
                const int c16 = s % SPP, hp = s / SPP;
                const int hy = hp / HW, hx = hp - hy * HW;
                const int iy = y0 + hy - 1, ix = x0 + hx - 1;
                const bool ok = s < XSLOT && iy >= 0 && iy < a.h && ix >= 0 && ix < a.w_;
                const int sy = a.up ? (iy >> 1) : iy, sxx = a.up ? (ix >> 1) : ix;
                const char* src = ok ? xg + (size_t)(sy * a.ws + sxx) * job.xstride_b + (c16 << 4) : a.zero;
                glds16(src, xl + (sbase << 4));
            }
        }
#pragma unroll
        for (int i = 0; i < NSG; ++i) {
            const int sbase = i * 256 + wave * 64;
            const int s = sbase + lane;
            const int c16 = s % SPP;
            const int r = s / SPP;
            const int px = r % 32, row = r / 32;
            const int iy = y0 + row, ix = x0 + px;
            const bool ok = iy < a.h && ix < a.w_;
            const char* src = ok ? gg + (size_t)(iy * a.w_ + ix) * job.gstride_b + (c16 << 4) : a.zero;
            glds16(src, gl + (sbase << 4));
        }
    };

    auto stage = [&](int tile, int buf) {
        if (a.fast_addr) stage_fast(tile, buf);
        else stage_generic(tile, buf);
    };

    float16v acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;

    const int kh = lane >> 5;
    int tile = split;
    if (tile < a.ntiles) stage(tile, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the asm LDS-DMA is not counted by the compiler
    __syncthreads();
    int it = 0;
    for (; tile < a.ntiles; tile += a.splits, ++it) {
        const int next = tile + a.splits;
        if (next < a.ntiles) stage(next, (it + 1) & 1);
        const char* xb = smem + (it & 1) * BUF;
        const char* gb = xb + XBUF;
        if constexpr (sizeof(T) == 2) {
            // lane -> (pixel sub-row a16>>2, 4-channel group a16&3) inside its 16-lane group; the two groups
            // of a half-wave take channels 0-15 / 16-31, the half-waves pixels +0..7 / +8..15 of the k-step.
            // Two waves per SIMD (144 accumulator + ~100 other registers each): the partner wave's MFMAs cover
            // this wave's 20 transpose reads, so no register ping-pong is needed.
            const int a16 = lane & 15;
            const int chb = ((((lane >> 4) & 1) << 4) + ((a16 & 3) << 2)) * 2;
            const int pxl = (kh << 3) + (a16 >> 2);
            constexpr int NSTEP = RPW * 2;
            uint4 fa[1], fb[1][9];
            // one per-lane base address per operand; every (row, k-step, tap) is a compile-time immediate offset
            const char* gbase = gb + ((wave * RPW * 32 + pxl) * PB) + chb;
            const char* xbase = xb + ((wave * RPW * HW + pxl) * PB) + chb;
            auto fload = [&](int slot, int step) {
                const int r = step / 2, kb = (step & 1) * 16;
                const char* ga = gbase + (r * 32 + kb) * PB;
                const uint2 alo = tr_read(ga), ahi = tr_read(ga + 4 * PB);
                fa[slot] = make_uint4(alo.x, alo.y, ahi.x, ahi.y);
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int dy = tap / 3, dx = tap % 3;
                    const char* xa = xbase + ((r + dy) * HW + kb + dx) * PB;
                    const uint2 lo = tr_read(xa), hi = tr_read(xa + 4 * PB);
                    fb[slot][tap] = make_uint4(lo.x, lo.y, hi.x, hi.y);
                }
            };
#pragma unroll
            for (int step = 0; step < NSTEP; ++step) {
                fload(0, step);
                if (job.want_bias) bsum += sum8_f16(fa[0]);      // wave-uniform: one job per cout tile
#pragma unroll
                for (int tap = 0; tap < 9; ++tap)
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, fa[0]),
                                                                      __builtin_bit_cast(half8, fb[0][tap]), acc[tap], 0, 0, 0);
            }
        } else {
            const int ch = (lane & 31) * 4;
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr) {
                const int row = wave * RPW + rr;
#pragma unroll 4
                for (int kb = 0; kb < 32; kb += 2) {
                    const float av = *reinterpret_cast<const float*>(gb + ((row * 32 + kb + kh) * PB) + ch);
                    if (job.want_bias) bsum += av;
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
                        const int dy = tap / 3, dx = tap % 3;
                        const float bv = *reinterpret_cast<const float*>(xb + (((row + dy) * HW + kb + kh + dx) * PB) + ch);
                        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tap], 0, 0, 0);
                    }
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the next tile's asm LDS-DMA has landed
        __syncthreads();
    }

    // ---- sum the 4 row-waves through LDS, write one slab -------------------------------------------------
    float* red = reinterpret_cast<float*>(smem);               // [4 waves][16 regs][64 lanes]
    float* slab = a.partial + job.slab_off + (size_t)split * kSlab;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[tap][r];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * 256;                         // (reg, lane)
            const int r = e >> 6, l = e & 63;
            const float s = (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]);
            const int co = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), ci = l & 31;
            slab[tap * 1024 + co * 32 + ci] = s;
        }
        __syncthreads();
    }
    if (job.want_bias) {
        red[wave * 64 + lane] = bsum;
        __syncthreads();
        if (tid < 32) {
            float s = 0.f;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) s += red[wv * 64 + tid] + red[wv * 64 + 32 + tid];
            slab[9 * 1024 + tid] = s;
        }
    }
}

// The same reduction for launches of at most eight pixel splits (the discriminator's deep layers: 2..8 splits of 128..512
// products): one thread per slab element adds its <= 8 slabs in the tree the kernel above uses (bit-identical results), 256
// elements per workgroup -- the 32-element workgroups above are 148 K workgroups of mostly idle threads for a 512-product layer.
__global__ __launch_bounds__(256) void wgrad_reduce_wide_kernel(const ReduceArgs a) {
    ReduceJob job = a.jobs[a.layer_nck > 0 ? 0 : blockIdx.x];
    if (a.layer_nck > 0) {
        const int ct = blockIdx.x / a.layer_nck, ck = blockIdx.x - ct * a.layer_nck;
        job.slab_off = blockIdx.x * (unsigned)a.splits * (unsigned)kSlab;
        if (a.layer_part_stride) { job.slab_b = job.slab_off + a.layer_part_stride; job.slab_c = job.slab_b + a.layer_part_stride; }
        job.co_base = (short)(ct * 32); job.ci_base = (short)(ck * 32);
        job.want_bias = (short)((ck == 0 && job.db) ? 1 : 0);
    }
    const int e = blockIdx.y * 256 + threadIdx.x;
    if (e >= kSlab) return;
    auto tree = [&](unsigned off) {
        const float* p = a.partial + off + e;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = k < a.splits ? p[(size_t)k * kSlab] : 0.f;
        return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    };
    float s = tree(job.slab_off);
    if (job.slab_b != ~0u || job.slab_c != ~0u) {   // RESR_F16X2: dW = A + (B + C) * 2^-12, bias sums: only the g_lo product (absent for a single-f16 G)
        float s2 = job.slab_b != ~0u ? tree(job.slab_b) : 0.f;
        if (job.slab_c != ~0u && e < 9 * 1024) s2 += tree(job.slab_c);
        s = __builtin_fmaf(s2, kLoInv, s);
    }
    s *= job.scale;
    if (a.unscale) {
        const unsigned mb = *a.unscale;
        s *= grad_prescale(mb, true);
        // a LIFTED pass that overflowed: tell the next pass to aim lower (common.h: the slot's sticky words)
        if (grad_prescale_lifted(mb) && !(__builtin_fabsf(s) <= 3.0e38f)) const_cast<unsigned*>(a.unscale)[2] = 1u;
    }
    if (e < 9 * 1024) {
        const int tap = e >> 10, co = job.co_base + ((e >> 5) & 31), ci = job.ci_base + (e & 31);
        if (co < job.cout && ci < job.cin_real) job.dw[((size_t)co * job.cin_real + ci) * 9 + tap] = s;
    } else if (job.want_bias && job.db) {
        const int co = job.co_base + (e - 9 * 1024);
        if (co < job.cout) job.db[co] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------
// f16 quad kernel: one job = up to 2 X chunks x 2 G tiles (four 32x32x9 products) on one staged pixel tile.
//
// The pair kernel above stages 128 B per pixel for one product and is bound by the L2 -> LDS delivery rate of a CU
// (~25 GB/s), not by the matrix pipe.  Here the same 8 waves of a CU form ONE workgroup that shares a staged tile of
// 2 x 64 B (X) + 2 x 64 B (G) per pixel between four products: half the staged bytes per FLOP.  Wave w computes product
// p = w >> 1 (X chunk p & 1, G tile p >> 1) on rows [4 (w & 1), +4) of the 8 x 32 tile.  The LDS-DMA of the next tile
// is issued one instruction at a time between the MFMAs of the current one, so a wave that stalls on the memory
// pipe's back-pressure leaves the matrix pipe to its SIMD partner instead of stalling a whole staging phase.
// Products a job does not need (slab_off == ~0u) are computed by their two waves all the same and not written.
struct WgradQuad {
    const char* x[2];
    const char* g[2];
    unsigned xstride_b[2], gstride_b[2];
    unsigned slab_off[4];   // float offset of product p's [splits][kSlab] slabs, ~0u = product not wanted
    unsigned bias_mask;     // bit p: product p also yields sum_p G (one product per G tile does); bit 4 + p: product p yields ONLY that (no taps);
                            // bit 8 + p: product p is an MX job (q records, 8-bit MFMAs)
    unsigned xsub;          // byte xi: tap pattern of X chunk xi (WgradJob::xsub)
};

// "Layer mode": ONE convolution whose products form a regular (X chunk, G tile) grid -- nck x nct of them, more than the
// argument block's job table holds (the discriminator's 256..512-channel layers: 128..512 products on 16 K..65 K pixels).  Quad
// job j = (X chunk pair j % nxp, G tile pair j / nxp); everything a table entry would hold follows from the indices, so a
// whole layer is one launch pair whatever its product count: product (ct, ck) has its slabs at (ct * nck + ck) * splits * kSlab.
struct WgradLayer {
    const char* x; const char* g;       // first X chunk / first G tile
    long x_chunk_b, g_chunk_b;          // bytes between consecutive 32-channel chunks / tiles
    unsigned xstride_b, gstride_b;
    int nck, nct, nxp;                  // X chunks, G tiles, X chunk pairs ((nck + 1) / 2); nxp == 0: table mode
    int x_s2d_c;                        // > 0: X is a space-to-depth image with this many channels per sub-position (sparse taps)
    int want_bias;
    // RESR_F16X2 with three tap-products per product: quad job j >= nq repeats quad j - nq on (x_hi, g_lo), j >= 2 nq on (x_lo, g_hi);
    // part p's slabs start p * part_slabs floats behind part 0's
    int nparts, ngp;                    // 1 or 3; G tile pairs ((nct + 1) / 2)
    long x_lo_b, g_lo_b;                // bytes from the hi to the lo tensor
    unsigned part_slabs;
};

struct WgradQuadArgs {
    WgradQuad jobs[kMaxQuads];
    WgradLayer layer;
    float* partial;
    const char* zero;
    int n, h, w_, hs, ws;
    int up, splits, njobs;
    int tiles_x, tiles_y, ntiles;
};

static_assert(sizeof(WgradQuadArgs) <= 4096, "kernel arguments");

constexpr int kQXW = 22;                          // waves of LDS-DMA lanes per X chunk (10 x 34 px x 4 pieces = 1360 slots)
constexpr int kQXCH = kQXW * 1024;                // bytes per X chunk image in LDS
constexpr int kQGT = 16 * 1024;                   // bytes per G tile image (8 x 32 px x 64 B)
constexpr int kQBUF = 2 * kQXCH + 2 * kQGT;       // one stage: 77,824 B; two stages = 152 KB of the 160 KB LDS

// MXK: the MX instantiation (x2_plan bit 9) -- every product of every quad of the launch is an MX job (or empty); its own kernel, because the
// two tile loops as alternatives of one branch make the register allocator spill the 144 accumulator registers (252 spilled VGPRs).
template <bool MXK>
__global__ __launch_bounds__(512, 1) void wgrad_quad_kernel_t(const WgradQuadArgs a) {
    constexpr int PB = 64, HW = 34;
    constexpr int NSX = 6, NSG = 4;               // LDS-DMA instructions per wave per tile: 44 X waves / 8, 32 G waves / 8
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bx = blockIdx.x & 7, bk = blockIdx.x >> 3;   // all jobs of one pixel split on one XCD (see wgrad_kernel)
    int split = bx + 8 * (bk / a.njobs), jq = bk % a.njobs;
    if (a.layer.nxp > 0) {
        // layer mode: the (quad job, split) pairs as one list -- X chunk pair fastest, then the split, then the G tile pair -- cut
        // into eight contiguous ranges, one per XCD (block b runs on XCD b % 8): the 16..32 workgroups an XCD holds share their
        // G tiles and pixel range.  (Table mode's "split s on XCD s % 8" leaves six XCDs idle when a layer has two splits.)
        const int W = a.njobs * a.splits, per = (W + 7) >> 3;
        const int wi = bx * per + bk;
        if (bk >= per || wi >= W) return;
        const int xp = wi % a.layer.nxp, t = wi / a.layer.nxp;
        split = t % a.splits;
        jq = (t / a.splits) * a.layer.nxp + xp;
    } else if (split >= a.splits) return;
    // scalar copies: a field read inside the MFMA loop would be an s_load + lgkmcnt(0) in the middle of the LDS reads
    const char *jx0, *jx1, *jg0, *jg1;
    unsigned sx0, sx1, sg0, sg1, j_xsub, j_bias_mask, j_slab0, j_slab1, j_slab2, j_slab3;
    if (a.layer.nxp > 0) {   // layer mode: the quad's operands and slabs from its grid position
        const WgradLayer& L = a.layer;
        const int xp = jq % L.nxp, gpt = jq / L.nxp;
        const int part = L.nparts > 1 ? gpt / L.ngp : 0, gp = gpt - part * L.ngp;
        const int ck0 = 2 * xp, ct0 = 2 * gp;
        const bool x2nd = ck0 + 1 < L.nck, g2nd = ct0 + 1 < L.nct;
        jx0 = L.x + (size_t)ck0 * L.x_chunk_b + (part == 2 ? L.x_lo_b : 0L); jx1 = x2nd ? jx0 + L.x_chunk_b : nullptr;
        jg0 = L.g + (size_t)ct0 * L.g_chunk_b + (part == 1 ? L.g_lo_b : 0L); jg1 = g2nd ? jg0 + L.g_chunk_b : nullptr;
        sx0 = sx1 = L.xstride_b; sg0 = sg1 = L.gstride_b;
        const unsigned sub0 = L.x_s2d_c > 0 ? (unsigned)((ck0 * 32) / L.x_s2d_c) : 4u;
        const unsigned sub1 = (L.x_s2d_c > 0 && x2nd) ? (unsigned)(((ck0 + 1) * 32) / L.x_s2d_c) : 4u;
        j_xsub = sub0 | (sub1 << 8);
        const unsigned per = (unsigned)a.splits * (unsigned)kSlab;        // floats per product
        const unsigned pbase = (unsigned)part * L.part_slabs;
        auto slab = [&](int ct, int ck) { return pbase + (unsigned)(ct * L.nck + ck) * per; };
        j_slab0 = slab(ct0, ck0);
        j_slab1 = x2nd ? slab(ct0, ck0 + 1) : ~0u;
        j_slab2 = g2nd ? slab(ct0 + 1, ck0) : ~0u;
        j_slab3 = (x2nd && g2nd) ? slab(ct0 + 1, ck0 + 1) : ~0u;
        j_bias_mask = (L.want_bias && ck0 == 0 && part < 2) ? 0x5u : 0u;  // products p = 0 (x0, g0) and p = 2 (x0, g1); not of (x_lo, g_hi)
    } else {
        const WgradQuad& job = a.jobs[jq];
        jx0 = job.x[0]; jx1 = job.x[1]; jg0 = job.g[0]; jg1 = job.g[1];
        sx0 = job.xstride_b[0]; sx1 = job.xstride_b[1]; sg0 = job.gstride_b[0]; sg1 = job.gstride_b[1];
        j_xsub = job.xsub; j_bias_mask = job.bias_mask;
        j_slab0 = job.slab_off[0]; j_slab1 = job.slab_off[1]; j_slab2 = job.slab_off[2]; j_slab3 = job.slab_off[3];
    }
    auto slab_of = [&](int p) { return p == 0 ? j_slab0 : p == 1 ? j_slab1 : p == 2 ? j_slab2 : j_slab3; };
    const char* const zero = a.zero;
    const int img_h = a.h, img_w = a.w_, src_w = a.ws, up = a.up;
    const bool two_x = jx1 != nullptr, two_g = jg1 != nullptr;

    // lane -> slot maps (tile independent).  X wave ws = i*8 + wave: chunk ws >= 22, slot (ws % 22)*64 + lane.
    unsigned cx[NSX], cg[NSG];
#pragma unroll
    for (int i = 0; i < NSX; ++i) {
        const int ws = i * 8 + wave;
        const int chunk = ws >= kQXW ? 1 : 0;
        const unsigned s = (unsigned)(ws - kQXW * chunk) * 64 + lane;
        const unsigned c16 = s & 3, hp = s >> 2;
        const unsigned hy = (hp * 61681u) >> 21;  // hp / 34, exact below 100000
        const bool ok = ws < 2 * kQXW && s < 1360u && (chunk == 0 || two_x);
        cx[i] = ok ? (hy << 8 | (hp - hy * HW) | (c16 << 20)) : ~0u;
    }
#pragma unroll
    for (int i = 0; i < NSG; ++i) {
        const unsigned s = (unsigned)(((i & 1) * 8 + wave) * 64 + lane);   // tile = i >> 1
        const unsigned c16 = s & 3, r = s >> 2;
        cg[i] = (i < 2 || two_g) ? ((r >> 5) << 8 | (r & 31) | (c16 << 20)) : ~0u;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    struct TileAt { int x0, y0; unsigned xn, gn; };
    auto tile_at = [&](int tile) {
        const int tx = tile % a.tiles_x;
        const int t2 = tile / a.tiles_x;
        const int ty = t2 % a.tiles_y;
        const int n = t2 / a.tiles_y;
        TileAt t;
        t.x0 = tx * 32; t.y0 = ty * 8;
        t.xn = (unsigned)n * a.hs * a.ws; t.gn = (unsigned)n * a.h * a.w_;
        return t;
    };
    // Branch-free staging (the MFMA loop stays one basic block): lanes outside the image read the zero page through a
    // per-lane address select; lanes / waves without a slot are masked off inside the asm by narrowing EXEC.
    auto dma = [&](const char* src, unsigned dst, bool on) {
        const unsigned long long mask = __ballot(on);
        unsigned long long save;
        asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %1\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %3, off\n\ts_mov_b64 exec, %0"
                     : "=&s"(save) : "s"(mask), "s"(dst), "v"(src) : "memory", "m0");
    };
    auto stage_x = [&](int i, const TileAt& t, int buf, bool go) {
        const unsigned c = cx[i];
        const int ws = i * 8 + wave;
        const int chunk = ws >= kQXW ? 1 : 0;                                  // wave-uniform
        const int iy = t.y0 - 1 + (int)((c >> 8) & 0xff), ix = t.x0 - 1 + (int)(c & 0xff);
        const bool in = (unsigned)iy < (unsigned)img_h && (unsigned)ix < (unsigned)img_w;
        const unsigned pix = t.xn + (unsigned)(iy >> up) * src_w + (unsigned)(ix >> up);
        const char* src = (chunk ? jx1 : jx0) + (__umul24(pix, chunk ? sx1 : sx0) + ((c >> 16) & 0xfff));
        src = in ? src : zero;
        dma(src, lds0 + buf * kQBUF + ws * 1024, go && c != ~0u);
    };
    auto stage_g = [&](int i, const TileAt& t, int buf, bool go) {
        const unsigned c = cg[i];
        const int gt = i >> 1;
        const int iy = t.y0 + (int)((c >> 8) & 0xff), ix = t.x0 + (int)(c & 0xff);
        const bool in = iy < img_h && ix < img_w;
        const unsigned pix = t.gn + (unsigned)iy * img_w + (unsigned)ix;
        const char* src = (gt ? jg1 : jg0) + (__umul24(pix, gt ? sg1 : sg0) + ((c >> 16) & 0xfff));
        src = in ? src : zero;
        dma(src, lds0 + buf * kQBUF + 2 * kQXCH + (i * 8 + wave) * 1024, go && c != ~0u);
    };
    auto stage_slot = [&](int k, const TileAt& t, int buf, bool go) {   // k = 0..9
        if (k < NSX) stage_x(k, t, buf, go);
        else stage_g(k - NSX, t, buf, go);
    };

    float16v acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;

    const int prod = wave >> 1, khalf = wave & 1;
    const int xi = prod & 1, gi = prod >> 1;
    const int kh = lane >> 5, a16 = lane & 15;
    const int chb = ((((lane >> 4) & 1) << 4) + ((a16 & 3) << 2)) * 2;
    const int pxl = (kh << 3) + (a16 >> 2);
    const int goff = 2 * kQXCH + gi * kQGT + ((khalf * 4 * 32 + pxl) * PB) + chb;
    const int xoff = xi * kQXCH + ((khalf * 4 * HW + pxl) * PB) + chb;

    int tile = split;
    {
        const TileAt t0 = tile_at(tile < a.ntiles ? tile : 0);
#pragma unroll
        for (int k = 0; k < NSX + NSG; ++k) stage_slot(k, t0, 0, tile < a.ntiles);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the asm LDS-DMA is not counted by the compiler
    __syncthreads();
    // The tile loop, instantiated per tap pattern of this wave's X chunk (MODE 0..3: sub-position (i, j) = (MODE >> 1, MODE & 1)
    // of a space-to-depth image: valid dy in {1,2} for i = 0, {0,1} for i = 1, same for dx / j -- 4 of the 9 accumulators, 4/9
    // of the MFMAs and 2/3 of the X fragment reads; MODE 4: all taps).  Waves of one workgroup may run different
    // instantiations: every one executes the same staging instructions and one barrier per tile.
    auto tile_loop = [&](auto mode_c) {
        constexpr int MODE = decltype(mode_c)::value;
        // MODE 5: no taps at all -- a product nobody wants (a quad's empty slot) or one that exists for its bias sum alone: the wave
        // stages, reads its G fragments (the sum) and keeps the barriers, its accumulators stay zero
        auto vdy = [](int dy) { return MODE != 5 && (MODE == 4 || (MODE >> 1 == 0 ? dy >= 1 : dy <= 1)); };
        auto vdx = [](int dx) { return MODE != 5 && (MODE == 4 || ((MODE & 1) == 0 ? dx >= 1 : dx <= 1)); };
        int it = 0;
        for (; tile < a.ntiles; tile += a.splits, ++it) {
            const int next = tile + a.splits;
            const bool has_next = next < a.ntiles;
            const TileAt tn = tile_at(has_next ? next : tile);
            const int nb = (it + 1) & 1;
            const char* gbase = smem + (it & 1) * kQBUF + goff;
            const char* xbase = smem + (it & 1) * kQBUF + xoff;
            // X-row-major walk: halo row j of the wave's 6 serves output rows j (dy 0), j-1 (dy 1), j-2 (dy 2), so every
            // X fragment is read from LDS once (88 transpose reads per tile instead of 160 -- the LDS pipe, not the
            // matrix pipe, was the limit); the G fragments of the last three output rows stay in registers.
            // Software pipeline: the transpose reads of step st+1 are issued before the MFMAs of step st (a step = one
            // halo row j of one 16-pixel half kbh), so the matrix pipe never waits for an LDS round trip inside a tile.
            uint4 fg[2][4], fb[2][3];
            auto fload = [&](int st) {
                const int kbh = st / 6, j = st % 6;
                if (j < 4) {
                    const char* ga = gbase + (j * 32 + kbh * 16) * PB;
                    const uint2 alo = tr_read(ga), ahi = tr_read(ga + 4 * PB);
                    fg[kbh][j] = make_uint4(alo.x, alo.y, ahi.x, ahi.y);
                }
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    if (!vdx(dx)) continue;
                    const char* xa = xbase + (j * HW + kbh * 16 + dx) * PB;
                    const uint2 lo = tr_read(xa), hi = tr_read(xa + 4 * PB);
                    fb[st & 1][dx] = make_uint4(lo.x, lo.y, hi.x, hi.y);
                }
            };
            fload(0);
#pragma unroll
            for (int st = 0; st < 12; ++st) {
                const int kbh = st / 6, j = st % 6;
                if (st + 1 < 12) fload(st + 1);
                if (j < 4) bsum += sum8_f16(fg[kbh][j]);     // 4 dot instructions; only written where a bias is wanted
                // the next tile's ten requests go out during the first five steps, two per step, each behind a group of MFMAs:
                // the last one then has seven steps (~2.5 us) to land before the tile barrier
                int slot = 2 * st;
#pragma unroll
                for (int dy = 2; dy >= 0; --dy) {    // descending: the accumulators the previous step touched last come last
                    const int r = j - dy;
                    if (r < 0 || r > 3 || !vdy(dy)) continue;
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        if (!vdx(dx)) continue;
                        acc[dy * 3 + dx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, fg[kbh][r]),
                                                                                  __builtin_bit_cast(half8, fb[st & 1][dx]), acc[dy * 3 + dx], 0, 0, 0);
                    }
                    if (slot < 2 * st + 2 && slot < NSX + NSG) { stage_slot(slot, tn, nb, has_next); ++slot; }
                }
#pragma unroll
                for (int k = 0; k < 2; ++k)   // steps with fewer than two MFMA groups still issue their two requests
                    if (slot < 2 * st + 2 && slot < NSX + NSG) { stage_slot(slot, tn, nb, has_next); ++slot; }
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the next tile's asm LDS-DMA has landed
            __syncthreads();
        }
    };
    using M0 = std::integral_constant<int, 0>;
    using M1 = std::integral_constant<int, 1>;
    using M2 = std::integral_constant<int, 2>;
    using M3 = std::integral_constant<int, 3>;
    using M4 = std::integral_constant<int, 4>;
    using M5 = std::integral_constant<int, 5>;
    // An MX job (x2_plan bit 9): the staged records are q records -- per pixel 32 bytes bf8(hi) | 32 bytes bf8(lo) -- and the wave computes
    // BOTH 2^-12-weighted tap-products of its product in one pass: per output row and tap ONE v_mfma_scale_f32_32x32x64_f8f6f4 whose K = 64
    // is the row's 32 pixels twice, block 0 = (g_lo, x_hi), block 1 = (g_hi, x_lo), unit scales.  Fragments come from 8-bit transpose
    // reads: ds_read_b64_tr_b8 hands lane n of a 16-lane group column n of an 8 x 16 byte block, i.e. 8 consecutive pixels of one channel
    // (tools/micro/tr8_probe.hip); four reads make a 32-byte operand.  Any permutation of K inside a block is harmless as long as both
    // operands share it, which they do (same read pattern on both tiles).  Same staging instructions and barriers as the f16 loop.
    auto tile_loop_mx = [&]() {
        typedef int v2i __attribute__((ext_vector_type(2)));
        typedef int v8i __attribute__((ext_vector_type(8)));
        auto tr8 = [&](const char* p) -> v2i {
            return __builtin_amdgcn_ds_read_tr8_b64_v2i32(reinterpret_cast<__attribute__((address_space(3))) v2i*>((__attribute__((address_space(3))) char*)p));
        };
        // lane -> its 8-byte piece of an 8-pixel x 16-byte block: pixel (a16 >> 1), piece (a16 & 1); channel group (lane >> 4) & 1; pixel half kh
        const int lane_off = ((kh << 4) + (a16 >> 1)) * PB + (((lane >> 4) & 1) << 4) + ((a16 & 1) << 3);
        // operand of one pixel row: [first part: 16 pixels | second part: 16 pixels] of this lane's channel
        auto frag = [&](const char* row, int first_part_off, int second_part_off) -> v8i {
            const v2i a0 = tr8(row + first_part_off), a1 = tr8(row + first_part_off + 8 * PB);
            const v2i b0 = tr8(row + second_part_off), b1 = tr8(row + second_part_off + 8 * PB);
            return v8i{a0.x, a0.y, a1.x, a1.y, b0.x, b0.y, b1.x, b1.y};
        };
        const int one = 0x7f7f7f7f;
        int it = 0;
        for (; tile < a.ntiles; tile += a.splits, ++it) {
            const int next = tile + a.splits;
            const bool has_next = next < a.ntiles;
            const TileAt tn = tile_at(has_next ? next : tile);
            const int nb = (it + 1) & 1;
            const char* gb = smem + (it & 1) * kQBUF + 2 * kQXCH + gi * kQGT + (khalf * 4 * 32) * PB + lane_off;
            const char* xb = smem + (it & 1) * kQBUF + xi * kQXCH + (khalf * 4 * HW) * PB + lane_off;
            // the next tile's ten LDS-DMA requests go out up front (between the MFMA groups, as the f16 loop spreads them, their address
            // selects became branches inside the unrolled loop and the accumulators spilled -- 158 to 504 registers in three placements
            // tried); they land under the 36 MFMAs, the wave pays their issue back-pressure before its first MFMA of the tile
#ifndef RESR_WGRAD_MX_PREP
#pragma unroll
            for (int k = 0; k < NSX + NSG; ++k) stage_slot(k, tn, nb, has_next);
            __builtin_amdgcn_sched_barrier(0);
#else
            // the next tile's requests: addresses, lane masks and LDS destinations worked out HERE (selects and all), the ten LDS-DMA
            // instructions themselves go out as bare asm between the MFMA groups below
            const char* ssrc[NSX + NSG];
            unsigned long long smask[NSX + NSG];
            unsigned sdst[NSX + NSG];
#pragma unroll
            for (int k = 0; k < NSX + NSG; ++k) {
                if (k < NSX) {
                    const unsigned c = cx[k];
                    const int ws = k * 8 + wave;
                    const int chunk = ws >= kQXW ? 1 : 0;
                    const int iy = tn.y0 - 1 + (int)((c >> 8) & 0xff), ix = tn.x0 - 1 + (int)(c & 0xff);
                    const bool in = (unsigned)iy < (unsigned)img_h && (unsigned)ix < (unsigned)img_w;
                    const unsigned pix = tn.xn + (unsigned)(iy >> up) * src_w + (unsigned)(ix >> up);
                    const char* src = (chunk ? jx1 : jx0) + (__umul24(pix, chunk ? sx1 : sx0) + ((c >> 16) & 0xfff));
                    ssrc[k] = in ? src : zero;
                    smask[k] = __ballot(has_next && c != ~0u);
                    sdst[k] = lds0 + nb * kQBUF + ws * 1024;
                } else {
                    const int i = k - NSX;
                    const unsigned c = cg[i];
                    const int gt = i >> 1;
                    const int iy = tn.y0 + (int)((c >> 8) & 0xff), ix = tn.x0 + (int)(c & 0xff);
                    const bool in = iy < img_h && ix < img_w;
                    const unsigned pix = tn.gn + (unsigned)iy * img_w + (unsigned)ix;
                    const char* src = (gt ? jg1 : jg0) + (__umul24(pix, gt ? sg1 : sg0) + ((c >> 16) & 0xfff));
                    ssrc[k] = in ? src : zero;
                    smask[k] = __ballot(has_next && c != ~0u);
                    sdst[k] = lds0 + nb * kQBUF + 2 * kQXCH + (i * 8 + wave) * 1024;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#endif
            v8i fgm[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) fgm[r] = frag(gb + r * 32 * PB, 32, 0);          // A = [g_lo | g_hi]
            if ((j_bias_mask >> prod) & 1u) {     // wave-uniform: the product that carries the bias also sums g_lo's share of it -- the first
                // four dwords of a G fragment are 16 pixels of bf8(g_lo) of this lane's channel, and a bf8 byte IS the upper byte of the f16
                // with the same value: two bytes -> one f16 pair by a byte permute, summed in fp32 by a dot product against (1, 1)
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                const h2 ones = {(_Float16)1.f, (_Float16)1.f};
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const unsigned d = (unsigned)fgm[r][k];
                        bsum = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, __builtin_amdgcn_perm(d, 0u, 0x050c040cu)), ones, bsum, false);
                        bsum = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, __builtin_amdgcn_perm(d, 0u, 0x070c060cu)), ones, bsum, false);
                    }
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) {          // halo row j of the wave's six serves output rows j (dy 0), j - 1 (dy 1), j - 2 (dy 2)
                v8i fx[3];
                __builtin_amdgcn_sched_barrier(0);   // (one halo row's fragments live at a time: 144 accumulator + 32 G registers leave room for no more)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) fx[dx] = frag(xb + (j * HW + dx) * PB, 0, 32);   // B = [x_hi | x_lo]
#pragma unroll
                for (int dy = 2; dy >= 0; --dy) {
                    const int r = j - dy;
                    if (r < 0 || r > 3) continue;
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
                        acc[dy * 3 + dx] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fgm[r], fx[dx], acc[dy * 3 + dx], 1, 1, 0, one, 0, one);
                }
#ifdef RESR_WGRAD_MX_PREP
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    const int k = 2 * j + q2;
                    if (k < NSX + NSG) {
                        unsigned long long save;
                        asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %1\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                                     "global_load_lds_dwordx4 %3, off\n\ts_mov_b64 exec, %0"
                                     : "=&s"(save) : "s"(smask[k]), "s"(sdst[k]), "v"(ssrc[k]) : "memory", "m0");
                    }
                }
#endif
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the next tile's asm LDS-DMA has landed
            __syncthreads();
        }
    };
    const bool no_taps = slab_of(prod) == ~0u || ((j_bias_mask >> (4 + prod)) & 1u);   // wave-uniform
    if constexpr (MXK) {
        tile_loop_mx();     // (an empty product slot multiplies whatever its LDS region holds and is not written: one loop body, no spills)
    } else {
        if (no_taps) tile_loop(M5{});
        else switch ((j_xsub >> (8 * xi)) & 0xffu) {   // wave-uniform
            case 0: tile_loop(M0{}); break;
            case 1: tile_loop(M1{}); break;
            case 2: tile_loop(M2{}); break;
            case 3: tile_loop(M3{}); break;
            default: tile_loop(M4{}); break;
        }
    }

    // ---- sum the two row-halves of every product through LDS, write the slabs ---------------------------------
    float* red = reinterpret_cast<float*>(smem);               // [8 waves][16 regs][64 lanes]
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[tap][r];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + i * 512;
            const int p = e >> 10, q = e & 1023;                 // p is wave-uniform (512 | 1024)
            const int r = q >> 6, l = q & 63;
            const float s = red[(2 * p) * 1024 + q] + red[(2 * p + 1) * 1024 + q];
            const int co = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), ci = l & 31;
            const unsigned so = slab_of(p);
            if (so != ~0u) a.partial[so + (size_t)split * kSlab + tap * 1024 + co * 32 + ci] = s;
        }
        __syncthreads();
    }
    red[wave * 64 + lane] = bsum;
    __syncthreads();
    if (tid < 128) {
        const int p = tid >> 5, c = tid & 31;
        const float s = (red[(2 * p) * 64 + c] + red[(2 * p) * 64 + 32 + c]) + (red[(2 * p + 1) * 64 + c] + red[(2 * p + 1) * 64 + 32 + c]);
        const unsigned so = slab_of(p);
        if (((j_bias_mask >> p) & 1) && so != ~0u)
            a.partial[so + (size_t)split * kSlab + 9 * 1024 + c] = s;
    }
}

#define wgrad_quad_kernel wgrad_quad_kernel_t<false>

// deterministic slab reduction -> OIHW fp32 gradient (+ bias gradient).  32 slab elements per workgroup, the splits dealt
// to eight thread groups (a thread sums splits/8 slabs: the launch is latency-bound at small images -- 19 us with four
// groups of 64 elements -- and the slabs of a full-size dense block are 69 MB) and combined in a fixed order.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const ReduceArgs a) {
    __shared__ float red[8][32];
    ReduceJob job = a.jobs[a.layer_nck > 0 ? 0 : blockIdx.x];
    if (a.layer_nck > 0) {
        const int ct = blockIdx.x / a.layer_nck, ck = blockIdx.x - ct * a.layer_nck;
        job.slab_off = blockIdx.x * (unsigned)a.splits * (unsigned)kSlab;
        if (a.layer_part_stride) { job.slab_b = job.slab_off + a.layer_part_stride; job.slab_c = job.slab_b + a.layer_part_stride; }
        job.co_base = (short)(ct * 32); job.ci_base = (short)(ck * 32);
        job.want_bias = (short)((ck == 0 && job.db) ? 1 : 0);
    }
    const int lane = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int e = blockIdx.y * 32 + lane;
    float s = 0.f;
    // a thread's slabs k = g, g + 8, ... are requested eight at a time and then added in that order: the plain loop waited
    // for every load before issuing the next (4-9 dependent memory round trips at the 36-72 splits of a training launch)
    auto sum_splits = [&](const float* p, float acc, int nsplits) {
        constexpr int U = 8;
        for (int k0 = g; k0 < nsplits; k0 += 8 * U) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = k0 + 8 * u;
                v[u] = k < nsplits ? p[(size_t)k * kSlab] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (k0 + 8 * u < nsplits) acc += v[u];
        }
        return acc;
    };
    if (e < kSlab) {
        s = sum_splits(a.partial + job.slab_off + e, 0.f, a.splits);
        if (job.slab_b != ~0u || job.slab_c != ~0u) {   // RESR_F16X2: the cross products (bias sums: only the g_lo one; a single-f16 G has none)
            float s2 = job.slab_b != ~0u ? sum_splits(a.partial + job.slab_b + e, 0.f, a.splits) : 0.f;
            if (job.slab_c != ~0u && (e < 9 * 1024 || job.c_bias)) s2 = sum_splits(a.partial + job.slab_c + e, s2, a.splits_c > 0 ? a.splits_c : a.splits);
            s = __builtin_fmaf(s2, kLoInv, s);
        }
    }
    red[g][lane] = s;
    __syncthreads();
    if (g != 0 || e >= kSlab) return;
    s = (((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane])) + ((red[4][lane] + red[5][lane]) + (red[6][lane] + red[7][lane]))) * job.scale;
    if (a.unscale) {
        const unsigned mb = *a.unscale;
        s *= grad_prescale(mb, true);
        // a LIFTED pass that overflowed: tell the next pass to aim lower (common.h: the slot's sticky words)
        if (grad_prescale_lifted(mb) && !(__builtin_fabsf(s) <= 3.0e38f)) const_cast<unsigned*>(a.unscale)[2] = 1u;
    }
    if (e < 9 * 1024) {
        const int tap = e >> 10, co = job.co_base + ((e >> 5) & 31), ci = job.ci_base + (e & 31);
        if (co < job.cout && ci < job.cin_real) job.dw[((size_t)co * job.cin_real + ci) * 9 + tap] = s;
    } else if (job.want_bias && job.db) {
        const int co = job.co_base + (e - 9 * 1024);
        if (co < job.cout) job.db[co] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
int wgrad_tile_rows(int dtype) { return dtype != RESR_F32 ? 8 : 4; }

// `nprod`: algorithmic products of the launch (= njobs, or njobs / 3 with RESR_F16X2) for the profiling record
template <typename T, int RPW>
static int launch_wgrad(WgradArgs& a, int njobs, int nprod, hipStream_t stream) {
    constexpr int PB = 32 * (int)sizeof(T);
    constexpr int TH = 4 * RPW;
    constexpr int SPP_ = PB / 16;
    constexpr int BUF = (((TH + 2) * 34 * SPP_ + 63) / 64 * 64) * 16 + TH * 32 * PB;
    a.tiles_x = (a.w_ + 31) / 32;
    a.tiles_y = (a.h + TH - 1) / TH;
    a.ntiles = a.tiles_x * a.tiles_y * a.n;
    const size_t lds = 2 * BUF;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<T, RPW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    a.njobs = njobs;
    {
        void* zp = nullptr;
        if (hipGetSymbolAddress(&zp, HIP_SYMBOL(g_zero16)) != hipSuccess || !zp) return fail(RESR_ERR_LAUNCH, "wgrad: zero page");
        a.zero = (const char*)zp;
    }
    const int splits8 = (a.splits + 7) / 8 * 8;
    prof_before(stream);
    hipLaunchKernelGGL((wgrad_kernel<T, RPW>), dim3(njobs * splits8), dim3(256), lds, stream, a);
    // algorithmic bytes: each job's X chunk and G tile once (jobs that share a tile re-read it from L2, not counted twice
    // would need the conv list; this is the upper, per-job figure)
    prof_after(stream, 50000 + (sizeof(T) == 2 ? 0 : 100) + RPW + (nprod != njobs ? 300 : 0), 2.0 * 9 * 32 * 32 * nprod * (double)a.n * a.h * a.w_,
               (double)njobs * 2 * 32 * sizeof(T) * (double)a.n * a.h * a.w_);
    RESR_CHECK_LAUNCH("wgrad_kernel");
    return RESR_OK;
}

// Group the (X chunk, G tile) pairs of a launch into 2x2 quads: greedily take the pair of X chunks and pair of G tiles
// that covers the most still-unassigned products (ties: fewer staged operands).  A dense block's 26 products become 6 full
// quads + one diagonal (2 products); a single 64->64 conv is one quad.  The search only depends on which (X, G) index
// pairs are wanted, so its result is cached per pattern (a RESR_F16X2 dense block has 78 products over 12 x 12 operands:
// ~10^5 candidate evaluations, once).
struct QuadIdx { short xa, xb, ga, gb; short prod[4]; };   // operand indices; job index per slot or -1

static const std::vector<QuadIdx>* plan_quads(const int* jx, const int* jg, int nj, int nx, int ng) {
    static std::mutex mu;
    static std::map<std::vector<short>, std::vector<QuadIdx>> cache;
    std::vector<short> key;
    key.reserve(2 * nj + 2);
    key.push_back((short)nx); key.push_back((short)ng);
    for (int i = 0; i < nj; ++i) { key.push_back((short)jx[i]); key.push_back((short)jg[i]); }
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second.empty() ? nullptr : &it->second;
    std::vector<QuadIdx>& out = cache[key];
    std::vector<short> need((size_t)nx * ng, 0);   // job index + 1, 0 = not wanted / already assigned
    auto N = [&](int x, int g) -> short& { return need[(size_t)x * ng + g]; };
    for (int i = 0; i < nj; ++i) {
        if (N(jx[i], jg[i])) return nullptr;                 // the same product twice: keep the pair kernel (cached as empty)
        N(jx[i], jg[i]) = (short)(i + 1);
    }
    int left = nj;
    while (left > 0) {
        int best = -1, bxa = 0, bxb = 0, bga = 0, bgb = 0;
        for (int xa = 0; xa < nx; ++xa)
            for (int xb = xa; xb < nx; ++xb)
                for (int ga = 0; ga < ng; ++ga)
                    for (int gb = ga; gb < ng; ++gb) {
                        int cnt = (N(xa, ga) != 0);
                        if (xb != xa) cnt += (N(xb, ga) != 0);
                        if (gb != ga) cnt += (N(xa, gb) != 0);
                        if (xb != xa && gb != ga) cnt += (N(xb, gb) != 0);
                        if (!cnt) continue;
                        const int score = cnt * 8 - (xb != xa) - (gb != ga);
                        if (score > best) { best = score; bxa = xa; bxb = xb; bga = ga; bgb = gb; }
                    }
        if (best < 0 || (int)out.size() >= kMaxQuads) { out.clear(); return nullptr; }
        QuadIdx q;
        q.xa = (short)bxa; q.xb = (short)bxb; q.ga = (short)bga; q.gb = (short)bgb;
        const int qx[2] = {bxa, bxb}, qg[2] = {bga, bgb};
        for (int p = 0; p < 4; ++p) {
            const int xi = p & 1, gi = p >> 1;
            q.prod[p] = -1;
            if ((xi && bxb == bxa) || (gi && bgb == bga)) continue;
            short& n = N(qx[xi], qg[gi]);
            if (!n) continue;
            q.prod[p] = (short)(n - 1);
            n = 0;
            --left;
        }
        out.push_back(q);
    }
    return &out;
}

// quads of the jobs of one kind (mx = 0: the f16 tap-products, 1: the MX jobs -- their operands are q tensors, so the two kinds never share
// an operand, and they run as two launches of two kernel instantiations); 0 when there is no job of the kind
static int build_quads(const WgradArgs& a, int nj_all, WgradQuadArgs& q, unsigned mx = 0) {
    const char* xs[kMaxJobs]; const char* gs[kMaxJobs];
    unsigned xstr[kMaxJobs], gstr[kMaxJobs], xsub[kMaxJobs];
    int nx = 0, ng = 0, nj = 0;
    int jx[kMaxJobs], jg[kMaxJobs], jid[kMaxJobs];
    for (int i = 0; i < nj_all; ++i) {
        if (a.jobs[i].mx != mx) continue;
        int xi = 0, gi = 0;
        while (xi < nx && xs[xi] != a.jobs[i].x) ++xi;
        if (xi == nx) { xs[nx] = a.jobs[i].x; xstr[nx] = a.jobs[i].xstride_b; xsub[nx] = a.jobs[i].xsub; ++nx; }
        while (gi < ng && gs[gi] != a.jobs[i].g) ++gi;
        if (gi == ng) { gs[ng] = a.jobs[i].g; gstr[ng] = a.jobs[i].gstride_b; ++ng; }
        jx[nj] = xi; jg[nj] = gi; jid[nj] = i; ++nj;
    }
    if (nj == 0) return 0;
    const std::vector<QuadIdx>* plan = plan_quads(jx, jg, nj, nx, ng);
    if (!plan) return -1;
    int nq = 0;
    for (const QuadIdx& qi : *plan) {
        WgradQuad& w = q.jobs[nq++];
        memset(&w, 0, sizeof(w));
        w.x[0] = xs[qi.xa]; w.xstride_b[0] = xstr[qi.xa];
        w.xsub = xsub[qi.xa] | (4u << 8);
        if (qi.xb != qi.xa) { w.x[1] = xs[qi.xb]; w.xstride_b[1] = xstr[qi.xb]; w.xsub = xsub[qi.xa] | (xsub[qi.xb] << 8); }
        w.g[0] = gs[qi.ga]; w.gstride_b[0] = gstr[qi.ga];
        if (qi.gb != qi.ga) { w.g[1] = gs[qi.gb]; w.gstride_b[1] = gstr[qi.gb]; }
        for (int p = 0; p < 4; ++p) {
            w.slab_off[p] = ~0u;
            if (qi.prod[p] < 0) continue;
            const WgradJob& j = a.jobs[jid[qi.prod[p]]];
            w.slab_off[p] = j.slab_off;
            if (j.want_bias & 1u) w.bias_mask |= 1u << p;
            if (j.want_bias & 2u) w.bias_mask |= 16u << p;
            if (j.mx) w.bias_mask |= 256u << p;
        }
    }
    return nq;
}

template <bool MXK>
static int launch_wgrad_quad_kind(const WgradArgs& a, int nj, int nprod, hipStream_t stream, bool* done);

static int launch_wgrad_quad(const WgradArgs& a, int nj, int nprod, hipStream_t stream, bool* done) {
    bool any_mx = false;
    for (int i = 0; i < nj; ++i) any_mx = any_mx || a.jobs[i].mx;
    const int rc = launch_wgrad_quad_kind<false>(a, nj, nprod, stream, done);
    if (rc || !*done || !any_mx) return rc;
    *done = false;     // the MX jobs' quads: a second launch (wgrad_quad_kernel_t<true>); no pair-kernel fallback exists for them
    return launch_wgrad_quad_kind<true>(a, nj, nprod, stream, done);
}

template <bool MXK>
static int launch_wgrad_quad_kind(const WgradArgs& a, int nj, int nprod, hipStream_t stream, bool* done) {
    static thread_local WgradQuadArgs q;
    *done = false;
    const int nq = build_quads(a, nj, q, MXK ? 1u : 0u);
    if (nq <= 0) return RESR_OK;
    q.partial = a.partial;
    q.n = a.n; q.h = a.h; q.w_ = a.w_; q.hs = a.hs; q.ws = a.ws; q.up = a.up; q.splits = MXK ? a.splits_mx : a.splits; q.njobs = nq;
    q.tiles_x = (a.w_ + 31) / 32;
    q.tiles_y = (a.h + 7) / 8;
    q.ntiles = q.tiles_x * q.tiles_y * a.n;
    const size_t lds = 2 * kQBUF;
    static bool attr_done = false;     // (one flag per instantiation of this function template)
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_quad_kernel_t<MXK>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return fail(RESR_ERR_LAUNCH, "wgrad: cannot reserve %zu B of LDS", lds);
        attr_done = true;
    }
    void* zp = nullptr;
    if (hipGetSymbolAddress(&zp, HIP_SYMBOL(g_zero16)) != hipSuccess || !zp) return fail(RESR_ERR_LAUNCH, "wgrad: zero page");
    q.zero = (const char*)zp;
    const int splits8 = (q.splits + 7) / 8 * 8;
    double staged = 0;
    for (int i = 0; i < nq; ++i) staged += 1 + (q.jobs[i].x[1] != nullptr) + 1 + (q.jobs[i].g[1] != nullptr);
    prof_before(stream);
    hipLaunchKernelGGL(wgrad_quad_kernel_t<MXK>, dim3(nq * splits8), dim3(512), lds, stream, q);
    // algorithmic bytes: every quad's X chunks and G tiles once (quads that share an operand find it in their XCD's L2)
    // (the MX launch: kernel id 50600, no algorithmic FLOPs of its own -- the products are counted with the f16 launch)
    prof_after(stream, MXK ? 50600 : (nprod != nj ? 50500 : 50200), MXK ? 0.0 : 2.0 * 9 * 32 * 32 * nprod * (double)a.n * a.h * a.w_, staged * 64.0 * (double)a.n * a.h * a.w_);
    RESR_CHECK_LAUNCH("wgrad_quad_kernel");
    *done = true;
    return RESR_OK;
}

// resr_debug_wgrad_plan (include/resr.h): the grouping above on synthetic operand addresses
int wgrad_debug_plan(const int* cin, const int* cout_pad, int nconv, int* out, int max_jobs) {
    if (!cin || !cout_pad || !out || nconv <= 0 || max_jobs <= 0) return fail(RESR_ERR_ARG, "wgrad plan: null argument");
    static thread_local WgradArgs a;
    static thread_local WgradQuadArgs q;
    memset(&a, 0, sizeof(a));
    const char* xbase = reinterpret_cast<const char*>(0x10000000);   // never dereferenced
    const char* gbase = reinterpret_cast<const char*>(0x20000000);
    int nj = 0, gt = 0;
    for (int i = 0; i < nconv; ++i) {
        if (cin[i] <= 0 || (cin[i] & 31) || (cout_pad[i] != 32 && cout_pad[i] != 64)) return fail(RESR_ERR_ARG, "wgrad plan: cin=%d cout_pad=%d", cin[i], cout_pad[i]);
        for (int ct = 0; ct < cout_pad[i] / 32; ++ct, ++gt)
            for (int ck = 0; ck < cin[i] / 32; ++ck) {
                if (nj >= kMaxJobs) return fail(RESR_ERR_ARG, "wgrad plan: more than %d products", kMaxJobs);
                a.jobs[nj].x = xbase + ck * 64;
                a.jobs[nj].g = gbase + gt * 64;
                a.jobs[nj].slab_off = (unsigned)nj;     // product index
                a.jobs[nj].xsub = 4;
                ++nj;
            }
    }
    const int nq = build_quads(a, nj, q);
    if (nq < 0 || nq > max_jobs) return fail(RESR_ERR_ARG, "wgrad plan: %d jobs", nq);
    for (int i = 0; i < nq; ++i)
        for (int p = 0; p < 4; ++p) out[i * 4 + p] = q.jobs[i].slab_off[p] == ~0u ? -1 : (int)q.jobs[i].slab_off[p];
    return nq;
}

// One batched launch pair.  `convs` (wgrad.h) describes up to a dense block's worth of convolutions that share
// n/h/w/flags; jobs are generated as (conv, ci chunk, co tile) -- three per product with RESR_F16X2.
// RESR_F16X2: tap-products per algorithmic product of a weight gradient.  3 (default): dW = X_hi^T G_hi + 2^-12 (X_hi^T G_lo +
// X_lo^T G_hi) -- fp32-class (every one of the 702 tensors of the 23-block generator within 5.8e-6 of the float64 evaluation of the
// oracle), nothing rests on averaging.  $RESR_X2_WGRAD_PRODUCTS=1 (opt-in): the hi tensors only, dW = X_hi^T G_hi, a third of the
// matrix work.  The lo parts are rounding residues of relative size 2^-12 with zero mean, independent from pixel to pixel; what
// they add to a weight gradient is a random walk next to the gradient's own sum, so the relative error does NOT shrink with the
// pixel count: it sits at ~2^-11 whatever the size and grows where the true sum cancels more than a random walk does.  Measured on
// all 702 tensors against the three-product form (tools/x2_wgrad_validate.py, profiles/r03_x2_wgrad_validate.json): worst tensor
// 9.0e-4 / 4.8e-4 / 4.1e-4 at 2 x 256^2 / 16 x 64^2 / 1 x 24^2 -- inside the 1e-3 gate with under 10 % margin on one seed per
// geometry, which is why it is not the default (round 3 had it as the default; round 4 put the three products back).
// Forward and backward-data always use the three stages.
int wgrad_x2_products() {
    const char* e = getenv("RESR_X2_WGRAD_PRODUCTS");   // read per call: a validation run flips it inside one process
    if (e && e[0] == '3') return 3;
    if (e && e[0] == '1') return 1;
    return kX2WgradProductsDefault;
}

// The jobs (kernel launches' slab regions) of ONE algorithmic product (G tile ct, X chunk ck): part 0 (x_hi, g_hi) always; RESR_F16X2 with
// three tap-products adds 1 = (x_hi, g_lo) and 2 = (x_lo, g_hi) as the plan bits allow -- or, where both operands carry q tensors and the
// X chunk is a pair (WgradConv.x_q_off / g_q_off, x2_plan bit 9), ONE job 3 = "MX" for both corrections, which also sums
// g_lo's share of the bias (X chunk 0) from the bf8 bytes of its own G fragments.  (Until late in round 6 a tap-free f16 job (x_hi chunk 0, g_lo)
// per G tile summed that share from the f16 lo tensor: six product slots per dense block that staged two tiles per step for 32 sums each --
// 7.4 ms of the 262 ms exact16 step, measured by leaving them out.  The bf8 rounding of a 2^-12-weighted term is 2^-15 of the gradient.)
// One definition for the job count, the quad count and the job table.
static int product_parts(const WgradConv& c, int dtype, int nparts, int ck, int parts[4]) {
    int n = 0;
    parts[n++] = 0;
    if (dtype != RESR_F16X2 || nparts == 1) return n;
    const bool g_single = c.g_lo_off == 0;
    const bool single_chunk = c.x_pair_chunks > 0 && ck >= c.x_pair_chunks;
    if (nparts == 3 && c.x_q_off != 0 && c.g_q_off != 0 && !g_single && !single_chunk) {
        parts[n++] = 3;     // (where a bias is wanted -- ck 0 -- the MX job also sums g_lo's share of it from its own G fragments)
        return n;
    }
    for (int part = 1; part < nparts; ++part) {
        if (part == 1 && g_single) continue;   // no g_lo: dW = X_hi^T G + 2^-12 X_lo^T G
        if (part == 1 && c.g_lo_bias_only && ck != 0) continue;   // g_lo only where the bias is summed (X chunk 0)
        if (single_chunk && (part == 2 || c.x_single_g_hi)) continue;   // this X chunk enters as its hi tensor: dW = X_hi^T G (x_single_g_hi: X_hi^T G_hi)
        parts[n++] = part;
    }
    return n;
}

// tap-products of one convolution
static size_t wgrad_conv_jobs(const WgradConv& c, int dtype) {
    const int nparts = dtype == RESR_F16X2 ? wgrad_x2_products() : 1;
    size_t jobs = 0;
    int parts[4];
    for (int ck = 0; ck < c.cin / 32; ++ck) jobs += (size_t)product_parts(c, dtype, nparts, ck, parts);
    return jobs * (size_t)(c.cout_pad / 32);
}

// Quad jobs (workgroups per pixel split) the batched launch of `convs` will run: the planners size their pixel splits by it --
// a dense block's 64 exact16 tap-products with single-f16 growth gradients group into 17 quads, not 64 / 4 (the plan is cached
// per operand pattern, so this costs a map lookup after the first call).  Falls back to ceil(jobs / 4).
int wgrad_batch_quads(const WgradConv* convs, int nconv, int dtype) {
    const size_t es = elem_size(dtype);
    const bool x2 = dtype == RESR_F16X2;
    const int nparts = x2 ? wgrad_x2_products() : 1;
    const char* xs[kMaxJobs]; const char* gs[kMaxJobs];
    int jx[kMaxJobs], jg[kMaxJobs], nx = 0, ng = 0, nj = 0, total = 0;
    for (int i = 0; i < nconv; ++i) {
        const WgradConv& c = convs[i];
        for (int ct = 0; ct < c.cout_pad / 32; ++ct)
            for (int ck = 0; ck < c.cin / 32; ++ck) {
                int parts[4];
                const int np = product_parts(c, dtype, nparts, ck, parts);
                for (int pi = 0; pi < np; ++pi) {
                    const int part = parts[pi];
                    if (part == 3) continue;     // (the MX jobs run as their own launch with their own pixel splits: mx_split_factor)
                    ++total;
                    if (nj >= kMaxJobs) continue;
                    const char* xp = (const char*)c.x0 + (size_t)ck * (c.x_chunk_stride > 0 ? c.x_chunk_stride : 32) * es + (part == 2 ? (size_t)c.x_lo_off * es : part == 3 ? (size_t)c.x_q_off * es : 0);
                    const char* gp = (const char*)c.g + (size_t)ct * (c.g_chunk_stride > 0 ? c.g_chunk_stride : 32) * es + (part == 1 ? (size_t)c.g_lo_off * es : part == 3 ? (size_t)c.g_q_off * es : 0);
                    int xi = 0, gi = 0;
                    while (xi < nx && xs[xi] != xp) ++xi;
                    if (xi == nx) xs[nx++] = xp;
                    while (gi < ng && gs[gi] != gp) ++gi;
                    if (gi == ng) gs[ng++] = gp;
                    jx[nj] = xi; jg[nj] = gi; ++nj;
                }
            }
    }
    if (total > kMaxJobs || dtype == RESR_F32) return (total + 3) / 4;
    const std::vector<QuadIdx>* plan = plan_quads(jx, jg, nj, nx, ng);
    return plan ? (int)plan->size() : (total + 3) / 4;
}

// tap-products (= slab regions, kernel jobs) of a batched launch
int wgrad_batch_jobs(const WgradConv* convs, int nconv, int dtype) {
    size_t jobs = 0;
    for (int i = 0; i < nconv; ++i) jobs += wgrad_conv_jobs(convs[i], dtype);
    return (int)jobs;
}

// MX jobs of a batch, and the factor their launch's pixel splits take over the f16 launch's: a dense block has 12 of them = three quads,
// and 3 x 32 workgroups would leave two thirds of the CUs idle -- k x splits with 3 quads x k x splits ~ one residency round (256)
static int wgrad_mx_jobs(const WgradConv* convs, int nconv, int dtype) {
    const int nparts = dtype == RESR_F16X2 ? wgrad_x2_products() : 1;
    int n = 0, parts[4];
    for (int i = 0; i < nconv; ++i)
        for (int ck = 0; ck < convs[i].cin / 32; ++ck) {
            const int np = product_parts(convs[i], dtype, nparts, ck, parts);
            for (int pi = 0; pi < np; ++pi) n += parts[pi] == 3 ? convs[i].cout_pad / 32 : 0;
        }
    return n;
}
static int mx_split_factor(int n_mx, int splits) {
    if (n_mx <= 0) return 1;
    const int nq = (n_mx + 3) / 4, s8 = (splits + 7) / 8 * 8;
    int k = 256 / (nq * s8);
    return k < 1 ? 1 : (k > 4 ? 4 : k);
}

size_t wgrad_batch_partial_bytes(const WgradConv* convs, int nconv, int splits, int dtype) {
    size_t jobs = 0;
    for (int i = 0; i < nconv; ++i) jobs += wgrad_conv_jobs(convs[i], dtype);
    const int n_mx = wgrad_mx_jobs(convs, nconv, dtype);
    jobs += (size_t)n_mx * (mx_split_factor(n_mx, splits) - 1);
    return jobs * splits * kSlab * sizeof(float);
}

// One convolution with a regular grid of more products than a launch's job table holds (WgradLayer): one launch pair for the
// whole layer.  RESR_F16 and RESR_F16X2 (nparts = 3: the (hi, lo) and (lo, hi) tap-products as two more rounds of quad jobs with
// their own slab regions, combined by the reduction exactly as in table mode); cout_pad any multiple of 32.
static int wgrad_layer_launch(const WgradConv& c, int n, int h, int w, int flags, int splits, int nparts, float* partial, hipStream_t stream) {
    const bool up = flags & RESR_CONV_UPSAMPLE_IN;
    const size_t es = 2;
    static thread_local WgradQuadArgs q;
    static thread_local ReduceArgs r;
    memset(&q, 0, sizeof(q));
    memset(&r, 0, sizeof(r));
    const int nck = c.cin / 32, nct = c.cout_pad / 32;
    WgradLayer& L = q.layer;
    L.x = (const char*)c.x0; L.g = (const char*)c.g;
    L.x_chunk_b = (c.x_chunk_stride > 0 ? c.x_chunk_stride : 32) * (long)es;
    L.g_chunk_b = (c.g_chunk_stride > 0 ? c.g_chunk_stride : 32) * (long)es;
    L.xstride_b = (unsigned)(c.in0_stride * es); L.gstride_b = (unsigned)(c.g_stride * es);
    L.nck = nck; L.nct = nct; L.nxp = (nck + 1) / 2;
    L.x_s2d_c = c.x_s2d_c; L.want_bias = c.db ? 1 : 0;
    L.nparts = nparts; L.ngp = (nct + 1) / 2;
    L.x_lo_b = c.x_lo_off * (long)es; L.g_lo_b = c.g_lo_off * (long)es;
    L.part_slabs = (unsigned)((size_t)nck * nct * splits * kSlab);
    const int nq = L.nxp * L.ngp * nparts;
    {   // 32-bit lane offsets: every operand below 2^24 pixels and 4 GB
        const size_t px = (size_t)n * h * w;
        const size_t smax = L.xstride_b > L.gstride_b ? L.xstride_b : L.gstride_b;
        if (!(px <= (1u << 24) && smax < (1u << 24) && px * smax + 64 < (1ull << 32))) return fail(RESR_ERR_ARG, "wgrad (layer mode): operand too large for 32-bit offsets");
        if ((size_t)nck * nct * splits * kSlab * nparts >= (1ull << 32)) return fail(RESR_ERR_ARG, "wgrad (layer mode): slab offsets exceed 32 bits");
    }
    q.partial = partial;
    q.n = n; q.h = h; q.w_ = w; q.hs = up ? h / 2 : h; q.ws = up ? w / 2 : w; q.up = up ? 1 : 0; q.splits = splits; q.njobs = nq;
    q.tiles_x = (w + 31) / 32; q.tiles_y = (h + 7) / 8; q.ntiles = q.tiles_x * q.tiles_y * n;
    const size_t lds = 2 * kQBUF;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_quad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return fail(RESR_ERR_LAUNCH, "wgrad: cannot reserve %zu B of LDS", lds);
        attr_done = true;
    }
    void* zp = nullptr;
    if (hipGetSymbolAddress(&zp, HIP_SYMBOL(g_zero16)) != hipSuccess || !zp) return fail(RESR_ERR_LAUNCH, "wgrad: zero page");
    q.zero = (const char*)zp;
    const int per_xcd = (nq * splits + 7) / 8;          // (quad job, split) pairs per XCD: eight contiguous ranges
    const double sparse = c.x_s2d_c > 0 ? 4.0 / 9.0 : 1.0;
    prof_before(stream);
    hipLaunchKernelGGL(wgrad_quad_kernel, dim3(per_xcd * 8), dim3(512), lds, stream, q);
    prof_after(stream, 50200, 2.0 * 9 * 32 * 32 * sparse * nck * nct * nparts * (double)n * h * w, (double)(nck + nct) * 64.0 * nparts * (double)n * h * w);
    RESR_CHECK_LAUNCH("wgrad_quad_kernel (layer mode)");
    ReduceJob& j = r.jobs[0];
    j.dw = c.dw; j.db = c.db; j.slab_off = 0; j.slab_b = j.slab_c = ~0u; j.co_base = j.ci_base = 0;
    j.cout = c.cout; j.cin_real = c.cin_real; j.scale = c.scale; j.want_bias = 0; j.c_bias = 0;
    r.partial = partial; r.splits = splits; r.layer_nck = nck; r.layer_part_stride = nparts > 1 ? L.part_slabs : 0u;
    r.unscale = c.unscale;
    if (splits <= 8) hipLaunchKernelGGL(wgrad_reduce_wide_kernel, dim3(nck * nct, (kSlab + 255) / 256), dim3(256), 0, stream, r);
    else hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nck * nct, (kSlab + 31) / 32), dim3(256), 0, stream, r);
    RESR_CHECK_LAUNCH("wgrad_reduce_kernel (layer mode)");
    return RESR_OK;
}

// floats of slab scratch a layer-mode launch needs
size_t wgrad_layer_partial_bytes(int cin, int cout_pad, int splits, int dtype) {
    return (size_t)(cin / 32) * (cout_pad / 32) * (dtype == RESR_F16X2 ? wgrad_x2_products() : 1) * splits * kSlab * sizeof(float);
}

// entry for the whole-network planners: one f16 convolution, cout_pad any multiple of 32, as one layer-mode launch pair
int wgrad_layer(const WgradConv* c, int n, int h, int w, int dtype, int flags, int splits, float* partial, hipStream_t stream) {
    if (!c || !partial || !c->x0 || !c->g || !c->dw) return fail(RESR_ERR_ARG, "wgrad_layer: null argument");
    if (dtype != RESR_F16 && dtype != RESR_F16X2) return fail(RESR_ERR_ARG, "wgrad_layer: RESR_F16 or RESR_F16X2 only");
    if (dtype == RESR_F16X2 && (c->x_lo_off <= 0 || c->g_lo_off <= 0)) return fail(RESR_ERR_ARG, "wgrad_layer: RESR_F16X2 needs the hi -> lo offsets of X and G");
    if (splits <= 0 || splits > 65535) return fail(RESR_ERR_ARG, "wgrad_layer: splits=%d", splits);
    if (c->cin <= 0 || (c->cin & 31) || c->cout_pad <= 0 || (c->cout_pad & 31) || c->cout <= 0 || c->cout > c->cout_pad || c->cin_real <= 0 || c->cin_real > c->cin)
        return fail(RESR_ERR_ARG, "wgrad_layer: cin=%d cin_real=%d cout=%d cout_pad=%d", c->cin, c->cin_real, c->cout, c->cout_pad);
    if ((flags & RESR_CONV_UPSAMPLE_IN) && ((h | w) & 1)) return fail(RESR_ERR_ARG, "wgrad_layer: upsampled input needs even h,w");
    return wgrad_layer_launch(*c, n, h, w, flags, splits, dtype == RESR_F16X2 ? wgrad_x2_products() : 1, partial, stream);
}

int wgrad_batch(const WgradConv* convs, int nconv, int n, int h, int w, int dtype, int flags, int splits,
                float* partial, hipStream_t stream) {
    if (!convs || nconv <= 0 || !partial) return fail(RESR_ERR_ARG, "wgrad: null argument");
    if (splits <= 0 || splits > 65535) return fail(RESR_ERR_ARG, "wgrad: splits=%d", splits);
    const bool up = flags & RESR_CONV_UPSAMPLE_IN;
    if (up && ((h | w) & 1)) return fail(RESR_ERR_ARG, "wgrad: upsampled input needs even h,w");
    const size_t es = elem_size(dtype);
    WgradArgs a;
    ReduceArgs r;
    memset(&a, 0, sizeof(a));
    memset(&r, 0, sizeof(r));
    int nj = 0, nr = 0;
    unsigned off = 0;
    bool any_mx = false;
    const bool x2 = dtype == RESR_F16X2;
    const int nparts = x2 ? wgrad_x2_products() : 1;
    int splits_mx = splits;   // the MX launch's own pixel splits: a multiple of `splits`, at least two tiles per workgroup (the size check above assumes the uncapped factor)
    {
        const long tiles = (long)((w + 31) / 32) * ((h + 7) / 8) * n;
        int k = mx_split_factor(wgrad_mx_jobs(convs, nconv, dtype), splits);
        while (k > 1 && (long)splits * k > tiles / 2) --k;
        splits_mx = splits * k;
    }
    for (int i = 0; i < nconv; ++i) {
        const WgradConv& c = convs[i];
        if (!c.x0 || !c.g || !c.dw) return fail(RESR_ERR_ARG, "wgrad: null tensor");
        if (x2 && (c.x_lo_off <= 0 || c.g_lo_off < 0)) return fail(RESR_ERR_ARG, "wgrad: RESR_F16X2 needs the hi -> lo offsets of X and G (g_lo_offset = 0: G is a single f16 tensor)");
        if (c.cin <= 0 || (c.cin & 31) || (c.cout_pad != 32 && c.cout_pad != 64) || c.cout <= 0 || c.cout > c.cout_pad ||
            c.cin_real <= 0 || c.cin_real > c.cin)
            return fail(RESR_ERR_ARG, "wgrad: cin=%d cin_real=%d cout=%d cout_pad=%d", c.cin, c.cin_real, c.cout, c.cout_pad);
        for (int ct = 0; ct < c.cout_pad / 32; ++ct)
            for (int ck = 0; ck < c.cin / 32; ++ck) {
                if (nr >= kMaxReduce) return fail(RESR_ERR_ARG, "wgrad: more than %d products in one batch", kMaxReduce);
                const char* xh = (const char*)c.x0 + (size_t)ck * (c.x_chunk_stride > 0 ? c.x_chunk_stride : 32) * es;
                const char* gh = (const char*)c.g + (size_t)ct * (c.g_chunk_stride > 0 ? c.g_chunk_stride : 32) * es;
                const int want_bias = (ck == 0 && c.db) ? 1 : 0;
                ReduceJob& q = r.jobs[nr++];
                q.dw = c.dw; q.db = c.db; q.slab_off = off; q.slab_b = q.slab_c = ~0u;
                q.co_base = (short)(ct * 32); q.ci_base = (short)(ck * 32);
                q.cout = c.cout; q.cin_real = c.cin_real; q.scale = c.scale; q.want_bias = (short)want_bias; q.c_bias = 0;
                // parts: (x_hi, g_hi); RESR_F16X2 adds (x_hi, g_lo) and (x_lo, g_hi) -- or the MX job for both: product_parts
                int parts[4];
                const int np = product_parts(c, dtype, nparts, ck, parts);
                if (nj + np > kMaxJobs) return fail(RESR_ERR_ARG, "wgrad: more than %d jobs in one batch", kMaxJobs);
                for (int pi = 0; pi < np; ++pi) {
                    const int part = parts[pi];
                    WgradJob& j = a.jobs[nj++];
                    j.x = xh + (part == 2 ? (size_t)c.x_lo_off * es : part == 3 ? (size_t)c.x_q_off * es : 0);
                    j.g = gh + (part == 1 ? (size_t)c.g_lo_off * es : part == 3 ? (size_t)c.g_q_off * es : 0);
                    j.xstride_b = (unsigned)(c.in0_stride * es);
                    j.gstride_b = (unsigned)(c.g_stride * es);
                    j.slab_off = off;
                    j.mx = part == 3 ? 1u : 0u;
                    any_mx = any_mx || part == 3;
                    j.want_bias = (part < 2 || part == 3) ? want_bias : 0;   // (an MX job sums the bf8 g_lo bytes of its G fragments)
                    if (part == 3 && want_bias) q.c_bias = 1;
                    // (x_hi chunk 0, g_lo) carries the bias sum AND a real term of dW: skipping its taps was measured (+1 % on the exact16 step)
                    // and rejected -- the worst gradient tensor against the all-pairs plan rises by a fifth (32 x 64^2: 5.1e-4 -> 6.3e-4)
                    if (part == 1 && c.g_lo_bias_only && getenv("RESR_WGRAD_BIAS_JOBS_NO_TAPS")) j.want_bias |= 2u;   // experiment knob, read per call
                    j.xsub = (c.x_s2d_c > 0 && dtype != RESR_F32) ? (unsigned)((ck * 32) / c.x_s2d_c) : 4u;
                    // (the reduction takes dW = A + (B + C) 2^-12 with the bias from A and B -- and from C where C is an MX job that summed g_lo: c_bias)
                    if (part == 1) q.slab_b = off;
                    if (part == 2 || part == 3) q.slab_c = off;
                    off += (unsigned)((part == 3 ? splits_mx : splits) * kSlab);
                }
            }
    }
    a.partial = partial; r.partial = partial; r.splits = splits;
    r.splits_c = any_mx ? splits_mx : 0;
    a.splits_mx = splits_mx;
    r.unscale = convs[0].unscale;
    for (int i = 1; i < nconv; ++i)
        if (convs[i].unscale != convs[0].unscale) return fail(RESR_ERR_ARG, "wgrad: one pre-scale per launch");
    {   // 32-bit lane offsets are enough when every operand stays below 2^24 pixels and 4 GB
        const size_t px = (size_t)n * h * w;
        size_t smax = 0;
        for (int i = 0; i < nj; ++i) smax = smax > a.jobs[i].xstride_b ? smax : a.jobs[i].xstride_b, smax = smax > a.jobs[i].gstride_b ? smax : a.jobs[i].gstride_b;
        a.fast_addr = (px <= (1u << 24) && smax < (1u << 24) && px * smax + 64 < (1ull << 32)) ? 1 : 0;
        static const char* env = getenv("RESR_WGRAD_GENERIC_ADDR");  // test knob: force the 64-bit addressing path
        if (env) a.fast_addr = 0;
    }
    a.n = n; a.h = h; a.w_ = w; a.hs = up ? h / 2 : h; a.ws = up ? w / 2 : w; a.up = up ? 1 : 0; a.splits = splits;
    int rc;
    bool quad_done = false;
    static const char* pair_env = getenv("RESR_WGRAD_PAIR_KERNEL");   // test knob: keep the f16 pair kernel
    if (dtype != RESR_F32 && a.fast_addr && !pair_env) {
        rc = launch_wgrad_quad(a, nj, nr, stream, &quad_done);
        if (rc) return rc;
    }
    if (quad_done) rc = RESR_OK;
    else if (any_mx) return fail(RESR_ERR_ARG, "wgrad: MX jobs (x_q_off / g_q_off) need the quad kernel's preconditions (32-bit addressing, a quad plan of the products)");
    else if (dtype == RESR_F16 || dtype == RESR_F16X2) rc = launch_wgrad<half_t, 2>(a, nj, nr, stream);
    else if (dtype == RESR_F32) rc = launch_wgrad<float, 1>(a, nj, nr, stream);
    else return fail(RESR_ERR_ARG, "wgrad: dtype=%d", dtype);
    if (rc) return rc;
    if (splits <= 8 && !x2) hipLaunchKernelGGL(wgrad_reduce_wide_kernel, dim3(nr, (kSlab + 255) / 256), dim3(256), 0, stream, r);
    else hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nr, (kSlab + 31) / 32), dim3(256), 0, stream, r);
    RESR_CHECK_LAUNCH("wgrad_reduce_kernel");
    return RESR_OK;
}

// resr_debug_wgrad_dense_blocks (include/resr_debug.h): the batched launch pair the generator's backward pass issues per RRDB --
// the five convolutions of `nblocks` dense blocks (26 products each) in ONE wgrad_batch call -- on caller-made operands, so that
// tools/energy.py can hold exactly that launch in a loop.  x[b]: chunk-planar workspace [6][n,h,w,32] f16 (conv_k reads the
// prefix of 2 + (k - 1) planes), g[b]: chunk-planar gradients [6][n,h,w,32] -- planes 0,1 the closing convolution's 64
// channels, plane 1 + k conv_k's 32; dw: 26624 * 9 floats per block (every convolution's OIHW gradient back to back).
int wgrad_debug_dense_blocks(int nblocks, const void* const* x, const void* const* g, int n, int h, int w, int splits, float* partial,
                             size_t partial_bytes, float* dw, hipStream_t stream) {
    if (nblocks < 1 || nblocks > 3 || !x || !g || !partial || !dw) return fail(RESR_ERR_ARG, "wgrad_dense_blocks: bad argument");
    WgradConv wc[15];
    const long plane = (long)n * h * w * 32;
    int nc = 0;
    float* out = dw;
    for (int b = 0; b < nblocks; ++b)
        for (int k = 1; k <= 5; ++k) {
            WgradConv& c = wc[nc++];
            const int cin = 64 + 32 * (k - 1), cout = k < 5 ? 32 : 64;
            c.x0 = x[b]; c.cin = cin; c.in0_stride = 32; c.cin_real = cin;
            c.g = (const char*)g[b] + (k < 5 ? (size_t)(1 + k) * plane * 2 : 0); c.cout = cout; c.cout_pad = cout; c.g_stride = 32;
            c.x_chunk_stride = plane; c.g_chunk_stride = plane; c.x_lo_off = c.g_lo_off = 0; c.x_s2d_c = 0; c.g_lo_bias_only = 0;
            c.dw = out; c.db = nullptr; c.scale = 1.f;
            out += (size_t)cout * cin * 9;
        }
    if (wgrad_batch_partial_bytes(wc, nc, splits, RESR_F16) > partial_bytes) return fail(RESR_ERR_WORKSPACE, "wgrad_dense_blocks: slab buffer too small");
    return wgrad_batch(wc, nc, n, h, w, RESR_F16, 0, splits, partial, stream);
}

// single-conv C-ABI entry (include/resr.h resr_conv3x3_wgrad)
size_t wgrad_partial_bytes(const ResrWgradDesc* d) {
    return (size_t)(d->cin / 32) * (d->cout_pad / 32) * (d->dtype == RESR_F16X2 ? wgrad_x2_products() : 1) * d->splits * kSlab * sizeof(float);
}

int wgrad_dispatch(const ResrWgradDesc* d, const void* x0, const void* x1, const void* g, float* partial, float* dw,
                   float* db, hipStream_t stream) {
    if (!d || !x0 || !g || !partial || !dw) return fail(RESR_ERR_ARG, "wgrad: null argument");
    if (d->cin0 != d->cin || x1) return fail(RESR_ERR_ARG, "wgrad: two-segment X is not supported (cin0 must equal cin)");
    // (a caller built against the version-1 header passes a shorter struct: whatever lies behind it is read as chunk strides)
    {
        const int64_t px = (int64_t)d->n * d->h * d->w;
        auto bad = [&](int64_t cs) { return cs < 0 || (cs > 0 && (cs < 32 || cs > px * 32 * 64)); };
        if (bad(d->x_chunk_stride) || bad(d->g_chunk_stride) || d->x_lo_offset < 0 || d->g_lo_offset < 0)
            return fail(RESR_ERR_ARG, "wgrad: chunk strides %lld / %lld, lo offsets %lld / %lld out of range (zero-initialise ResrWgradDesc; resr_version() = %d)",
                        (long long)d->x_chunk_stride, (long long)d->g_chunk_stride, (long long)d->x_lo_offset, (long long)d->g_lo_offset, RESR_VERSION);
    }
    // RESR_F16X2: a single-f16 G is selected EXPLICITLY (RESR_CONV_OUT_SINGLE in flags: the tensor a pass with that flag wrote); a zero
    // g_lo_offset without it is a forgotten field, not a mode -- the internal WgradConv keeps the "0 = single" convention
    int flags = d->flags;
    long g_lo = (long)d->g_lo_offset;
    if (d->dtype == RESR_F16X2) {
        if (flags & RESR_CONV_OUT_SINGLE) g_lo = 0;
        else if (g_lo == 0) return fail(RESR_ERR_ARG, "wgrad: RESR_F16X2 needs g_lo_offset (or RESR_CONV_OUT_SINGLE in flags for a single f16 G)");
    }
    flags &= ~RESR_CONV_OUT_SINGLE;
    WgradConv c;
    c.x0 = x0; c.cin = d->cin; c.in0_stride = d->in0_stride; c.cin_real = d->cin_real;
    c.g = g; c.cout = d->cout; c.cout_pad = d->cout_pad; c.g_stride = d->g_stride;
    c.x_chunk_stride = (long)d->x_chunk_stride; c.g_chunk_stride = (long)d->g_chunk_stride;
    c.x_lo_off = (long)d->x_lo_offset; c.g_lo_off = g_lo; c.x_s2d_c = 0; c.g_lo_bias_only = 0;
    c.dw = dw; c.db = db; c.scale = d->scale;
    // more products than one launch's job table holds (or an output wider than 64 channels): the layer mode (f16, exact16)
    const int tap_products = (c.cin / 32) * (c.cout_pad / 32) * (d->dtype == RESR_F16X2 ? wgrad_x2_products() : 1);
    if ((d->dtype == RESR_F16 || d->dtype == RESR_F16X2) &&
        (c.cout_pad > 64 || (c.cin / 32) * (c.cout_pad / 32) > kMaxReduce || tap_products > kMaxJobs))
        return wgrad_layer(&c, d->n, d->h, d->w, d->dtype, flags, d->splits, partial, stream);
    return wgrad_batch(&c, 1, d->n, d->h, d->w, d->dtype, flags, d->splits, partial, stream);
}

}  // namespace resr
