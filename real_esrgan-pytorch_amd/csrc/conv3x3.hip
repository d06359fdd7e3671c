// conv3x3.hip -- 3x3 / stride 1 / pad 1 convolution as an implicit GEMM on the gfx950 matrix cores: the C-ABI dispatch
// and the one-role kernel (every wave stages and multiplies).  Strict (f32) mode runs on this kernel; fast (f16) mode
// runs on the producer/consumer kernel of conv3x3_ws.h and falls back to this one when its preconditions fail.
//
// Replaces the F.conv2d + LeakyReLU + torch.cat + mul/add call sites of the reference generator
// (model.py:87-98 dense block, :123-132 RRDB tail, :255-272 head/tail) and, with transposed
// weights, their autograd backward-data passes.
//
// GEMM view:  C[cout][pixel] = sum_{tap, cin} W[cout][cin][tap] * X[pixel + tap][cin]
//   A operand = packed weights (M = cout, 32 rows per MFMA tile), streamed from L2 in fragment order
//   B operand = input pixels   (N = 32 pixels of one image row), read from an LDS halo tile
//   K         = 9 taps x cin, walked as 32-channel chunks; "im2col" is only an LDS address offset.
// A workgroup (NW waves) owns a (NW*NT) x 32 pixel tile and all output channels; wave w owns rows
// [w*NT, w*NT+NT).  The (NW*NT+2) x 34 x 32ch halo tile of the next chunk is fetched while the
// current one is multiplied (double-buffered LDS, one barrier per chunk).
//
// f16 (fast) : v_mfma_f32_32x32x16_f16, fp32 accumulate.   f32 (strict): v_mfma_f32_32x32x2_f32.
// Both read 16 bytes per lane per operand per k-step, so one template serves both.
//
// LDS layout: pixel-major, PB = 32*sizeof(T) bytes per pixel, i.e. SPP = PB/16 sixteen-byte slots.
// ds_read_b128 is served in 16-lane groups whose pixel x-coordinates cover all residues mod 16;
// XOR-ing the slot index with a function of x makes the 16 lanes hit 16 distinct slots of the
// 256-byte bank row (conflict-free) -- see swz().
#include <stdlib.h>

#include <type_traits>

#include "conv3x3.h"

namespace resr {

template <typename T, int MT, int NT, int NW>
__global__ __launch_bounds__(64 * NW) void conv3x3_kernel(const ConvArgs a) {
    constexpr int NTHR = 64 * NW;            // NW waves per workgroup, each owning NT rows of the tile
    constexpr int E = 16 / (int)sizeof(T);  // elements per 16-byte slot
    constexpr int SPP = 32 / E;             // slots per pixel per chunk (4 f16, 8 f32)
    constexpr int KS = SPP / 2;             // k-steps per chunk
    constexpr int PB = 32 * (int)sizeof(T); // bytes per pixel per chunk
    constexpr int TH = NW * NT, TW = 32, HH = TH + 2, HW = TW + 2;
    constexpr int NSLOT = HH * HW * SPP;
    constexpr int NS = (NSLOT + NTHR - 1) / NTHR;
    constexpr int BUF = HH * HW * PB;
    constexpr int WTAP = KS * MT * 1024;    // packed weight bytes per (chunk, tap)

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tx = bid % a.tiles_x;
    const int t2 = bid / a.tiles_x;
    const int ty = t2 % a.tiles_y;
    const int n = t2 / a.tiles_y;
    const int x0 = tx * TW, y0 = ty * TH;
    const bool up = (a.flags & RESR_CONV_UPSAMPLE_IN) != 0;

    // ---- staging map: slot s = tid + i*NTHR  ->  halo pixel (hy,hx), 16-byte piece c16 ----------
    const int c16 = tid % SPP;  // NTHR % SPP == 0: identical for every i
    int pix[NS];                // source pixel index inside the image, -1 = zero (padding / unused)
    int loff[NS];               // swizzled LDS byte offset inside one buffer, -1 = no slot
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int s = tid + i * NTHR;
        const int hp = s / SPP;
        const int hy = hp / HW, hx = hp - hy * HW;
        const int iy = y0 + hy - 1, ix = x0 + hx - 1;
        const bool slot_ok = s < NSLOT;
        const bool in_img = slot_ok && iy >= 0 && iy < a.h && ix >= 0 && ix < a.w_;
        const int sy = up ? (iy >> 1) : iy, sx = up ? (ix >> 1) : ix;
        pix[i] = in_img ? sy * a.ws + sx : -1;
        loff[i] = slot_ok ? hp * PB + ((c16 ^ swz<SPP>(hx)) << 4) : -1;
    }
    const size_t img_px = (size_t)a.hs * a.ws;
    const char* in0 = a.in0 + (size_t)n * img_px * a.in0_stride_b;
    const char* in1 = a.in1 ? a.in1 + (size_t)n * img_px * a.in1_stride_b : nullptr;

    uint4 stg[NS];
    // branch-free: out-of-image / unused slots read pixel 0 of the image (always mapped) and are zeroed by a
    // select -- a branch per load would make the compiler wait for each load separately
    auto stage_load = [&](int ck) {
        const int c0 = ck * 32;
        const bool seg1 = c0 >= a.cin0;
        const char* base = seg1 ? in1 + (size_t)((c0 - a.cin0) >> 5) * a.in1_chunk_b : in0 + (size_t)(c0 >> 5) * a.in0_chunk_b;
        const unsigned stride_b = seg1 ? a.in1_stride_b : a.in0_stride_b;
        const unsigned ch_b = c16 << 4;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const unsigned pi = pix[i] >= 0 ? (unsigned)pix[i] : 0u;
            stg[i] = *reinterpret_cast<const uint4*>(base + (size_t)pi * stride_b + ch_b);
        }
    };
    auto stage_store = [&](int buf) {
        char* dst = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < NS; ++i)
            if (loff[i] >= 0) *reinterpret_cast<uint4*>(dst + loff[i]) = pix[i] >= 0 ? stg[i] : make_uint4(0, 0, 0, 0);
    };

    // ---- per-lane operand addressing -----------------------------------------------------------
    const int lx = lane & 31, kh = lane >> 5;
    int colb[3], colsw[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int hx = lx + dx;
        colb[dx] = hx * PB;
        colsw[dx] = swz<SPP>(hx);
    }
    const int row0 = wave * NT;
    const char* wp = a.w + (lane << 4);
    // per-lane LDS byte offsets of the B fragment for (dx, k-step); rows / taps add compile-time immediates
    int boff[3][KS];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) boff[dx][ks] = row0 * (HW * PB) + colb[dx] + (((ks * 2 + kh) ^ colsw[dx]) << 4);

    float16v acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;

    const int nchunks = a.cin >> 5;

    // Weight fragments: 3-slot register ring, fetched two taps ahead of use (L2 latency ~ 2 taps of MFMA
    // work per wave).  9 taps per chunk and 3 slots => the slot index stays static across the chunk loop.
    // sched_barrier pins the loads where they are written; otherwise the scheduler sinks them to their use.
    uint4 wr[3][KS * MT];
    auto wload = [&](int slot, int q) {
        const char* wq = wp + (size_t)q * WTAP;
#pragma unroll
        for (int j = 0; j < KS * MT; ++j) wr[slot][j] = *reinterpret_cast<const uint4*>(wq + j * 1024);
    };
    wload(0, 0);
    wload(1, 1);

    stage_load(0);
    stage_store(0);
    __syncthreads();

    // The last chunk is peeled (compile-time `more`): with a run-time flag the compiler's wait-count merge of
    // the two paths makes every chunk wait for its own staging loads before the first MFMA.
    auto chunk_body = [&](int ck, auto more_tag) {
        constexpr bool more = decltype(more_tag)::value;
        if (more) stage_load(ck + 1);
        const char* lbuf = smem + (ck & 1) * BUF;
        // k-steps of the chunk, software-pipelined: the B fragments of step s+1 are read from LDS while the
        // MFMAs of step s run (ping-pong registers; 9*KS is even, so the parity is static across chunks)
        uint4 bb[2][NT];
        auto bload = [&](int slot, int sidx) {
            const int tap = sidx / KS, ks = sidx % KS;
            const int dy = tap / 3, dx = tap % 3;
            const char* bp = lbuf + boff[dx][ks];
#pragma unroll
            for (int t = 0; t < NT; ++t) bb[slot][t] = *reinterpret_cast<const uint4*>(bp + (t + dy) * (HW * PB));
        };
        bload(0, 0);
#pragma unroll
        for (int sidx = 0; sidx < 9 * KS; ++sidx) {
            const int tap = sidx / KS, ks = sidx % KS;
            // weights two taps ahead; the tail over-reads stay inside the packed buffer's slack
            if (ks == 0) wload((tap + 2) % 3, ck * 9 + tap + 2);
            if (sidx + 1 < 9 * KS) bload((sidx + 1) & 1, sidx + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[m][t] = Frag<T>::mma(wr[tap % 3][ks * MT + m], bb[sidx & 1][t], acc[m][t]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) stage_store((ck + 1) & 1);
        __syncthreads();
    };
    for (int ck = 0; ck + 1 < nchunks; ++ck) chunk_body(ck, std::true_type{});
    chunk_body(nchunks - 1, std::false_type{});

    // ---- epilogue: lane owns pixel (row0+t, lx) and 4 consecutive couts per accumulator quad ----
    const bool f_lrelu = a.flags & RESR_CONV_LRELU, f_clamp = a.flags & RESR_CONV_CLAMP01;
    const bool f_nchw = a.flags & RESR_CONV_OUT_NCHW_F32, f_mask = a.flags & RESR_CONV_MASK;
    const bool f_bias = !(a.flags & RESR_CONV_NO_BIAS) && a.bias != nullptr;
    const bool f_aux_mask = (a.flags & RESR_CONV_AUX_BEFORE_MASK) && a.aux && !f_nchw;
    const bool f_aux_res = (a.flags & RESR_CONV_AUX_BEFORE_RES) && a.aux && !f_nchw;
    const bool f_sbits = (a.flags & RESR_CONV_WRITE_SIGNBITS) != 0, f_mbits = (a.flags & RESR_CONV_MASK_BITS) != 0;
    const int x = x0 + lx;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int y = y0 + row0 + t;
        if (y >= a.h || x >= a.w_) continue;
        const size_t p = ((size_t)n * a.h + y) * a.w_ + x;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const size_t wi = p * (size_t)((a.cout + 31) >> 5) + m;   // sign word of (pixel, chunk m)
            const unsigned mbits = f_mbits ? reinterpret_cast<const unsigned*>(a.mask)[wi] : 0u;
            unsigned sbits = 0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cq = g * 8 + kh * 4, co = m * 32 + cq;  // cq: channel inside the 32-channel chunk m
                if (co >= a.cout) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[m][t][g * 4 + r];
                if (f_bias) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (co + r < a.cout) v[r] += a.bias[co + r];
                }
                if (f_aux_mask) store4<T>(reinterpret_cast<char*>(a.aux), p * a.out_stride + (size_t)m * a.out_chunk + cq, v);
                if (f_mbits) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= ((mbits >> (cq + r)) & 1u) ? 1.f : a.slope;
                } else if (f_mask) {
                    float mk[4];
                    load4<T>(a.mask, p * a.mask_stride + (size_t)m * a.mask_chunk + cq, mk);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= (mk[r] > 0.f ? 1.f : a.slope);
                }
                if (f_lrelu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : v[r] * a.slope;
                }
                if (f_aux_res) store4<T>(reinterpret_cast<char*>(a.aux), p * a.out_stride + (size_t)m * a.out_chunk + cq, v);
                if (a.res0) {
                    float rr[4];
                    load4<T>(a.res0, p * a.res0_stride + (size_t)m * a.res0_chunk + cq, rr);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(v[r], a.s0, a.t0 * rr[r]);   // explicit: one rounding, the same in every instantiation
                }
                if (a.res1) {
                    float rr[4];
                    load4<T>(a.res1, p * a.res1_stride + (size_t)m * a.res1_chunk + cq, rr);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(v[r], a.s1, a.t1 * rr[r]);
                }
                if (f_nchw) {
                    float* o = reinterpret_cast<float*>(a.out);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (co + r >= a.cout) continue;
                        const size_t q = (((size_t)n * a.cout + co + r) * a.h + y) * a.w_ + x;
                        float u = v[r];
                        if (f_clamp) {
                            if (a.aux) a.aux[q] = (u >= 0.f && u <= 1.f) ? 1 : 0;
                            u = u < 0.f ? 0.f : (u > 1.f ? 1.f : u);   // torch.clamp_ semantics: a NaN stays a NaN (fminf / fmaxf would drop it)
                        }
                        o[q] = u;
                    }
                } else {
                    if (f_clamp) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = v[r] < 0.f ? 0.f : (v[r] > 1.f ? 1.f : v[r]);
                    }
                    store4<T>(a.out, p * a.out_stride + (size_t)m * a.out_chunk + cq, v);
                    if (f_sbits) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) sbits |= ((float)(T)v[r] > 0.f ? 1u : 0u) << (cq + r);
                    }
                }
            }
            if (f_sbits) {   // lanes lx and lx+32 hold complementary nibbles of the pixel's word: merge, lane kh = 0 stores
                const unsigned word = sbits | (unsigned)__shfl_xor((int)sbits, 32);
                if (kh == 0 && m * 32 < a.cout) reinterpret_cast<unsigned*>(a.aux)[wi] = word;
            }
        }
    }
}

template <typename T, int MT, int NT, int NW = 4>
static int launch_conv(const ConvArgs& a, hipStream_t stream) {
    constexpr int PB = 32 * (int)sizeof(T);
    constexpr int TH = NW * NT;
    constexpr int BUF = (TH + 2) * 34 * PB;
    ConvArgs args = a;
    args.tiles_x = (a.w_ + 31) / 32;
    args.tiles_y = (a.h + TH - 1) / TH;
    const size_t lds = 2 * BUF;
    static bool attr_done = false;  // benign race: idempotent
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_kernel<T, MT, NT, NW>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    const unsigned grid = (unsigned)(args.tiles_x * args.tiles_y * a.n);
    prof_before(stream);
    hipLaunchKernelGGL((conv3x3_kernel<T, MT, NT, NW>), dim3(grid), dim3(64 * NW), lds, stream, args);
    prof_after(stream, (sizeof(T) == 2 ? 0 : 10000) + MT * 100 + NT * 10 + NW,
               2.0 * 9 * a.cin * a.cout * (double)a.n * a.h * a.w_, conv_algorithmic_bytes(a, sizeof(T)));
    RESR_CHECK_LAUNCH("conv3x3_kernel");
    return RESR_OK;
}

int conv3x3_ws_f16(const ConvArgs& a, int mt, bool x2, hipStream_t stream);  // conv3x3_ws.hip
bool conv3x3_ws_supported(const ConvArgs& a);
bool conv3x3_chain_device_ok();

int conv3x3_ws_chain_f16(const ConvArgs& a, const ChainJob* jobs, int njobs, const double* flop, const double* bytes, bool x2, void* state, size_t state_bytes,
                         hipStream_t stream);   // conv3x3_ws.hip (1 = the state buffer does not cover this geometry)
static int conv3x3_route(const ResrConvDesc* d, ConvArgs& a, bool have_bias, bool have_in1, hipStream_t stream);

// descriptor + pointers -> kernel arguments (validation included)
static int conv3x3_args(const ResrConvDesc* d, const void* in0, const void* in1, const void* w,
                        const float* bias, const void* res0, const void* res1, const void* mask,
                        void* out, void* aux, ConvArgs& a) {
    if (!d || !in0 || !w || !out) return fail(RESR_ERR_ARG, "conv3x3: null argument");
    if (d->cin <= 0 || (d->cin & 31) || (d->cin0 & 31) || d->cin0 <= 0 || d->cin0 > d->cin)
        return fail(RESR_ERR_ARG, "conv3x3: cin=%d cin0=%d must be positive multiples of 32", d->cin, d->cin0);
    if (d->cin0 < d->cin && !in1) return fail(RESR_ERR_ARG, "conv3x3: in1 missing for cin0 < cin");
    if (d->cout_pad != 32 && d->cout_pad != 64) return fail(RESR_ERR_ARG, "conv3x3: cout_pad=%d", d->cout_pad);
    if (d->cout <= 0 || d->cout > d->cout_pad) return fail(RESR_ERR_ARG, "conv3x3: cout=%d", d->cout);
    if (!(d->flags & RESR_CONV_OUT_NCHW_F32) && (d->cout & 3))
        return fail(RESR_ERR_ARG, "conv3x3: NHWC output needs cout %% 4 == 0");
    if (d->n <= 0 || d->h <= 0 || d->w <= 0) return fail(RESR_ERR_ARG, "conv3x3: empty shape");
    if ((d->flags & RESR_CONV_UPSAMPLE_IN) && ((d->h | d->w) & 1))
        return fail(RESR_ERR_ARG, "conv3x3: upsampled input needs even h,w");
    if ((d->flags & RESR_CONV_MASK) && !mask) return fail(RESR_ERR_ARG, "conv3x3: mask missing");
    if (d->flags & RESR_CONV_WRITE_SIGNBITS) {
        if (!aux || (d->cout & 7) || res0 || res1 ||
            (d->flags & (RESR_CONV_MASK | RESR_CONV_OUT_NCHW_F32 | RESR_CONV_CLAMP01 | RESR_CONV_AUX_BEFORE_MASK | RESR_CONV_AUX_BEFORE_RES)))
            return fail(RESR_ERR_ARG, "conv3x3: WRITE_SIGNBITS needs aux_out, cout %% 8 == 0 and a plain (bias/LeakyReLU) epilogue");
    }
    if (d->flags & RESR_CONV_MASK_BITS) {
        if (!(d->flags & RESR_CONV_MASK) || (d->cout & 7) || res0 || res1 || aux ||
            (d->flags & (RESR_CONV_OUT_NCHW_F32 | RESR_CONV_CLAMP01 | RESR_CONV_WRITE_SIGNBITS)))
            return fail(RESR_ERR_ARG, "conv3x3: MASK_BITS goes with MASK, cout %% 8 == 0, no residuals / aux");
    }
    const size_t es = elem_size(d->dtype);
    memset(&a, 0, sizeof(a));
    a.in0 = (const char*)in0; a.in1 = (const char*)in1; a.w = (const char*)w; a.bias = bias;
    a.res0 = (const char*)res0; a.res1 = (const char*)res1; a.mask = (const char*)mask;
    a.out = (char*)out; a.aux = (uint8_t*)aux;
    a.n = d->n; a.h = d->h; a.w_ = d->w;
    const bool up = d->flags & RESR_CONV_UPSAMPLE_IN;
    a.hs = up ? d->h / 2 : d->h; a.ws = up ? d->w / 2 : d->w;
    a.cin = d->cin; a.cin0 = d->cin0;
    a.in0_stride_b = (int)(d->in0_stride * es); a.in1_stride_b = (int)(d->in1_stride * es);
    a.cout = d->cout; a.out_stride = d->out_stride;
    a.res0_stride = d->res0_stride; a.res1_stride = d->res1_stride; a.mask_stride = d->mask_stride;
    auto chunk = [](int32_t v) { return v > 0 ? v : 32; };
    a.in0_chunk_b = (size_t)chunk(d->in0_chunk_stride) * es; a.in1_chunk_b = (size_t)chunk(d->in1_chunk_stride) * es;
    a.out_chunk = chunk(d->out_chunk_stride); a.res0_chunk = chunk(d->res0_chunk_stride);
    a.res1_chunk = chunk(d->res1_chunk_stride); a.mask_chunk = chunk(d->mask_chunk_stride);
    a.flags = d->flags & ~(RESR_CONV_OUT_SINGLE | RESR_CONV_SINGLE_W16 | RESR_CONV_MX_PAIRS); a.s0 = d->s0; a.t0 = d->t0; a.s1 = d->s1; a.t1 = d->t1; a.slope = d->slope;
    a.s2d_c = 0; a.tap_c = 0; a.ngroups = 1; a.w_group_b = 0;
    // RESR_F16X2: leading pair chunks / a single-f16 output (kept out of a.flags: the kernels and the chain checks never see the bit)
    a.pair_chunks = d->cin / 32; a.out_single = 0;
    a.mask_lo = (d->dtype == RESR_F16X2 && mask && (d->flags & RESR_CONV_MASK) && !(d->flags & RESR_CONV_MASK_BITS)) ? (long)d->mask_lo_offset : 0L;
    if (a.mask_lo < 0) return fail(RESR_ERR_ARG, "conv3x3: mask_lo_offset=%ld", a.mask_lo);
    if (d->dtype == RESR_F16X2) {
        if (d->x2_pair_chunks < 0 || d->x2_pair_chunks > d->cin / 32)
            return fail(RESR_ERR_ARG, "conv3x3: x2_pair_chunks=%d of %d chunks", d->x2_pair_chunks, d->cin / 32);
        if (d->x2_pair_chunks > 0) a.pair_chunks = d->x2_pair_chunks;
        if ((d->flags & RESR_CONV_SINGLE_W16) && a.pair_chunks * 32 < d->cin) a.single_stages = 1;
        if (d->flags & RESR_CONV_OUT_SINGLE) {
            if (d->flags & RESR_CONV_OUT_NCHW_F32) return fail(RESR_ERR_ARG, "conv3x3: OUT_SINGLE is an NHWC f16 output");
            a.out_single = 1;
        }
        // MX stages for the pair chunks (RESR_CONV_MX_PAIRS) and / or a q tensor next to a pair output
        if (d->in0_q_offset < 0 || d->in1_q_offset < 0 || d->out_q_offset < 0 || d->w_mx_offset < 0)
            return fail(RESR_ERR_ARG, "conv3x3: negative q / MX offset");
        if (d->flags & RESR_CONV_MX_PAIRS) {
            const bool in1_pairs = in1 && d->cin0 < d->cin && a.pair_chunks * 32 > d->cin0;
            if (d->in0_q_offset == 0 || (in1_pairs && d->in1_q_offset == 0) || d->w_mx_offset == 0)
                return fail(RESR_ERR_ARG, "conv3x3: RESR_CONV_MX_PAIRS needs the q offset of every pair operand and w_mx_offset");
            if (d->cout_groups > 1 || d->s2d_in_channels > 0 || d->s2d_out_channels > 0)
                return fail(RESR_ERR_ARG, "conv3x3: RESR_CONV_MX_PAIRS is a dense 3x3 pass of one output group");
            a.mx = 1;
            a.in0_q_b = (size_t)d->in0_q_offset * 2; a.in1_q_b = (size_t)d->in1_q_offset * 2;
            a.w_mx = (const char*)w + d->w_mx_offset;
        }
        if (d->out_q_offset != 0) {
            if (a.out_single || (d->flags & RESR_CONV_OUT_NCHW_F32) || d->cout_groups > 1)
                return fail(RESR_ERR_ARG, "conv3x3: out_q_offset goes with a pair NHWC output of one output group");
            a.out_q = (long)d->out_q_offset;
        }
    } else if (d->flags & RESR_CONV_MX_PAIRS) {
        return fail(RESR_ERR_ARG, "conv3x3: RESR_CONV_MX_PAIRS is an RESR_F16X2 mode");
    }
    const int groups = d->cout_groups > 1 ? d->cout_groups : 1;
    if (groups > 1) {
        const bool biased = bias && !(d->flags & RESR_CONV_NO_BIAS);
        if ((d->dtype != RESR_F16 && d->dtype != RESR_F16X2) || d->cout != 64 || d->cout_pad != 64 || in1 || (biased && groups > kMaxBiasGroups) ||
            (d->flags & (RESR_CONV_OUT_NCHW_F32 | RESR_CONV_WRITE_SIGNBITS | RESR_CONV_MASK_BITS | RESR_CONV_CLAMP01)))
            return fail(RESR_ERR_ARG, "conv3x3: cout_groups > 1 needs f16 / f16x2, cout = cout_pad = 64 per group, NHWC output, no sign-bit tensors, and at most %d groups with a bias", kMaxBiasGroups);
        a.ngroups = groups;
        a.w_group_b = (size_t)(d->cin / 32) * 9 * 2 * 1024 * es * (d->dtype == RESR_F16X2 ? 3 : 1);
    }
    if (d->s2d_in_channels > 0) {
        if ((d->s2d_in_channels & 31) || d->cin != 4 * d->s2d_in_channels)
            return fail(RESR_ERR_ARG, "conv3x3: s2d_in_channels=%d must be a multiple of 32 with cin = 4 * s2d_in_channels", d->s2d_in_channels);
        a.s2d_c = d->s2d_in_channels;
    }
    if (d->s2d_out_channels > 0) {
        if ((d->s2d_out_channels & 63) || d->s2d_in_channels > 0 || groups * 64 != 4 * d->s2d_out_channels)
            return fail(RESR_ERR_ARG, "conv3x3: s2d_out_channels=%d needs cout_groups * 64 = 4 * s2d_out_channels", d->s2d_out_channels);
        a.tap_c = d->s2d_out_channels;
    }
    return RESR_OK;
}

int conv3x3_dispatch(const ResrConvDesc* d, const void* in0, const void* in1, const void* w,
                     const float* bias, const void* res0, const void* res1, const void* mask,
                     void* out, void* aux, hipStream_t stream) {
    ConvArgs a;
    const int rc = conv3x3_args(d, in0, in1, w, bias, res0, res1, mask, out, aux, a);
    if (rc) return rc;
    return conv3x3_route(d, a, bias != nullptr, in1 != nullptr, stream);
}

static int conv3x3_route(const ResrConvDesc* d, ConvArgs& a, bool have_bias, bool have_in1, hipStream_t stream) {
    const size_t es = elem_size(d->dtype);
    const void* in1 = have_in1 ? (const void*)a.in1 : nullptr;
    const void* res0 = a.res0;
    const void* res1 = a.res1;
    const int groups = d->cout_groups > 1 ? d->cout_groups : 1;
    (void)have_bias;
    const int mt = d->cout_pad / 32;
    if (d->dtype == RESR_F16X2) {
        // hi/lo pairs: only the producer/consumer kernel has the mode (an aux tensor of AUX_BEFORE_* has out's shape and out's
        // hi -> lo offset)
        const bool nchw = d->flags & RESR_CONV_OUT_NCHW_F32;
        const bool in1_pairs = in1 && d->cin0 < d->cin && a.pair_chunks * 32 > d->cin0;   // does the second segment hold pair chunks?
        if (d->in0_lo_offset == 0 || (in1_pairs && d->in1_lo_offset == 0) || (!nchw && !a.out_single && d->out_lo_offset == 0) ||
            (res0 && d->res0_lo_offset == 0) || (res1 && d->res1_lo_offset == 0))
            return fail(RESR_ERR_ARG, "conv3x3: RESR_F16X2 needs the hi -> lo offset of every pair operand");
        a.in0_lo_b = (size_t)d->in0_lo_offset * es; a.in1_lo_b = (size_t)d->in1_lo_offset * es;
        a.out_lo = a.out_single ? 0L : (long)d->out_lo_offset; a.res0_lo = (long)d->res0_lo_offset; a.res1_lo = (long)d->res1_lo_offset;
        if (!conv3x3_ws_supported(a))
            return fail(RESR_ERR_ARG, "conv3x3: RESR_F16X2 needs tensors below 4 GB / 2^24 pixels and cout %% 8 == 0");
        if (a.mx && (a.s2d_c > 0 || a.tap_c > 0)) return fail(RESR_ERR_ARG, "conv3x3: RESR_CONV_MX_PAIRS with sparse taps");
        return conv3x3_ws_f16(a, mt, true, stream);
    }
    // 16-row tiles only when that still yields enough workgroups to fill 256 CUs twice over
    const long tiles4 = (long)((d->w + 31) / 32) * ((d->h + 15) / 16) * d->n;
    const bool big = tiles4 >= 512;
    if (d->dtype == RESR_F16) {
        static const char* old_env = getenv("RESR_CONV_ONE_ROLE");  // test knob: fast mode on the one-role kernel below
        if (!old_env && conv3x3_ws_supported(a)) return conv3x3_ws_f16(a, mt, false, stream);
        if (groups > 1) return fail(RESR_ERR_ARG, "conv3x3: cout_groups > 1 needs the producer/consumer kernel's preconditions");
        // measured on MI355X (B=8, 256^2): cout 32 -> 8 waves x 2 rows (4 waves/SIMD, 2 workgroups/CU) beats
        // 4 waves x 4 rows by 5-16 %; cout 64 -> 4 waves x 2 rows (2 waves/SIMD) beats every 8-wave shape
        if (mt == 1) return big ? launch_conv<half_t, 1, 2, 8>(a, stream) : launch_conv<half_t, 1, 2, 4>(a, stream);
        return launch_conv<half_t, 2, 2, 4>(a, stream);
    } else if (d->dtype == RESR_F32) {
        if (mt == 1) return launch_conv<float, 1, 2>(a, stream);
        return launch_conv<float, 2, 2>(a, stream);
    }
    return fail(RESR_ERR_ARG, "conv3x3: dtype=%d", d->dtype);
}

// The passes of one dense block (forward conv1..conv4 [+ conv5], model.py:90-96, or the mirrored backward-data passes) as
// ONE persistent launch when the fast-mode kernel can chain them (conv3x3_ws.h, CH); otherwise one launch each.
// Job j < njobs: descriptor d[j], weights w[j], bias[j] (or null), mask[j] (or null), output out[j], aux[j] (or null); the
// shared inputs in0 / in1 hold the plane prefix every job reads.  d5 (optional): the block's closing cout-64 convolution
// over the same inputs with its residuals; on launches of at most two tiles per CU (the 64^2 training crops) it joins the
// chain as two cout-32 jobs that read the cout-64 weight packing in place -- there a launch's fill / drain costs more than
// the cout-64 shape's better register tile gains; at full size it stays its own launch.
int conv3x3_block_dispatch(int njobs, const ResrConvDesc* d, const void* in0, const void* in1, const void* const* w,
                           const float* const* bias, const void* const* mask, void* const* out, void* const* aux,
                           const ResrConvDesc* d5, const void* w5, const float* bias5, const void* res0_5, const void* res1_5,
                           void* out5, void* chain_state, size_t chain_state_bytes, hipStream_t stream) {
    if (njobs <= 0 || njobs > 4 || !d) return fail(RESR_ERR_ARG, "conv3x3_chain: njobs=%d", njobs);
    ConvArgs a[4], a5;
    for (int j = 0; j < njobs; ++j) {
        const int rc = conv3x3_args(&d[j], in0, in1, w[j], bias ? bias[j] : nullptr, nullptr, nullptr, mask ? mask[j] : nullptr,
                                    out[j], aux ? aux[j] : nullptr, a[j]);
        if (rc) return rc;
    }
    if (d5) {
        const int rc = conv3x3_args(d5, in0, in1, w5, bias5, res0_5, res1_5, nullptr, out5, nullptr, a5);
        if (rc) return rc;
    }
    const char* no_chain = getenv("RESR_CONV_NO_CHAIN");   // test / A-B knob: one launch per job (read per call, so a test can flip it)
    const bool x2 = d[0].dtype == RESR_F16X2;
    bool ok = !no_chain && chain_state && njobs >= 2 && (d[0].dtype == RESR_F16 || x2) && conv3x3_chain_device_ok();
    if (ok && x2) {   // hi -> lo offsets as conv3x3_route sets them for single launches
        const size_t es2 = 2;
        for (int j = 0; j < njobs; ++j) {
            const bool in1_pairs = in1 && d[j].cin0 < d[j].cin && a[j].pair_chunks * 32 > d[j].cin0;
            if (d[j].dtype != RESR_F16X2 || d[j].in0_lo_offset == 0 || (d[j].out_lo_offset == 0 && !a[j].out_single) || (in1_pairs && d[j].in1_lo_offset == 0)) { ok = false; break; }
            a[j].in0_lo_b = (size_t)d[j].in0_lo_offset * es2; a[j].in1_lo_b = (size_t)d[j].in1_lo_offset * es2;
            a[j].out_lo = a[j].out_single ? 0L : (long)d[j].out_lo_offset;
            if (a[j].mx != a[0].mx || a[j].in0_q_b != a[0].in0_q_b || a[j].in1_q_b != a[0].in1_q_b) { ok = false; break; }   // one stage map per chain
        }
        if (ok && d5) {
            if (d5->dtype != RESR_F16X2 || d5->in0_lo_offset == 0 || d5->out_lo_offset == 0 || (res0_5 && d5->res0_lo_offset == 0) ||
                (res1_5 && d5->res1_lo_offset == 0)) ok = false;
            a5.in0_lo_b = (size_t)d5->in0_lo_offset * es2; a5.in1_lo_b = (size_t)d5->in1_lo_offset * es2;
            a5.out_lo = (long)d5->out_lo_offset; a5.res0_lo = (long)d5->res0_lo_offset; a5.res1_lo = (long)d5->res1_lo_offset;
        }
    }
    const int fwd_flags = RESR_CONV_LRELU | RESR_CONV_WRITE_SIGNBITS, inf_flags = RESR_CONV_LRELU;
    const int bwd_flags = RESR_CONV_MASK | RESR_CONV_MASK_BITS | RESR_CONV_NO_BIAS;
    const ConvArgs& b = a[njobs - 1];   // the widest growth job: its in0 / in1 split describes every prefix
    // exact16: one "leading pair chunks" count for the whole chain (kAllPairs: every chunk of every job is a pair)
    constexpr int kAllPairs = 1 << 20;
    auto eff_pairs = [&](const ConvArgs& c) { return c.pair_chunks * 32 < c.cin ? c.pair_chunks : kAllPairs; };
    const int chain_pairs = eff_pairs(b);
    for (int j = 0; ok && j < njobs; ++j) {
        const ConvArgs& c = a[j];
        ok = d[j].dtype == d[0].dtype && (!x2 || eff_pairs(c) == chain_pairs || c.cin <= chain_pairs * 32) && c.out_single == b.out_single && (c.single_stages == b.single_stages || c.pair_chunks * 32 >= c.cin) && d[j].cout_pad == 32 && c.cout == 32 && c.in0_lo_b == b.in0_lo_b && c.in1_lo_b == b.in1_lo_b && (c.flags == fwd_flags || c.flags == bwd_flags || c.flags == inf_flags) &&
             c.flags == b.flags && c.n == b.n && c.h == b.h && c.w_ == b.w_ && c.hs == c.h && c.ws == c.w_ &&
             (c.n % 8) == 0 && (c.w_ % 2) == 0 && c.slope == b.slope &&
             c.in0_stride_b == 64 && c.in0_chunk_b == b.in0_chunk_b && c.out_stride == 32 &&
             (c.cin0 == c.cin || (c.cin0 == b.cin0 && c.in1_stride_b == 64 && c.in1_chunk_b == b.in1_chunk_b)) &&
             c.cin >= (j == 0 ? 64 : 96) && c.ngroups == 1 && !c.s2d_c && !c.tap_c &&
             (c.flags == fwd_flags ? c.aux != nullptr : c.flags == bwd_flags ? c.mask != nullptr : (!c.aux && !c.mask)) &&
             (c.slope >= 0.f && c.slope <= 1.f) && !c.res0 && !c.res1 && conv3x3_ws_supported(c);
        if (ok && j + 1 < njobs) {   // this job's output plane is the next job's last input chunk
            const ConvArgs& nx = a[j + 1];
            const int c0 = nx.cin - 32;
            const bool seg1 = c0 >= nx.cin0;
            const char* plane = seg1 ? nx.in1 + (size_t)((c0 - nx.cin0) >> 5) * nx.in1_chunk_b : nx.in0 + (size_t)(c0 >> 5) * nx.in0_chunk_b;
            ok = nx.cin == c.cin + 32 && plane == c.out;
        }
    }
    if (ok) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) ok = false;   // the flag epoch is a launch argument
    }
    // the closing convolution as jobs 4 and 5 of the chain
    bool with5 = ok && d5 && njobs == 4;
    if (with5) {
        const char* knob = getenv("RESR_CHAIN_CONV5");   // A/B knob: 0 never, 1 whenever possible; default by launch size
        const long tiles16 = (long)((b.w_ + 31) / 32) * ((b.h + 15) / 16) * b.n;
        with5 = knob ? knob[0] == '1' : tiles16 <= 512;
        const ConvArgs& c = a5;
        const int c0 = c.cin - 32;
        const bool seg1 = c0 >= c.cin0;
        const char* last = seg1 ? c.in1 + (size_t)((c0 - c.cin0) >> 5) * c.in1_chunk_b : c.in0 + (size_t)(c0 >> 5) * c.in0_chunk_b;
        with5 = with5 && d5->dtype == d[0].dtype && c.mx == b.mx && c.in0_q_b == b.in0_q_b && c.in1_q_b == b.in1_q_b && (!x2 || eff_pairs(c) == chain_pairs) && !c.out_single && c.single_stages == b.single_stages && c.in0_lo_b == b.in0_lo_b && c.in1_lo_b == b.in1_lo_b && d5->cout_pad == 64 && c.cout == 64 && c.cin == b.cin + 32 && last == b.out &&
                (c.flags & ~RESR_CONV_NO_BIAS) == 0 && ((c.flags & RESR_CONV_NO_BIAS) != 0) == ((b.flags & RESR_CONV_NO_BIAS) != 0) &&
                c.n == b.n && c.h == b.h && c.w_ == b.w_ && c.hs == c.h && c.ws == c.w_ &&
                c.in0_stride_b == 64 && c.in0_chunk_b == b.in0_chunk_b &&
                (c.cin0 == c.cin || (c.cin0 == b.cin0 && c.in1_stride_b == 64 && c.in1_chunk_b == b.in1_chunk_b)) &&
                c.out_stride == 32 && c.out_chunk > 32 && c.res0 && c.res0_stride == 32 && c.res0_chunk > 32 &&
                (!c.res1 || (c.res1_stride == 32 && c.res1_chunk > 32)) && !c.mask && !c.aux && c.ngroups == 1 && !c.s2d_c && !c.tap_c &&
                conv3x3_ws_supported(c);
    }
    if (!ok) {
        for (int j = 0; j < njobs; ++j) {
            const int rc = conv3x3_route(&d[j], a[j], a[j].bias != nullptr, a[j].in1 != nullptr, stream);
            if (rc) return rc;
        }
        return d5 ? conv3x3_route(d5, a5, a5.bias != nullptr, a5.in1 != nullptr, stream) : RESR_OK;
    }
    ChainJob jobs[kMaxChain];
    double flop[kMaxChain], bytes[kMaxChain];
    memset(jobs, 0, sizeof(jobs));
    for (int j = 0; j < njobs; ++j) {
        jobs[j].w = a[j].w; jobs[j].bias = a[j].bias; jobs[j].out = a[j].out;
        jobs[j].aux = (a[j].flags == bwd_flags) ? (void*)a[j].mask : (void*)a[j].aux;
        jobs[j].cin = a[j].cin; jobs[j].dep = j - 1; jobs[j].kind = 0; jobs[j].w_mt = 1; jobs[j].w_m = 0;
        jobs[j].out_lo = a[j].out_lo;
        jobs[j].w_mx = a[j].w_mx; jobs[j].out_q = a[j].out_q;
        flop[j] = 2.0 * 9 * a[j].cin * a[j].cout * (double)a[j].n * a[j].h * a[j].w_;
        bytes[j] = conv_algorithmic_bytes(a[j], x2 ? 4 : 2);
    }
    ConvArgs base = b;
    base.pair_chunks = chain_pairs;
    if (base.cin0 == base.cin) base.cin0 = base.cin;   // single-segment prefix: every chunk of every job lies in in0
    int total = njobs;
    if (with5) {
        for (int m = 0; m < 2; ++m) {
            ChainJob& q = jobs[4 + m];
            q.w = a5.w; q.w_mt = 2; q.w_m = m;
            q.out_lo = a5.out_lo; q.res0_lo = a5.res0_lo; q.res1_lo = a5.res1_lo;
            q.w_mx = a5.w_mx; q.out_q = a5.out_q;
            q.bias = a5.bias ? a5.bias + 32 * m : nullptr;
            q.out = a5.out + (size_t)m * a5.out_chunk * 2;
            q.aux = nullptr; q.cin = a5.cin; q.dep = 3; q.kind = 3;
            q.res0 = a5.res0 + (size_t)m * a5.res0_chunk * 2;
            q.res1 = a5.res1 ? a5.res1 + (size_t)m * a5.res1_chunk * 2 : nullptr;
            q.s0 = a5.s0; q.t0 = a5.t0; q.s1 = a5.s1; q.t1 = a5.t1;
            flop[4 + m] = 2.0 * 9 * a5.cin * 32 * (double)a5.n * a5.h * a5.w_;
            bytes[4 + m] = conv_algorithmic_bytes(a5, x2 ? 4 : 2) * 0.5;
        }
        if (a5.cin0 == a5.cin) base.cin0 = a5.cin;   // the single segment now reaches the closing convolution's last chunk
        total = 6;
    }
    const int rc = conv3x3_ws_chain_f16(base, jobs, total, flop, bytes, x2, chain_state, chain_state_bytes, stream);
    if (rc == 1) {   // the state buffer does not cover this geometry: one launch per job
        for (int j = 0; j < njobs; ++j) {
            const int r2 = conv3x3_route(&d[j], a[j], a[j].bias != nullptr, a[j].in1 != nullptr, stream);
            if (r2) return r2;
        }
        return d5 ? conv3x3_route(d5, a5, a5.bias != nullptr, a5.in1 != nullptr, stream) : RESR_OK;
    }
    if (rc || !d5 || with5) return rc;
    return conv3x3_route(d5, a5, a5.bias != nullptr, a5.in1 != nullptr, stream);
}

int conv3x3_chain_dispatch(int njobs, const ResrConvDesc* d, const void* in0, const void* in1, const void* const* w,
                           const float* const* bias, const void* const* mask, void* const* out, void* const* aux,
                           void* chain_state, size_t chain_state_bytes, hipStream_t stream) {
    return conv3x3_block_dispatch(njobs, d, in0, in1, w, bias, mask, out, aux, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                  chain_state, chain_state_bytes, stream);
}

}  // namespace resr
