// disc_native.hip -- whole U-Net discriminator passes enqueued natively (reference model.py:135-203 and its autograd
// backward, spectral norm of torch.nn.utils.spectral_norm included).  No kernels here: this file owns the HBM plan and
// the launch order, like generator.hip does for the generator.  One forward is ~45 launches, one backward ~100; the
// caller passes ONE workspace (activations kept for backward, this call's spectral-norm vectors and packed weights,
// all scratch), so nothing is allocated per tensor.
//
// Parameter arena (fp32, the reference's named_parameters() order):
//   conv1.weight [64,3,3,3] conv1.bias [64] | down_block1..3.0.weight_orig [128,64,4,4] [256,128,4,4] [512,256,4,4]
//   | up_block1..3.0.weight_orig [256,512,3,3] [128,256,3,3] [64,128,3,3] | conv2.0 / conv3.0 .weight_orig [64,64,3,3]
//   | conv4.weight [1,64,3,3] conv4.bias [1]
// Spectral-norm arena (fp32, named_buffers() order): per normalised layer weight_u [cout], weight_v [cin*k*k].
//
// Workspace head (fixed offsets, so the pack table can point into it): sigma[10][2] (sigma, 1/sigma per layer), then this
// call's copy of the spectral-norm arena (the backward pass of THIS call needs the u / v / sigma of THIS call: the module is
// called three times per GAN step, train_realesrgan.py:479,500,508), the packed weights, then activations and scratch.
//
// 4x4 / stride-2 convolutions: 3x3 over the 2x2 space-to-depth image with the virtual kernel of pack.hip, sparse-tap
// stage of the conv kernel (16 of 36 tap blocks), all 64-channel output groups of a layer in one launch (f16).
#include <vector>

#include "common.h"
#include "wgrad.h"

namespace resr {

int conv3x3_dispatch(const ResrConvDesc*, const void*, const void*, const void*, const float*, const void*, const void*,
                     const void*, void*, void*, hipStream_t);
size_t wgrad_batch_partial_bytes(const WgradConv*, int, int, int);
int wgrad_batch(const WgradConv*, int, int, int, int, int, int, int, float*, hipStream_t);
int wgrad_tile_rows(int dtype);
int wgrad_layer(const WgradConv*, int, int, int, int, int, int, float*, hipStream_t);   // (called as resr::wgrad_layer below: a lambda shares the name)
size_t wgrad_layer_partial_bytes(int cin, int cout_pad, int splits, int dtype);
int wgrad_x2_products();
int wgrad_x2_products();
int pack_dispatch(const ResrPackChunk*, int, const float*, void*, int, hipStream_t);
int nchw_to_nhwc_dispatch(const float*, void*, int, int, int, int, int, int, int, const uint8_t*, hipStream_t, long);
int nhwc_to_nchw_dispatch(const void*, float*, int, int, int, int, int, int, int, hipStream_t, long);
int absmax_dispatch(const float*, long, unsigned*, int, hipStream_t);
int nchw_to_nhwc_scaled_dispatch(const float*, void*, int, int, int, int, int, int, int, const uint8_t*, hipStream_t, long, const unsigned*);
int nhwc_to_nchw_scaled_dispatch(const void*, float*, int, int, int, int, int, int, int, hipStream_t, long, const unsigned*);
int s2d_dispatch(const void*, void*, int, int, int, int, int, int, hipStream_t);
int bilinear_up_dispatch(const void*, void*, int, int, int, int, int, int, hipStream_t, long, long);
int d2s_add_mask_dispatch(const void*, const void*, const void*, void*, int, int, int, int, int, float, hipStream_t, long, long, long);
int bilinear_up_bwd_mask_dispatch(const void*, void*, const void*, void*, int, int, int, int, int, float, hipStream_t, long, long);
int spectral_norm_batch_dispatch(int, const float* const*, float* const*, float* const*, const int*, const int*, int, float, float* const*,
                                 float* const*, hipStream_t);
int spectral_norm_bwd_dispatch(const float*, const float*, const float*, const float*, const float*, float*, int, int, int, float*,
                               hipStream_t);
int fold4x4_dispatch(const float*, float*, int, int, hipStream_t);
int spectral_norm_bwd_batch_dispatch(int, const float* const*, const float* const*, const float* const*, const float* const*, const float* const*,
                                     float* const*, const int*, const int*, float*, hipStream_t);
int fold4x4_batch_dispatch(int, const float* const*, float* const*, const int*, const int*, hipStream_t);

namespace {

struct Layer { int cin, cout, k4, sn, bias; };
constexpr int kLayers = 10;
const Layer kL[kLayers] = {{3, 64, 0, 0, 1},    {64, 128, 1, 1, 0},  {128, 256, 1, 1, 0}, {256, 512, 1, 1, 0}, {512, 256, 0, 1, 0},
                           {256, 128, 0, 1, 0}, {128, 64, 0, 1, 0},  {64, 64, 0, 1, 0},   {64, 64, 0, 1, 0},   {64, 1, 0, 0, 1}};
enum { CONV1 = 0, DOWN1, DOWN2, DOWN3, UP1, UP2, UP3, CONV2, CONV3, CONV4 };
constexpr float kSlope = 0.2f;

int r32(int v) { return (v + 31) / 32 * 32; }

struct DPlan {
    ResrDiscriminatorDesc d;
    size_t w_off[kLayers], b_off[kLayers], n_params;
    size_t u_off[kLayers], v_off[kLayers], n_uv;
    size_t pk_fwd[kLayers], pk_bwd[kLayers], pk_elems;   // element offsets of a layer's first group in the packed buffer
    int cin_v[kLayers], cin_pad[kLayers], cout_pad[kLayers];
    int n_chunks;
};

bool build(const ResrDiscriminatorDesc* d, DPlan& p) {
    if (!d || d->n <= 0 || d->h <= 0 || d->w <= 0 || (d->h & 7) || (d->w & 7)) return false;
    if (d->dtype != RESR_F16 && d->dtype != RESR_F32 && d->dtype != RESR_F16X2) return false;
    p.d = *d;
    size_t off = 0, uv = 0, pk = 0;
    int nch = 0;
    for (int i = 0; i < kLayers; ++i) {
        const Layer& l = kL[i];
        const int kk = l.k4 ? 16 : 9;
        p.w_off[i] = off; off += (size_t)l.cout * l.cin * kk;
        p.b_off[i] = off; if (l.bias) off += l.cout;
        p.u_off[i] = uv; if (l.sn) uv += l.cout;
        p.v_off[i] = uv; if (l.sn) uv += (size_t)l.cin * kk;
        p.cin_v[i] = l.k4 ? 4 * l.cin : l.cin;
        p.cin_pad[i] = r32(p.cin_v[i]);
        p.cout_pad[i] = r32(l.cout);
    }
    p.n_params = off; p.n_uv = uv;
    for (int i = 0; i < kLayers; ++i) {   // forward groups of a layer, then its backward-data groups (as pack_table emits them)
        p.pk_fwd[i] = pk;
        for (int g0 = 0; g0 < p.cout_pad[i]; g0 += 64) {
            const int mt = (p.cout_pad[i] - g0 < 64 ? p.cout_pad[i] - g0 : 64) / 32;
            pk += (size_t)(p.cin_pad[i] / 32) * 9 * mt * 1024; nch += p.cin_pad[i] / 32;
        }
        p.pk_bwd[i] = pk;
        for (int g0 = 0; g0 < p.cin_pad[i]; g0 += 64) {
            const int mt = (p.cin_pad[i] - g0 < 64 ? p.cin_pad[i] - g0 : 64) / 32;
            pk += (size_t)(p.cout_pad[i] / 32) * 9 * mt * 1024; nch += p.cout_pad[i] / 32;
        }
    }
    p.pk_elems = pk; p.n_chunks = nch;
    return true;
}

struct DBufs {
    float* sigma;      // [kLayers][2]
    float* uv;         // this call's spectral-norm vectors
    char* packed;
    float* sn_tmp;
    // activations (T)
    char *x_in, *out1, *s1, *d1, *s2, *d2, *s3, *d3, *b1, *u1, *a1, *b2, *u2, *a2, *b3, *u3, *a3, *c2, *c3;
    // backward scratch (T)
    char *g4, *G8, *G7, *G6, *g_u3, *g_b3, *g_u2, *G5, *g_b2, *g_u1, *G4, *g_b1, *g_d3, *G3, *g_s3, *G2, *g_s2, *G1, *g_s1,
        *G0, *gxin;
    float *raw, *folded, *tmp1, *partial;
    unsigned* gscale;           // exact16: bits of max |g_y| of the running backward pass (common.h: grad_prescale)
    float* raw_l[kLayers];      // per normalised layer: the gradient wrt W = W_orig / sigma (virtual 3x3 form for the 4x4 layers) ...
    float* fold_l[kLayers];     // ... and its [cout][C][4][4] form (4x4 layers): all layers' folds and spectral-norm backward steps
                                // run as three batched launches at the end of the pass
    size_t partial_bytes, total;
};

int wsplits(int dtype, int jobs, int n, int h, int w) {
    const int th = wgrad_tile_rows(dtype);
    const long tiles = (long)((w + 31) / 32) * ((h + th - 1) / th) * n;
    long s;
    if (dtype != RESR_F32) { s = 512 / ((jobs + 3) / 4); if (s >= 16) s &= ~7L; if (s > 256) s = 256; }
    else { s = 768 / jobs; if (s > 128) s = 128; }
    if (s > tiles / 2) s = tiles / 2;
    return (int)(s < 1 ? 1 : s);
}

// Layers with more products than a launch's job table holds (>= 96: the 256..512-channel layers) run their weight gradients as ONE
// layer-mode launch pair (wgrad.hip, WgradLayer): one residency round of workgroups -- 256 / quad jobs pixel splits -- instead of
// 2..8 launch pairs of 24 splits each (the 512-channel 4x4 layer at 16 x 32^2: eight pairs of 52 us for 69 GFLOP).
constexpr int kLayerModeProducts = 96;
int layer_splits(int products, int n, int h, int w, int parts = 1) {
    const long tiles = (long)((w + 31) / 32) * ((h + 7) / 8) * n;
    const int nq = ((products + 3) / 4) * parts;
    long s = 256 / nq;
    if (s >= 16) s &= ~7L;
    if (parts > 1) {
        // exact16's three tap-products per product: up to 384 quad jobs -- more than one residency round.  The split count (<= 8)
        // whose workgroups fill whole rounds of 256 best, the smaller one on a tie (fewer slabs)
        double best = -1.0;
        for (long c = 1; c <= 8; ++c) {
            const long wg = (long)nq * c, rounds = (wg + 255) / 256;
            const double fill = (double)wg / (double)(rounds * 256);
            if (fill > best + 1e-9) { best = fill; s = c; }
        }
    }
    if (s > tiles / 2) s = tiles / 2;
    return (int)(s < 1 ? 1 : s);
}

// RESR_F16X2: every activation / gradient buffer below holds the hi tensor and, directly behind it, the lo tensor (hi -> lo
// element offset = the tensor's element count); packed weights take three f16 blocks per chunk.
void carve(const DPlan& p, char* base, DBufs& b) {
    const size_t es = elem_size(p.d.dtype) * act_tensors(p.d.dtype);
    const size_t px = (size_t)p.d.n * p.d.h * p.d.w;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* q = base ? base + off : nullptr; off += align_up(bytes, 256); return q; };
    b.sigma = (float*)take(kLayers * 2 * sizeof(float));   // (64 of the slot's 256 bytes)
    b.gscale = base ? reinterpret_cast<unsigned*>(base + 128) : nullptr;   // the gradient pre-scale slot (common.h): inside the head the caller zero-fills once
    b.uv = (float*)take(p.n_uv * sizeof(float));
    b.packed = take(p.pk_elems * elem_size(p.d.dtype) * (p.d.dtype == RESR_F16X2 ? 3 : 1) + 16384);
    {   // spectral-norm scratch of every normalised layer (all layers iterate in the same four launches): rows + ceil(rows / 32) * cols floats each
        size_t fl = 0;
        for (int i = 0; i < kLayers; ++i)
            if (kL[i].sn) fl += align_up((size_t)kL[i].cout + (size_t)((kL[i].cout + 31) / 32) * kL[i].cin * (kL[i].k4 ? 16 : 9), 64);
        b.sn_tmp = (float*)take(fl * sizeof(float));
    }
    b.x_in = take(px * 32 * es);
    b.out1 = take(px * 64 * es);
    b.s1 = take(px / 4 * 256 * es);   b.d1 = take(px / 4 * 128 * es);
    b.s2 = take(px / 16 * 512 * es);  b.d2 = take(px / 16 * 256 * es);
    b.s3 = take(px / 64 * 1024 * es); b.d3 = take(px / 64 * 512 * es);
    b.b1 = take(px / 16 * 512 * es);  b.u1 = take(px / 16 * 256 * es);
    b.b2 = take(px / 4 * 256 * es);   b.u2 = take(px / 4 * 128 * es);
    b.b3 = take(px * 128 * es);       b.u3 = take(px * 64 * es);
    b.c2 = take(px * 64 * es);        b.c3 = take(px * 64 * es);
    if (p.d.training) {
        b.a1 = take(px / 16 * 256 * es); b.a2 = take(px / 4 * 128 * es); b.a3 = take(px * 64 * es);
        b.g4 = take(px * 32 * es);
        b.G8 = take(px * 64 * es); b.G7 = take(px * 64 * es); b.G6 = take(px * 64 * es); b.g_u3 = take(px * 64 * es);
        b.g_b3 = take(px * 128 * es);
        b.g_u2 = take(px / 4 * 128 * es); b.G5 = take(px / 4 * 128 * es); b.g_b2 = take(px / 4 * 256 * es);
        b.g_u1 = take(px / 16 * 256 * es); b.G4 = take(px / 16 * 256 * es); b.g_b1 = take(px / 16 * 512 * es);
        b.g_d3 = take(px / 64 * 512 * es); b.G3 = take(px / 64 * 512 * es); b.g_s3 = take(px / 64 * 1024 * es);
        b.G2 = take(px / 16 * 256 * es); b.g_s2 = take(px / 16 * 512 * es);
        b.G1 = take(px / 4 * 128 * es); b.g_s1 = take(px / 4 * 256 * es);
        b.G0 = take(px * 64 * es); b.gxin = take(px * 32 * es);
        b.raw = (float*)take((size_t)512 * 1024 * 9 * sizeof(float));
        b.folded = (float*)take((size_t)512 * 256 * 16 * sizeof(float));
        b.tmp1 = (float*)take(8 * 512 * sizeof(float));   // block partials of the spectral-norm backward's <G, W>, all layers
        for (int li = 0; li < kLayers; ++li) {
            b.raw_l[li] = b.fold_l[li] = nullptr;
            if (!kL[li].sn) continue;
            b.raw_l[li] = (float*)take((size_t)kL[li].cout * p.cin_v[li] * 9 * sizeof(float));
            if (kL[li].k4) b.fold_l[li] = (float*)take((size_t)kL[li].cout * kL[li].cin * 16 * sizeof(float));
        }
        // weight-gradient slabs: the largest launch is <= 80 products (wgrad.hip kMaxJobs) x its splits
        size_t pb = 0;
        const int res[4][2] = {{p.d.h, p.d.w}, {p.d.h / 2, p.d.w / 2}, {p.d.h / 4, p.d.w / 4}, {p.d.h / 8, p.d.w / 8}};
        for (int q = 0; q < 4; ++q)
            for (int jobs = 1; jobs <= 96; ++jobs) {
                const size_t v = (size_t)jobs * wsplits(p.d.dtype, jobs, p.d.n, res[q][0], res[q][1]) * (9 * 1024 + 32) * sizeof(float);
                if (v > pb) pb = v;
            }
        if (p.d.dtype == RESR_F16 || p.d.dtype == RESR_F16X2)   // layer-mode launches (one per 256..512-channel layer)
            for (int li = 0; li < kLayers; ++li) {
                const int products = (p.cin_pad[li] / 32) * (p.cout_pad[li] / 32);
                if (products < kLayerModeProducts) continue;
                const int q = li == DOWN1 || li == UP2 ? 1 : li == DOWN2 || li == UP1 ? 2 : li == DOWN3 ? 3 : 0;
                // exact16: one or three tap-products per product ($RESR_X2_WGRAD_PRODUCTS is read per call) -- room for either
                for (int parts = 1; parts <= (p.d.dtype == RESR_F16X2 ? 3 : 1); parts += 2) {
                    const size_t v = (size_t)products * parts * layer_splits(products, p.d.n, res[q][0], res[q][1], parts) * (9 * 1024 + 32) * sizeof(float);
                    if (v > pb) pb = v;
                }
            }
        b.partial_bytes = pb;
        b.partial = (float*)take(pb);
    } else {
        b.a1 = b.a2 = b.a3 = nullptr;
        b.partial = nullptr; b.partial_bytes = 0;
    }
    b.total = off;
}

ResrConvDesc cdesc(const DPlan& p, int n, int h, int w, int cin_pad, int in_stride, int cout, int cout_pad, int out_stride, int flags) {
    ResrConvDesc c;
    memset(&c, 0, sizeof(c));
    c.n = n; c.h = h; c.w = w; c.cin = cin_pad; c.cin0 = cin_pad; c.in0_stride = in_stride;
    c.cout = cout; c.cout_pad = cout_pad; c.out_stride = out_stride; c.dtype = p.d.dtype; c.flags = flags;
    c.s0 = c.t0 = c.s1 = c.t1 = 1.f; c.slope = kSlope;
    return c;
}

#define DRUN(expr) do { int rc_ = (expr); if (rc_ != RESR_OK) return rc_; } while (0)

// One layer as 3x3 conv of NHWC x (first cin_pad channels, pixel stride in_stride) into NHWC out (pixel stride out_stride), in
// 64-channel output groups: one launch when the kernel takes groups (f16, no bias), else one per group.
// backward = packed backward-data form (M = cin_v groups, K = cout).
// lo_x / lo_out / lo_res0: RESR_F16X2 hi -> lo element offsets of x, out (and aux) and res0; 0 otherwise.
int conv_layer(const DPlan& p, const DBufs& b, int li, bool backward, const char* x, int in_stride, int n, int h, int w, char* out,
               int out_stride, int flags, const float* bias, const char* res0, int res0_stride, const char* mask, int mask_stride,
               char* aux, int s2d_in, int s2d_out, hipStream_t st, long lo_x, long lo_out, long lo_res0, float* out_nchw = nullptr) {
    const size_t es = elem_size(p.d.dtype);
    const size_t wes = es * (p.d.dtype == RESR_F16X2 ? 3 : 1);   // bytes per element of the plain packed layout
    const int kin = backward ? p.cout_pad[li] : p.cin_pad[li];                  // K channels read
    const int mtot = backward ? p.cin_pad[li] : p.cout_pad[li];                 // M channels written (padded)
    const bool nchw = flags & RESR_CONV_OUT_NCHW_F32;
    // NHWC outputs are written in whole 32-channel chunks (packed rows beyond the real count are zero); the planar fp32
    // output of conv4 holds the real channels only
    const int mreal = nchw ? kL[li].cout : r32(backward ? p.cin_v[li] : kL[li].cout);
    const size_t pk0 = backward ? p.pk_bwd[li] : p.pk_fwd[li];
    const int ngroups = (mtot + 63) / 64;
    if (ngroups > 1 && p.d.dtype != RESR_F32 && mreal == 64 * ngroups && !bias && !nchw && (flags & RESR_CONV_NO_BIAS)) {
        ResrConvDesc c = cdesc(p, n, h, w, kin, in_stride, 64, 64, out_stride, flags);
        c.res0_stride = res0_stride; c.mask_stride = mask_stride;
        c.cout_groups = ngroups; c.s2d_in_channels = s2d_in; c.s2d_out_channels = s2d_out;
        c.in0_lo_offset = lo_x; c.out_lo_offset = lo_out; c.res0_lo_offset = lo_res0;
        c.mask_lo_offset = mask ? lo_out : 0;   // the mask is a saved activation of out's shape
        return conv3x3_dispatch(&c, x, nullptr, b.packed + pk0 * wes, nullptr, res0, nullptr, mask, out, aux, st);
    }
    size_t pk = pk0;
    for (int g0 = 0; g0 < mtot; g0 += 64) {
        const int mt = (mtot - g0 < 64 ? mtot - g0 : 64) / 32;
        int co = mreal - g0; if (co > mt * 32) co = mt * 32;
        if (co > 0) {
            ResrConvDesc c = cdesc(p, n, h, w, kin, in_stride, co, mt * 32, nchw ? 0 : out_stride, flags);
            c.res0_stride = res0_stride; c.mask_stride = mask_stride;
            c.s2d_in_channels = (ngroups == 1 ? s2d_in : 0);
            c.in0_lo_offset = lo_x; c.out_lo_offset = lo_out; c.res0_lo_offset = lo_res0;
        c.mask_lo_offset = mask ? lo_out : 0;   // the mask is a saved activation of out's shape
            auto sh = [&](const char* q) { return q ? q + (size_t)g0 * es : nullptr; };
            DRUN(conv3x3_dispatch(&c, x, nullptr, b.packed + pk * wes, bias ? bias + g0 : nullptr, sh(res0), nullptr, sh(mask),
                                  nchw ? (void*)out_nchw : (void*)(out + (size_t)g0 * es), aux ? aux + (size_t)g0 * es : nullptr, st));
        }
        pk += (size_t)(kin / 32) * 9 * mt * 1024;
    }
    return RESR_OK;
}

}  // namespace

size_t discriminator_param_count() { DPlan p; ResrDiscriminatorDesc d = {1, 8, 8, RESR_F16, 0, 0}; return build(&d, p) ? p.n_params : 0; }
size_t discriminator_uv_count() { DPlan p; ResrDiscriminatorDesc d = {1, 8, 8, RESR_F16, 0, 0}; return build(&d, p) ? p.n_uv : 0; }

size_t discriminator_workspace_bytes(const ResrDiscriminatorDesc* d) {
    DPlan p;
    if (!build(d, p)) return 0;
    DBufs b;
    carve(p, nullptr, b);
    return b.total;
}

// chunk table of resr_pack_weights for all layers (forward groups, then backward-data groups, per layer); 1/sigma of the
// normalised layers is read on the device from the workspace head, so the table depends on the workspace address
int64_t discriminator_pack_table(const ResrDiscriminatorDesc* d, const void* workspace, ResrPackChunk* out, int64_t cap) {
    DPlan p;
    if (!build(d, p)) return fail(RESR_ERR_ARG, "discriminator: bad descriptor");
    if (!out) return p.n_chunks;
    if (cap < p.n_chunks || !workspace) return fail(RESR_ERR_ARG, "discriminator_pack_table: capacity / workspace");
    DBufs b;
    carve(p, (char*)workspace, b);
    int64_t n = 0;
    for (int li = 0; li < kLayers; ++li) {
        const Layer& l = kL[li];
        const float* sp = l.sn ? b.sigma + li * 2 + 1 : nullptr;
        for (int pass = 0; pass < 2; ++pass) {   // 0: forward (M = cout, K = cin_v); 1: backward-data (M = cin_v, K = cout)
            const int mtot = pass ? p.cin_pad[li] : p.cout_pad[li], mreal = pass ? p.cin_v[li] : l.cout;
            const int ktot = pass ? p.cout_pad[li] : p.cin_pad[li], kreal = pass ? l.cout : p.cin_v[li];
            size_t pk = pass ? p.pk_bwd[li] : p.pk_fwd[li];
            for (int g0 = 0; g0 < mtot; g0 += 64) {
                const int mt = (mtot - g0 < 64 ? mtot - g0 : 64) / 32;
                for (int ck = 0; ck < ktot / 32; ++ck) {
                    ResrPackChunk c;
                    memset(&c, 0, sizeof(c));
                    c.src_off = (int64_t)p.w_off[li]; c.dst_off = (int64_t)pk;
                    c.src_cout = l.cout; c.src_cin = l.cin;
                    c.m_off = g0; c.m_count = mreal - g0 < 64 ? (mreal - g0 > 0 ? mreal - g0 : 0) : 64;
                    c.k_off = ck * 32; c.k_count = kreal - ck * 32 < 32 ? (kreal - ck * 32 > 0 ? kreal - ck * 32 : 0) : 32;
                    c.mt = mt; c.transposed = pass; c.scale = 1.f; c.virtual4x4 = l.k4; c.scale_ptr = sp;
                    out[n++] = c;
                    pk += (size_t)9 * mt * 1024;
                }
            }
        }
    }
    return n;
}

int discriminator_forward(const ResrDiscriminatorDesc* d, const float* x, const float* params, float* uv, const ResrPackChunk* table,
                          int n_chunks, void* workspace, size_t workspace_bytes, float* y, hipStream_t st) {
    DPlan p;
    if (!build(d, p)) return fail(RESR_ERR_ARG, "discriminator_forward: bad descriptor (H, W must be multiples of 8)");
    if (!x || !params || !uv || !table || !workspace || !y) return fail(RESR_ERR_ARG, "discriminator_forward: null argument");
    if (n_chunks != p.n_chunks) return fail(RESR_ERR_ARG, "discriminator_forward: pack table has %d chunks, expected %d", n_chunks, p.n_chunks);
    DBufs b;
    carve(p, (char*)workspace, b);
    if (b.total > workspace_bytes) return fail(RESR_ERR_WORKSPACE, "discriminator_forward: workspace %zu < %zu", workspace_bytes, b.total);
    const int dt = d->dtype, N = d->n, S = d->h, W = d->w;
    const int H1 = S / 2, W1 = W / 2, H2 = S / 4, W2 = W / 4, H3 = S / 8, W3 = W / 8;
    // spectral norm: one power iteration per training-mode call, u / v updated in place like torch's hook (model.py:140-168)
    {   // all eight normalised layers in the same four launches
        const float* Ws[kLayers]; float* us[kLayers]; float* vs[kLayers]; float* sg[kLayers]; float* tm[kLayers];
        int rws[kLayers], cls[kLayers], n = 0;
        size_t fl = 0;
        for (int li = 0; li < kLayers; ++li) {
            if (!kL[li].sn) continue;
            const int cols = kL[li].cin * (kL[li].k4 ? 16 : 9);
            Ws[n] = params + p.w_off[li]; us[n] = uv + p.u_off[li]; vs[n] = uv + p.v_off[li]; sg[n] = b.sigma + li * 2; tm[n] = b.sn_tmp + fl;
            rws[n] = kL[li].cout; cls[n] = cols;
            fl += align_up((size_t)kL[li].cout + (size_t)((kL[li].cout + 31) / 32) * cols, 64);
            ++n;
        }
        DRUN(spectral_norm_batch_dispatch(n, Ws, us, vs, rws, cls, d->sn_training, 1e-12f, sg, tm, st));
    }
    if (d->training && hipMemcpyAsync(b.uv, uv, p.n_uv * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return fail(RESR_ERR_LAUNCH, "discriminator_forward: hipMemcpyAsync");
    DRUN(pack_dispatch(table, n_chunks, params, b.packed, dt, st));
    const bool x2 = dt == RESR_F16X2;
    const long px = (long)N * S * W;
    auto LO = [&](long elems) -> long { return x2 ? elems : 0L; };   // hi -> lo offset of a buffer of `elems` elements
    const long lo_in = LO(px * 32), lo64 = LO(px * 64), lo_d1 = LO(px / 4 * 128), lo_d2 = LO(px / 16 * 256), lo_d3 = LO(px / 64 * 512);
    const long lo_b1 = LO(px / 16 * 512), lo_b2 = LO(px / 4 * 256), lo_b3 = LO(px * 128);
    DRUN(nchw_to_nhwc_dispatch(x, b.x_in, N, 3, S, W, 1, 32, dt, nullptr, st, lo_in));
    const int NB = RESR_CONV_NO_BIAS, LR = RESR_CONV_LRELU;
    DRUN(conv_layer(p, b, CONV1, false, b.x_in, 32, N, S, W, b.out1, 64, 0, params + p.b_off[CONV1], nullptr, 0, nullptr, 0, nullptr, 0, 0, st, lo_in, lo64, 0));
    DRUN(s2d_dispatch(b.out1, b.s1, N, S, W, 64, dt, 0, st));                    // (hi and lo as one batch of 2N)
    DRUN(conv_layer(p, b, DOWN1, false, b.s1, 256, N, H1, W1, b.d1, 128, LR | NB, nullptr, nullptr, 0, nullptr, 0, nullptr, 64, 0, st, lo64, lo_d1, 0));
    DRUN(s2d_dispatch(b.d1, b.s2, N, H1, W1, 128, dt, 0, st));
    DRUN(conv_layer(p, b, DOWN2, false, b.s2, 512, N, H2, W2, b.d2, 256, LR | NB, nullptr, nullptr, 0, nullptr, 0, nullptr, 128, 0, st, lo_d1, lo_d2, 0));
    DRUN(s2d_dispatch(b.d2, b.s3, N, H2, W2, 256, dt, 0, st));
    DRUN(conv_layer(p, b, DOWN3, false, b.s3, 1024, N, H3, W3, b.d3, 512, LR | NB, nullptr, nullptr, 0, nullptr, 0, nullptr, 256, 0, st, lo_d2, lo_d3, 0));
    const int FL = LR | NB | (d->training ? RESR_CONV_AUX_BEFORE_RES : 0);
    DRUN(bilinear_up_dispatch(b.d3, b.b1, N, H3, W3, 512, dt, 0, st, lo_d3, lo_b1));
    DRUN(conv_layer(p, b, UP1, false, b.b1, 512, N, H2, W2, b.u1, 256, FL, nullptr, b.d2, 256, nullptr, 0, b.a1, 0, 0, st, lo_b1, lo_d2, lo_d2));
    DRUN(bilinear_up_dispatch(b.u1, b.b2, N, H2, W2, 256, dt, 0, st, lo_d2, lo_b2));
    DRUN(conv_layer(p, b, UP2, false, b.b2, 256, N, H1, W1, b.u2, 128, FL, nullptr, b.d1, 128, nullptr, 0, b.a2, 0, 0, st, lo_b2, lo_d1, lo_d1));
    DRUN(bilinear_up_dispatch(b.u2, b.b3, N, H1, W1, 128, dt, 0, st, lo_d1, lo_b3));
    DRUN(conv_layer(p, b, UP3, false, b.b3, 128, N, S, W, b.u3, 64, FL, nullptr, b.out1, 64, nullptr, 0, b.a3, 0, 0, st, lo_b3, lo64, lo64));
    DRUN(conv_layer(p, b, CONV2, false, b.u3, 64, N, S, W, b.c2, 64, LR | NB, nullptr, nullptr, 0, nullptr, 0, nullptr, 0, 0, st, lo64, lo64, 0));
    DRUN(conv_layer(p, b, CONV3, false, b.c2, 64, N, S, W, b.c3, 64, LR | NB, nullptr, nullptr, 0, nullptr, 0, nullptr, 0, 0, st, lo64, lo64, 0));
    DRUN(conv_layer(p, b, CONV4, false, b.c3, 64, N, S, W, nullptr, 0, RESR_CONV_OUT_NCHW_F32, params + p.b_off[CONV4], nullptr, 0, nullptr, 0,
                    nullptr, 0, 0, st, lo64, 0, 0, y));
    return RESR_OK;
}

int discriminator_backward(const ResrDiscriminatorDesc* d, const float* gy, const float* params, void* workspace, size_t workspace_bytes,
                           float* grad, float* gx, hipStream_t st) {
    DPlan p;
    if (!build(d, p)) return fail(RESR_ERR_ARG, "discriminator_backward: bad descriptor");
    if (!d->training) return fail(RESR_ERR_ARG, "discriminator_backward: forward was not run with training=1");
    if (!gy || !params || !workspace) return fail(RESR_ERR_ARG, "discriminator_backward: null argument");
    DBufs b;
    carve(p, (char*)workspace, b);
    if (b.total > workspace_bytes) return fail(RESR_ERR_WORKSPACE, "discriminator_backward: workspace %zu < %zu", workspace_bytes, b.total);
    const int dt = d->dtype, N = d->n, S = d->h, W = d->w;
    const size_t es = elem_size(dt);
    const int H1 = S / 2, W1 = W / 2, H2 = S / 4, W2 = W / 4, H3 = S / 8, W3 = W / 8;
    const int NB = RESR_CONV_NO_BIAS, MK = RESR_CONV_MASK;
    const bool need_w = grad != nullptr;
    const bool x2 = dt == RESR_F16X2;
    const long px = (long)N * S * W;
    auto LO = [&](long elems) -> long { return x2 ? elems : 0L; };   // hi -> lo offset of a buffer of `elems` elements
    const long lo_in = LO(px * 32), lo64 = LO(px * 64), lo128 = LO(px * 128);
    const long lo_h1_128 = LO(px / 4 * 128), lo_h1_256 = LO(px / 4 * 256), lo_h2_256 = LO(px / 16 * 256), lo_h2_512 = LO(px / 16 * 512);
    const long lo_h3_512 = LO(px / 64 * 512), lo_h3_1024 = LO(px / 64 * 1024);

    // exact16 and fast: a small incoming gradient is lifted into f16's normal range by a power of two and every result handed out unscaled
    // (generator.hip has the measurements; common.h grad_prescale).  The spectral-norm backward at the end of the pass reads the
    // raw weight gradients the reducers have already unscaled.
    const char* pre_t = getenv("RESR_X2_GRAD_PRESCALE_LOG2");
    const unsigned* gsc = (dt != RESR_F32 && !getenv("RESR_X2_NO_GRAD_PRESCALE")) ? b.gscale : nullptr;
    // weight (and bias) gradient of layer li: X = first cin_pad channels of x (pixel stride xs), G = first cout channels of g
    auto wgrad_layer = [&](int li, const char* x, int xs, const char* g, int gs, int h, int w, long lo_xw, long lo_gw) -> int {
        if (!need_w) return RESR_OK;
        const Layer& l = kL[li];
        const int cin_pad = p.cin_pad[li], cin_v = p.cin_v[li], chunks = cin_pad / 32;
        float* dst = grad + p.w_off[li];
        float* raw = (!l.sn && !l.k4) ? dst : (l.sn ? b.raw_l[li] : b.raw);
        // as many 32-channel output tiles per launch as the weight-gradient launcher takes: <= 96 tap-products (wgrad.h) and <= 80
        // algorithmic products (its reduction's argument block) -- a 128..256-channel layer is one or two launch pairs, not 2..4
        const int parts = x2 ? wgrad_x2_products() : 1;
        const char* no_layer_mode = getenv("RESR_WGRAD_NO_LAYER_MODE");   // A/B knob, read per call: the table-mode launch pairs
        if ((dt == RESR_F16 || dt == RESR_F16X2) && chunks * (r32(l.cout) / 32) >= kLayerModeProducts && !no_layer_mode) {
            WgradConv c;
            c.x0 = x; c.cin = cin_pad; c.in0_stride = xs; c.cin_real = cin_v;
            c.g = g; c.cout = l.cout; c.cout_pad = r32(l.cout); c.g_stride = gs;
            c.x_chunk_stride = c.g_chunk_stride = 0; c.x_lo_off = x2 ? lo_xw : 0; c.g_lo_off = x2 ? lo_gw : 0;
            c.x_s2d_c = l.k4 ? l.cin : 0; c.g_lo_bias_only = 0;
            c.dw = raw; c.db = (l.bias ? grad + p.b_off[li] : nullptr); c.scale = 1.f;
            c.unscale = gsc;
            const int splits = layer_splits(chunks * (c.cout_pad / 32), N, h, w, parts);
            if (wgrad_layer_partial_bytes(cin_pad, c.cout_pad, splits, dt) > b.partial_bytes) return fail(RESR_ERR_WORKSPACE, "discriminator: wgrad slabs (layer mode)");
            DRUN(resr::wgrad_layer(&c, N, h, w, dt, 0, splits, b.partial, st));
        } else {
        int tiles_per = kWgradMaxJobs / (chunks * parts);
        if (tiles_per > 80 / chunks) tiles_per = 80 / chunks;
        if (tiles_per < 1) tiles_per = 1;
        const int step = tiles_per * 32;
        for (int g0 = 0; g0 < r32(l.cout); g0 += step) {       // one launch pair per `step` output channels, as convolutions of <= 64
            WgradConv cs[8];
            int nc = 0, jobs = 0;
            for (int q0 = g0; q0 < g0 + step && q0 < r32(l.cout) && nc < 8; q0 += 64) {
                int co = l.cout - q0; if (co > 64) co = 64;
                if (g0 + step - q0 < co) co = g0 + step - q0;
                if (co <= 0) continue;
                WgradConv& c = cs[nc++];
                c.x0 = x; c.cin = cin_pad; c.in0_stride = xs; c.cin_real = cin_v;
                c.g = g + (size_t)q0 * es; c.cout = co; c.cout_pad = r32(co); c.g_stride = gs;
                c.x_chunk_stride = c.g_chunk_stride = 0; c.x_lo_off = lo_xw; c.g_lo_off = lo_gw;
                c.g_lo_bias_only = 0;
                c.x_s2d_c = l.k4 ? l.cin : 0;      // 4x4 / stride-2 layers: X is the space-to-depth image, skip the virtual kernel's zero taps
                c.dw = raw + (size_t)q0 * cin_v * 9; c.db = (l.bias ? grad + p.b_off[li] + q0 : nullptr); c.scale = 1.f;
                c.unscale = gsc;
                jobs += chunks * (c.cout_pad / 32);
            }
            if (!nc) continue;
            // slabs are per TAP-product (three per algorithmic product in exact16's default form): split by that count, which is
            // also what carve() sized the slab buffer for
            const int splits = wsplits(dt, jobs * parts, N, h, w);
            if (wgrad_batch_partial_bytes(cs, nc, splits, dt) > b.partial_bytes) return fail(RESR_ERR_WORKSPACE, "discriminator: wgrad slabs");
            DRUN(wgrad_batch(cs, nc, N, h, w, dt, 0, splits, b.partial, st));
        }
        }
        if (l.sn) return RESR_OK;   // fold + spectral-norm backward of all normalised layers: batched at the end of the pass
        if (l.k4) { DRUN(fold4x4_dispatch(raw, b.folded, l.cout, l.cin, st)); }   // (no such layer: every 4x4 layer is normalised)
        return RESR_OK;
    };
    // gradient wrt weight_orig from the gradient wrt W = weight_orig / sigma, with THIS call's u, v, sigma -- every normalised layer
    // in three launches (fold of the 4x4 layers, <G, W> partials, apply) instead of 38
    auto finish_sn = [&]() -> int {
        if (!need_w) return RESR_OK;
        const float* fs[4]; float* fd[4]; int fco[4], fc[4]; int nf = 0;
        const float* G[8]; const float* Wp[8]; const float* up[8]; const float* vp[8]; const float* sg[8]; float* ds[8]; int rws[8], cls[8]; int n = 0;
        for (int li = 0; li < kLayers; ++li) {
            const Layer& l = kL[li];
            if (!l.sn) continue;
            if (l.k4) { fs[nf] = b.raw_l[li]; fd[nf] = b.fold_l[li]; fco[nf] = l.cout; fc[nf] = l.cin; ++nf; }
            G[n] = l.k4 ? b.fold_l[li] : b.raw_l[li]; Wp[n] = params + p.w_off[li]; up[n] = b.uv + p.u_off[li]; vp[n] = b.uv + p.v_off[li];
            sg[n] = b.sigma + li * 2; ds[n] = grad + p.w_off[li]; rws[n] = l.cout; cls[n] = l.cin * (l.k4 ? 16 : 9);
            ++n;
        }
        if (nf) DRUN(fold4x4_batch_dispatch(nf, fs, fd, fco, fc, st));
        return spectral_norm_bwd_batch_dispatch(n, G, Wp, up, vp, sg, ds, rws, cls, b.tmp1, st);
    };
    auto dconv = [&](int li, const char* g, int gs, int h, int w, char* out, int out_stride, int flags, const char* mask, int mask_stride,
                     char* aux, int s2d_out, long lo_g, long lo_o) {
        return conv_layer(p, b, li, true, g, gs, N, h, w, out, out_stride, flags | NB, nullptr, nullptr, 0, mask, mask_stride, aux, 0, s2d_out, st,
                          lo_g, lo_o, 0);
    };

    if (gsc) DRUN(absmax_dispatch(gy, (long)N * S * W, b.gscale, pre_t ? atoi(pre_t) : 6, st));
    DRUN(nchw_to_nhwc_scaled_dispatch(gy, b.g4, N, 1, S, W, 1, 32, dt, nullptr, st, lo_in, gsc));
    DRUN(wgrad_layer(CONV4, b.c3, 64, b.g4, 32, S, W, lo64, lo_in));
    DRUN(dconv(CONV4, b.g4, 32, S, W, b.G8, 64, MK, b.c3, 64, nullptr, 0, lo_in, lo64));
    DRUN(wgrad_layer(CONV3, b.c2, 64, b.G8, 64, S, W, lo64, lo64));
    DRUN(dconv(CONV3, b.G8, 64, S, W, b.G7, 64, MK, b.c2, 64, nullptr, 0, lo64, lo64));
    DRUN(wgrad_layer(CONV2, b.u3, 64, b.G7, 64, S, W, lo64, lo64));
    DRUN(dconv(CONV2, b.G7, 64, S, W, b.G6, 64, MK | RESR_CONV_AUX_BEFORE_MASK, b.a3, 64, b.g_u3, 0, lo64, lo64));
    DRUN(wgrad_layer(UP3, b.b3, 128, b.G6, 64, S, W, lo128, lo64));
    DRUN(dconv(UP3, b.G6, 64, S, W, b.g_b3, 128, 0, nullptr, 0, nullptr, 0, lo64, lo128));
    DRUN(bilinear_up_bwd_mask_dispatch(b.g_b3, b.g_u2, b.a2, b.G5, N, H1, W1, 128, dt, kSlope, st, lo128, lo_h1_128));   // g_u2 (skip gradient) and G5 = masked
    DRUN(wgrad_layer(UP2, b.b2, 256, b.G5, 128, H1, W1, lo_h1_256, lo_h1_128));
    DRUN(dconv(UP2, b.G5, 128, H1, W1, b.g_b2, 256, 0, nullptr, 0, nullptr, 0, lo_h1_128, lo_h1_256));
    DRUN(bilinear_up_bwd_mask_dispatch(b.g_b2, b.g_u1, b.a1, b.G4, N, H2, W2, 256, dt, kSlope, st, lo_h1_256, lo_h2_256));
    DRUN(wgrad_layer(UP1, b.b1, 512, b.G4, 256, H2, W2, lo_h2_512, lo_h2_256));
    DRUN(dconv(UP1, b.G4, 256, H2, W2, b.g_b1, 512, 0, nullptr, 0, nullptr, 0, lo_h2_256, lo_h2_512));
    DRUN(bilinear_up_bwd_mask_dispatch(b.g_b1, b.g_d3, b.d3, b.G3, N, H3, W3, 512, dt, kSlope, st, lo_h2_512, lo_h3_512));
    DRUN(wgrad_layer(DOWN3, b.s3, 1024, b.G3, 512, H3, W3, lo_h3_1024, lo_h3_512));
    DRUN(dconv(DOWN3, b.G3, 512, H3, W3, b.g_s3, 1024, 0, nullptr, 0, nullptr, 256, lo_h3_512, lo_h3_1024));
    DRUN(d2s_add_mask_dispatch(b.g_s3, b.g_u1, b.d2, b.G2, N, H2, W2, 256, dt, kSlope, st, lo_h3_1024, lo_h2_256, lo_h2_256));   // depth-to-space + skip gradient + LeakyReLU backward
    DRUN(wgrad_layer(DOWN2, b.s2, 512, b.G2, 256, H2, W2, lo_h2_512, lo_h2_256));
    DRUN(dconv(DOWN2, b.G2, 256, H2, W2, b.g_s2, 512, 0, nullptr, 0, nullptr, 128, lo_h2_256, lo_h2_512));
    DRUN(d2s_add_mask_dispatch(b.g_s2, b.g_u2, b.d1, b.G1, N, H1, W1, 128, dt, kSlope, st, lo_h2_512, lo_h1_128, lo_h1_128));
    DRUN(wgrad_layer(DOWN1, b.s1, 256, b.G1, 128, H1, W1, lo_h1_256, lo_h1_128));
    DRUN(dconv(DOWN1, b.G1, 128, H1, W1, b.g_s1, 256, 0, nullptr, 0, nullptr, 64, lo_h1_128, lo_h1_256));
    DRUN(d2s_add_mask_dispatch(b.g_s1, b.g_u3, nullptr, b.G0, N, S, W, 64, dt, kSlope, st, lo_h1_256, lo64, lo64));
    DRUN(wgrad_layer(CONV1, b.x_in, 32, b.G0, 64, S, W, lo_in, lo64));
    DRUN(finish_sn());
    if (gx) {
        DRUN(dconv(CONV1, b.G0, 64, S, W, b.gxin, 32, 0, nullptr, 0, nullptr, 0, lo64, lo_in));
        DRUN(nhwc_to_nchw_scaled_dispatch(b.gxin, gx, N, 3, S, W, 1, 32, dt, st, lo_in, gsc));
    }
    return RESR_OK;
}

}  // namespace resr
