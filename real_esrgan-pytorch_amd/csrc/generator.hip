// generator.hip -- whole-RRDBNet passes enqueued natively (reference model.py:206-275 and its
// autograd backward).  No kernels here: this file owns the HBM plan and the launch order.
//
// RESR_F16X2 ("exact16"): every activation / gradient tensor below is a (hi, lo) pair of f16 tensors -- the lo tensor
// directly follows the hi tensor of the same buffer -- and packed weights take three blocks per chunk (include/resr.h).
// ResrGeneratorDesc.x2_plan (RESR_X2_PLAN_*) turns the dense blocks' growth planes (inference forward) / their gradients
// (backward) into single f16 tensors: their chunks then take two stages instead of three and conv1..conv4's weight gradients
// two tap-products instead of three; the lo halves of those planes stay allocated and unused.
// HBM plan (T = f16 fast / f32 strict, all tensors pixel-major / NHWC):
//   x_in              [N,h,w,CI]      input image, pixel-unshuffled, channels padded to 32/64
//   ws[r], r=0..3B-1  [6][N,h,w,32]   one dense-block workspace per RDB, chunk-planar: planes [x0 x1 | o1 | o2 | o3 | o4]
//                                     (a 32-channel chunk of a pixel = 64 B f16 = half an HBM line; planar, a pass
//                                     consumes whole lines -- interleaved [N,h,w,192] fetched every line twice);
//                                     conv_k reads the plane prefix [0, 2+(k-1)) and writes
//                                     its 32-channel plane, conv5 writes x of ws[r+1]  -> the four
//                                     torch.cat copies of model.py:91-94 never exist.
//                                     Inference keeps 3 rotating workspaces, training keeps all
//                                     (they are the saved activations: 192 ch/px/RDB, not 640).
//   bits[r]           [4][N,h,w] u32   training: 1-bit LeakyReLU masks of o1..o4, written by the forward convs and read by the
//                                     mirrored backward passes instead of the saved activations (64 B -> 4 B per pixel)
//   trunk_out         [2][N,h,w,32]   model.py:260 (chunk-planar, like the gT gradient ring of the backward pass)
//   feat              [2][N,h,w,32]   model.py:261-262
//   u1 [2][N,2h,2w,32], u2, c3 [2][N,4h,4w,32]   model.py:264-267 (nearest x2 folded into the conv's gather)
//   y                 [N,3,4h,4w] fp32 planar (module surface) + 1 byte/elem clamp pass-mask
// Backward-data runs the *mirrored* dense block: gradients are laid out [g_y | g_o4 | g_o3 | g_o2 | g_o1]
// (g_y: [N,h,w,64]; the slab gS = [g_o4 | g_o3 | g_o2 | g_o1] chunk-planar [4][N,h,w,32])
// so that every pass is again "3x3 conv over a channel prefix -> 32/64-channel slice" with the
// transposed/flipped weights (pack.hip) -- no read-modify-write accumulation of partial input grads.
#include <stdlib.h>
#include <vector>

#include "common.h"
#include "wgrad.h"

namespace resr {

int conv3x3_dispatch(const ResrConvDesc*, const void*, const void*, const void*, const float*, const void*,
                     const void*, const void*, void*, void*, hipStream_t);
size_t wgrad_batch_partial_bytes(const WgradConv*, int, int, int);
int conv3x3_block_dispatch(int njobs, const ResrConvDesc* d, const void* in0, const void* in1, const void* const* w,
                           const float* const* bias, const void* const* mask, void* const* out, void* const* aux,
                           const ResrConvDesc* d5, const void* w5, const float* bias5, const void* res0_5, const void* res1_5,
                           void* out5, void* chain_state, size_t chain_state_bytes, hipStream_t stream);
size_t conv3x3_chain_state_bytes(int, int, int);
int wgrad_batch(const WgradConv*, int, int, int, int, int, int, int, float*, hipStream_t);
int wgrad_batch_jobs(const WgradConv*, int, int);
int wgrad_batch_quads(const WgradConv*, int, int);
int wgrad_tile_rows(int dtype);
int wgrad_x2_products();
int nchw_to_nhwc_dispatch(const float*, void*, int, int, int, int, int, int, int, const uint8_t*, hipStream_t, long);
int nhwc_to_nchw_dispatch(const void*, float*, int, int, int, int, int, int, int, hipStream_t, long);
int absmax_dispatch(const float*, long, unsigned*, int, hipStream_t);
int nchw_to_nhwc_scaled_dispatch(const float*, void*, int, int, int, int, int, int, int, const uint8_t*, hipStream_t, long, const unsigned*);
int nchw_to_nhwc_q_dispatch(const float*, void*, int, int, int, int, int, int, int, const uint8_t*, hipStream_t, long, const unsigned*, long);
int nhwc_to_nchw_scaled_dispatch(const void*, float*, int, int, int, int, int, int, int, hipStream_t, long, const unsigned*);
int sumpool2x2_dispatch(const void*, void*, const void*, int, int, int, int, int, float, hipStream_t, long, long);
int add_inplace_dispatch(void*, const void*, long, int, hipStream_t, long, long);

namespace {

struct ConvSpec {
    int cout, cin, cin_pad, cout_pad;
    size_t w_off, b_off;      // element offsets in the fp32 parameter arena
    size_t pk_fwd, pk_bwd;    // element offsets in the packed buffer (bwd: first pass of this conv)
};

struct Plan {
    ResrGeneratorDesc d;
    int r;        // pixel-unshuffle factor
    int h, w;     // trunk resolution
    int ci_real, ci_pad;
    int nrdb;
    std::vector<ConvSpec> convs;  // conv1, trunk..., conv2, up1, up2, conv3, conv4
    size_t n_params;
    size_t pk_fwd_elems, pk_total_elems;
    // indices
    int i_conv1, i_trunk0, i_conv2, i_up1, i_up2, i_conv3, i_conv4;
    // backward packed offsets of trunk passes: [rdb][pass 0..4] (pass 0 -> g_o4 ... pass 4 -> g_x)
    std::vector<size_t> pk_bwd_trunk;
    size_t pk_bwd_conv4, pk_bwd_conv3, pk_bwd_up2, pk_bwd_up1, pk_bwd_conv2, pk_bwd_conv1;
};

int round32(int v) { return (v + 31) / 32 * 32; }

// RESR_X2_PLAN_MX_INFER: an exact16 INFERENCE forward whose pair chunks take one f16 stage + one MX stage (RESR_CONV_MX_PAIRS).  Rides
// on the single-f16 growth planes against f16 weights (bits 0 + 5): every activation buffer then holds THREE tensors -- hi, lo and the
// q tensor (bf8 of both, 2 bytes per element) -- and the packed weights an MX region behind the f16 blocks (generator_mx_offset).
// RESR_X2_PLAN_MX_BWD: the trunk's backward-data passes on MX stages; the gradient planes gT / gS carry q tensors.
bool plan_mx_bwd(const ResrGeneratorDesc& d) {
    return d.dtype == RESR_F16X2 && d.training && (d.x2_plan & RESR_X2_PLAN_MX_BWD) && !(d.x2_plan & RESR_X2_PLAN_GROWTH_GRAD_STORE_F16);
}
// RESR_X2_PLAN_MX_WGRAD: the stream chunks' correction tap-products of the dense blocks' weight gradients as MX jobs; the residual stream
// (planes 0, 1 of every dense-block workspace) carries a q tensor written by the training forward
bool plan_mx_wgrad(const ResrGeneratorDesc& d) {
    return plan_mx_bwd(d) && (d.x2_plan & RESR_X2_PLAN_MX_WGRAD) && !(d.x2_plan & RESR_X2_PLAN_F16_BACKWARD);
}
// RESR_X2_PLAN_MX_TAIL: conv3, conv4, upsampling2 the same way; u1, u2, c3 and the tail's gradient tensors g4, gA, gB carry q tensors
bool plan_mx_tail(const ResrGeneratorDesc& d) { return plan_mx_wgrad(d) && (d.x2_plan & RESR_X2_PLAN_MX_TAIL); }
bool plan_mx(const ResrGeneratorDesc& d) {
    const int need = RESR_X2_PLAN_GROWTH_F16_INFER | RESR_X2_PLAN_GROWTH_W16_INFER | RESR_X2_PLAN_MX_INFER;
    return d.dtype == RESR_F16X2 && !d.training && (d.x2_plan & need) == need;
}

bool build_plan(const ResrGeneratorDesc* d, Plan& p) {
    if (!d) return false;
    if (d->upscale != 4 && d->upscale != 2 && d->upscale != 1) return false;
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->n_blocks <= 0 || d->in_channels <= 0 || d->out_channels <= 0) return false;
    p.d = *d;
    p.r = d->upscale == 4 ? 1 : (d->upscale == 2 ? 2 : 4);
    if ((d->h % p.r) || (d->w % p.r)) return false;
    p.h = d->h / p.r;
    p.w = d->w / p.r;
    p.ci_real = d->in_channels * p.r * p.r;
    p.ci_pad = round32(p.ci_real);
    if (p.ci_pad > 64 || d->out_channels > 32) return false;
    p.nrdb = d->n_blocks * 3;
    size_t off = 0;
    auto add = [&](int cout, int cin) {
        ConvSpec c;
        c.cout = cout; c.cin = cin; c.cin_pad = round32(cin); c.cout_pad = round32(cout);
        c.w_off = off; off += (size_t)cout * cin * 9;
        c.b_off = off; off += cout;
        c.pk_fwd = c.pk_bwd = 0;
        p.convs.push_back(c);
        return (int)p.convs.size() - 1;
    };
    p.i_conv1 = add(64, p.ci_real);
    p.i_trunk0 = (int)p.convs.size();
    for (int r = 0; r < p.nrdb; ++r)
        for (int k = 1; k <= 5; ++k) add(k < 5 ? 32 : 64, 64 + 32 * (k - 1));
    p.i_conv2 = add(64, 64);
    p.i_up1 = add(64, 64);
    p.i_up2 = add(64, 64);
    p.i_conv3 = add(64, 64);
    p.i_conv4 = add(d->out_channels, 64);
    p.n_params = off;
    // packed layout: forward chunks of every conv, then backward-data passes
    size_t pk = 0;
    for (auto& c : p.convs) {
        c.pk_fwd = pk;
        pk += (size_t)(c.cin_pad / 32) * 9 * (c.cout_pad / 32) * 1024;
    }
    p.pk_fwd_elems = pk;
    auto bwd_simple = [&](int idx) {  // transposed conv: M = cin_pad (as cout_pad), K = cout_pad
        const ConvSpec& c = p.convs[idx];
        const size_t o = pk;
        pk += (size_t)(c.cout_pad / 32) * 9 * (c.cin_pad / 32) * 1024;
        return o;
    };
    p.pk_bwd_conv4 = bwd_simple(p.i_conv4);
    p.pk_bwd_conv3 = bwd_simple(p.i_conv3);
    p.pk_bwd_up2 = bwd_simple(p.i_up2);
    p.pk_bwd_up1 = bwd_simple(p.i_up1);
    p.pk_bwd_conv2 = bwd_simple(p.i_conv2);
    p.pk_bwd_conv1 = bwd_simple(p.i_conv1);
    p.pk_bwd_trunk.resize((size_t)p.nrdb * 5);
    for (int r = 0; r < p.nrdb; ++r)
        for (int ps = 0; ps < 5; ++ps) {
            // pass ps: K chunks = 2 (g_y through conv5) + ps (conv4 .. conv(5-ps)); M tiles = 1 (ps<4) or 2
            p.pk_bwd_trunk[(size_t)r * 5 + ps] = pk;
            pk += (size_t)(2 + ps) * 9 * (ps < 4 ? 1 : 2) * 1024;
        }
    p.pk_total_elems = pk;
    return true;
}

// ---------------------------------------------------------------------------------------------
// workspace carving
// ---------------------------------------------------------------------------------------------
struct Bufs {
    char* chain;              // device-side state of the chained dense-block launches (conv3x3.h ChainArgs): the HEAD of the workspace,
    size_t chain_bytes;       // zero-filled once by the caller (include/resr.h resr_generator_chain_state_bytes)
    char* x_in;
    std::vector<char*> ws;
    char *bits_u2, *bits_c3;   // training: sign words (2 per pixel) of the HR activations u2, c3
    std::vector<char*> bits;   // training: per RDB, 4 sign planes (uint32 per pixel) = the 1-bit LeakyReLU masks of o1..o4
    char* out1;       // conv1 output kept for the skip of model.py:262 (aliases ws[0] when training)
    int out1_stride;
    char *trunk_out, *feat, *u1, *u2, *c3;
    uint8_t* ymask;
    // backward
    char *g4, *gA, *gB, *gM1, *gF, *gT[4], *gS[3], *gxin;   // gS: one slab per dense block of an RRDB (their weight gradients run as one batch)
    float* partial;
    size_t partial_bytes;
    unsigned* gscale;         // exact16: bits of max |g_y| of the running backward pass (common.h: grad_prescale)
    size_t total;
};

// pixel splits of a batched weight-gradient launch (`npairs` = (X chunk, G tile) products), never fewer than two pixel
// tiles per workgroup.  strict: one workgroup per product and split, 256 CUs filled about three times over.  fast: the
// quad kernel runs one 8-wave workgroup per CU on npairs/4 jobs -- two rounds of 256, in whole groups of 8 splits
// (a split's jobs share one XCD).
// exact16 (nquads > 0: the launch's real quad-job count, wgrad_batch_quads): its job sets do not fill whole rounds the way fast
// mode's do -- a dense block with single-f16 growth gradients is 17 quads, and 17 x 32 splits = 544 workgroups is a third,
// nearly empty residency round (68 workgroups per XCD on 32 CUs).  Splits come in eights (a split's jobs share one XCD: nquads
// x s / 8 workgroups per XCD); the count minimises   rounds x (tiles / s x t_tile + t_round) + slab traffic,
// rounds = ceil(nquads x s / 8 / 32), with t_tile ~ 2.9 us per 8 x 32 tile and quad, t_round ~ 8 us of fill / drain, and every
// (job, split) slab written and read back once (36.9 KB each at ~4 TB/s).
int splits_for(const Plan& p, int npairs, int h, int w, int nquads = 0) {
    if (p.d.wgrad_splits > 0) return p.d.wgrad_splits;
    const int th = wgrad_tile_rows(p.d.dtype);
    const long tiles = (long)((w + 31) / 32) * ((h + th - 1) / th) * p.d.n;
    if (p.d.dtype == RESR_F16X2 && nquads > 0 && tiles >= 64 && !getenv("RESR_X2_WGRAD_OLD_SPLITS")) {
        long best_s = 8;
        double best = 1e30;
        for (long s = 8; s <= 256 && s <= tiles / 2; s += 8) {
            const long rounds = (nquads * (s / 8) + 31) / 32;
            const double cost = rounds * ((double)tiles / s * 2.9 + 8.0) + (double)npairs * s * 0.0185;
            if (cost < best) { best = cost; best_s = s; }
        }
        return (int)best_s;
    }
    long s;
    if (p.d.dtype != RESR_F32) {
        s = 512 / ((npairs + 3) / 4);
        if (s >= 16) s &= ~7L;
        if (s > 256) s = 256;
    } else {
        s = 768 / npairs;
        if (s > 128) s = 128;
    }
    if (s > tiles / 2) s = tiles / 2;
    if (s < 1) s = 1;
    return (int)s;
}

void carve(const Plan& p, char* base, Bufs& b) {
    const size_t es = elem_size(p.d.dtype) * (act_tensors(p.d.dtype) + (plan_mx(p.d) ? 1 : 0));   // bytes per activation element (hi + lo [+ q])
    const int wm = p.d.dtype == RESR_F16X2 ? 3 : 1;                     // weight-gradient jobs per product (upper bound: sizes the slab buffer)
    const size_t px = (size_t)p.d.n * p.h * p.w;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char* ptr = base ? base + off : nullptr;
        off += align_up(bytes, 256);
        return ptr;
    };
    b.chain_bytes = conv3x3_chain_state_bytes(p.d.n, p.h, p.w);
    b.chain = take(b.chain_bytes);
    b.gscale = (unsigned*)take(256);   // with the chain state: the zero-filled head of the workspace (its words 1, 2 are sticky: common.h)
    b.x_in = take(px * p.ci_pad * es);
    const int nws = p.d.training ? p.nrdb : 3;
    b.ws.resize(nws);
    for (int i = 0; i < nws; ++i) b.ws[i] = take(px * 192 * (es + (plan_mx_wgrad(p.d) ? 2 : 0)));   // (MX_WGRAD: hi, lo and q tensors)
    b.bits.clear();
    if (p.d.training)
        for (int i = 0; i < p.nrdb; ++i) b.bits.push_back(take(px * 4 * sizeof(uint32_t)));
    b.bits_u2 = p.d.training ? take(px * 16 * 2 * sizeof(uint32_t)) : nullptr;
    b.bits_c3 = p.d.training ? take(px * 16 * 2 * sizeof(uint32_t)) : nullptr;
    if (p.d.training) { b.out1 = b.ws[0]; b.out1_stride = 32; }  // planes 0,1 of ws[0] (chunk-planar)
    else { b.out1 = take(px * 64 * es); b.out1_stride = 64; }  // rotating workspaces overwrite ws[0]
    b.trunk_out = take(px * 64 * es);
    b.feat = take(px * 64 * es);
    const size_t est = es + (plan_mx_tail(p.d) ? 2 : 0);   // RESR_X2_PLAN_MX_TAIL: hi, lo and q tensors
    b.u1 = take(px * 4 * 64 * est);
    b.u2 = take(px * 16 * 64 * est);
    b.c3 = take(px * 16 * 64 * est);
    b.ymask = (uint8_t*)take(px * 16 * p.d.out_channels);
    if (p.d.training) {
        b.g4 = take(px * 16 * 32 * est);
        b.gA = take(px * 16 * 64 * est);
        b.gB = take(px * 16 * 64 * est);
        b.gM1 = take(px * 4 * 64 * es);
        b.gF = take(px * 64 * es);
        const size_t esg = es + (plan_mx_bwd(p.d) ? 2 : 0);   // RESR_X2_PLAN_MX_BWD: hi, lo and q tensors
        for (int i = 0; i < 4; ++i) b.gT[i] = take(px * 64 * esg);
        for (int i = 0; i < 3; ++i) b.gS[i] = take(px * 128 * esg);
        b.gxin = take(px * p.ci_pad * es);
        // wgrad slabs: largest batch (an RRDB = 78 products, a dense block = 26 at LR; single 64->64 convs = 4 jobs at 1x/2x/4x)
        const size_t slab = (9 * 1024 + 32) * sizeof(float);
        size_t pb = 0;
        if (p.d.dtype == RESR_F16X2) {   // exact16 picks its splits per launch (splits_for with the real quad count): size for any choice
            const int th = wgrad_tile_rows(p.d.dtype);
            for (int m = 1; m <= 4; m *= 2) {
                const long tiles = (long)((p.w * m + 31) / 32) * ((p.h * m + th - 1) / th) * p.d.n;
                const long smax = tiles / 2 < 256 ? (tiles / 2 < 1 ? 1 : tiles / 2) : 256;
                // (a single 64 -> 64 convolution: 12 tap-products; RESR_X2_PLAN_MX_TAIL: 4 f16 jobs + 4 MX jobs whose launch takes up to 4 x the splits)
                const size_t q = (size_t)(m == 1 ? 78 : (plan_mx_tail(p.d) ? 20 : 12)) * (size_t)smax * slab;
                if (q > pb) pb = q;
            }
        }
        for (int k = 1; k <= wm; k += 2) {          // both weight-gradient settings of RESR_F16X2 (1 or 3 jobs per product)
            const size_t q0 = (size_t)26 * k * splits_for(p, 26 * k, p.h, p.w) * slab;
            if (q0 > pb) pb = q0;
            if (k == 3) {   // RESR_X2_PLAN_GROWTH_GRAD_F16: conv1..conv4 (14 products) two tap-products each + 4 bias jobs, conv5 (12) three
                const size_t qg = (size_t)68 * splits_for(p, 68, p.h, p.w) * slab;
                if (qg > pb) pb = qg;
            }
            if (78 * k <= kWgradMaxJobs) {
                const size_t q3 = (size_t)78 * k * splits_for(p, 78 * k, p.h, p.w) * slab;
                if (q3 > pb) pb = q3;
            }
            for (int m = 1; m <= 4; m *= 2) {
                const size_t q = (size_t)4 * k * splits_for(p, 4 * k, p.h * m, p.w * m) * slab;
                if (q > pb) pb = q;
                const size_t q2 = (size_t)2 * k * splits_for(p, 2 * k, p.h * m, p.w * m) * slab;
                if (q2 > pb) pb = q2;
            }
        }
        b.partial_bytes = pb;
        b.partial = (float*)take(b.partial_bytes);
    } else {
        b.g4 = b.gA = b.gB = b.gM1 = b.gF = b.gxin = nullptr;
        for (int i = 0; i < 4; ++i) b.gT[i] = nullptr;
        for (int i = 0; i < 3; ++i) b.gS[i] = nullptr;
        b.partial = nullptr;
        b.partial_bytes = 0;
    }
    b.total = off;
}

ResrConvDesc conv_desc(const Plan& p, int n, int h, int w, int cin, int cin0, int s0, int s1, int cout, int cout_pad,
                       int out_stride, int flags) {
    ResrConvDesc c;
    memset(&c, 0, sizeof(c));
    c.n = n; c.h = h; c.w = w; c.cin = cin; c.cin0 = cin0; c.in0_stride = s0; c.in1_stride = s1;
    c.cout = cout; c.cout_pad = cout_pad; c.out_stride = out_stride; c.dtype = p.d.dtype; c.flags = flags;
    c.s0 = c.s1 = 1.f; c.t0 = c.t1 = 1.f; c.slope = 0.2f;
    return c;
}

}  // namespace

size_t generator_param_count(const ResrGeneratorDesc* d) {
    Plan p;
    return build_plan(d, p) ? p.n_params : 0;
}

// Byte offset of the MX region inside the packed buffer (RESR_F16X2): behind the f16 blocks of the forward AND backward-data tables
// whatever the pass, so that one buffer serves every use; the region mirrors a plain f16 packing (2 bytes per plain element).
size_t generator_mx_offset(const ResrGeneratorDesc* d) {
    Plan p;
    if (!build_plan(d, p) || d->dtype != RESR_F16X2) return 0;
    return align_up(p.pk_total_elems * 2 * 3 + 16384, 256);
}

size_t generator_packed_bytes(const ResrGeneratorDesc* d, int backward) {
    Plan p;
    if (!build_plan(d, p)) return 0;
    if (d->dtype == RESR_F16X2 && (d->x2_plan & (RESR_X2_PLAN_MX_INFER | RESR_X2_PLAN_MX_BWD | RESR_X2_PLAN_MX_WGRAD))) return generator_mx_offset(d) + p.pk_total_elems * 2 + 16384;
    // + two dummy (chunk,tap) of slack: conv3x3_kernel prefetches two taps past the end
    return (backward ? p.pk_total_elems : p.pk_fwd_elems) * elem_size(d->dtype) * (d->dtype == RESR_F16X2 ? 3 : 1) + 16384;
}

// the zero-filled head of a workspace: the chain state of the dense-block launches + the gradient pre-scale slot behind it
size_t generator_chain_state_bytes(const ResrGeneratorDesc* d) {
    Plan p;
    return build_plan(d, p) ? align_up(conv3x3_chain_state_bytes(p.d.n, p.h, p.w), 256) + 256 : 0;
}

size_t generator_workspace_bytes(const ResrGeneratorDesc* d) {
    Plan p;
    if (!build_plan(d, p)) return 0;
    Bufs b;
    carve(p, nullptr, b);
    return b.total;
}

// debug/test aid: byte offsets of the named workspace buffers (order documented in include/resr.h)
int64_t generator_buffer_offsets(const ResrGeneratorDesc* d, int64_t* out, int64_t cap) {
    Plan p;
    if (!build_plan(d, p)) return fail(RESR_ERR_ARG, "generator: bad descriptor");
    Bufs b;
    char* base = reinterpret_cast<char*>(4096);  // fake non-null base, only differences are used
    carve(p, base, b);
    const char* ptrs[] = {b.x_in, b.ws[0], b.out1, b.trunk_out, b.feat, b.u1, b.u2, b.c3, (char*)b.ymask, b.g4, b.gA,
                          b.gB, b.gM1, b.gF, b.gT[0], b.gT[1], b.gT[2], b.gT[3], b.gS[0], b.gxin, (char*)b.partial};
    const int64_t n = sizeof(ptrs) / sizeof(ptrs[0]);
    if (out) {
        if (cap < n) return fail(RESR_ERR_ARG, "generator_buffer_offsets: capacity");
        for (int64_t i = 0; i < n; ++i) out[i] = ptrs[i] ? (int64_t)(ptrs[i] - base) : -1;
    }
    return n;
}

int64_t generator_pack_table(const ResrGeneratorDesc* d, int backward, ResrPackChunk* out, int64_t cap) {
    Plan p;
    if (!build_plan(d, p)) return fail(RESR_ERR_ARG, "generator: bad descriptor");
    std::vector<ResrPackChunk> t;
    auto push = [&](const ConvSpec& c, size_t dst, int m_off, int m_count, int k_off, int k_count, int mt, int tr,
                    float scale) {
        ResrPackChunk ch;
        memset(&ch, 0, sizeof(ch));
        ch.src_off = (int64_t)c.w_off; ch.dst_off = (int64_t)dst;
        ch.src_cout = c.cout; ch.src_cin = c.cin;
        ch.m_off = m_off; ch.m_count = m_count; ch.k_off = k_off; ch.k_count = k_count;
        ch.mt = mt; ch.transposed = tr; ch.scale = scale;
        t.push_back(ch);
    };
    for (const auto& c : p.convs) {
        const int mt = c.cout_pad / 32;
        for (int ck = 0; ck < c.cin_pad / 32; ++ck) {
            const int kc = c.cin - ck * 32;
            push(c, c.pk_fwd + (size_t)ck * 9 * mt * 1024, 0, c.cout, ck * 32, kc > 32 ? 32 : kc, mt, 0, 1.f);
        }
    }
    if (backward) {
        auto simple = [&](int idx, size_t base) {
            const ConvSpec& c = p.convs[idx];
            const int mt = c.cin_pad / 32;
            for (int ck = 0; ck < c.cout_pad / 32; ++ck) {
                const int kc = c.cout - ck * 32;
                push(c, base + (size_t)ck * 9 * mt * 1024, 0, c.cin, ck * 32, kc > 32 ? 32 : kc, mt, 1, 1.f);
            }
        };
        simple(p.i_conv4, p.pk_bwd_conv4);
        simple(p.i_conv3, p.pk_bwd_conv3);
        simple(p.i_up2, p.pk_bwd_up2);
        simple(p.i_up1, p.pk_bwd_up1);
        simple(p.i_conv2, p.pk_bwd_conv2);
        simple(p.i_conv1, p.pk_bwd_conv1);
        for (int r = 0; r < p.nrdb; ++r) {
            // the 0.2 of model.py:95 (and, for the third RDB of a block, the 0.2 of model.py:129) folded in
            const float fold = (r % 3 == 2) ? 0.2f * 0.2f : 0.2f;
            for (int ps = 0; ps < 5; ++ps) {
                const int mt = ps < 4 ? 1 : 2;
                const int m_off = ps < 4 ? 64 + 32 * (3 - ps) : 0;  // slice of o_(4-ps), or x
                const int m_cnt = ps < 4 ? 32 : 64;
                size_t dst = p.pk_bwd_trunk[(size_t)r * 5 + ps];
                const ConvSpec& c5 = p.convs[p.i_trunk0 + r * 5 + 4];
                push(c5, dst, m_off, m_cnt, 0, 32, mt, 1, fold); dst += (size_t)9 * mt * 1024;
                push(c5, dst, m_off, m_cnt, 32, 32, mt, 1, fold); dst += (size_t)9 * mt * 1024;
                for (int j = 0; j < ps; ++j) {  // g_o4, g_o3, ... in gS channel order
                    const ConvSpec& ck = p.convs[p.i_trunk0 + r * 5 + (3 - j)];
                    push(ck, dst, m_off, m_cnt, 0, 32, mt, 1, 1.f);
                    dst += (size_t)9 * mt * 1024;
                }
            }
        }
    }
    if (out) {
        if ((int64_t)t.size() > cap) return fail(RESR_ERR_ARG, "generator_pack_table: capacity %lld < %zu", (long long)cap, t.size());
        memcpy(out, t.data(), t.size() * sizeof(ResrPackChunk));
    }
    return (int64_t)t.size();
}

static int debug_stop() {
    const char* e = getenv("RESR_DEBUG_STOP");
    return e ? atoi(e) : 0;
}

static bool debug_sync() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("RESR_DEBUG_SYNC"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
}

#define RUN(expr)                 \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != RESR_OK) return rc_; \
        if (debug_sync()) {       \
            hipError_t e_ = hipStreamSynchronize(st); \
            if (e_ != hipSuccess) return fail(RESR_ERR_LAUNCH, "%s: %s", #expr, hipGetErrorString(e_)); \
        }                         \
    } while (0)

int generator_forward(const ResrGeneratorDesc* d, const float* x, const float* params, const void* packed,
                      void* workspace, size_t workspace_bytes, float* y, hipStream_t st) {
    Plan p;
    if (!build_plan(d, p)) return fail(RESR_ERR_ARG, "generator_forward: bad descriptor");
    if (!x || !params || !packed || !workspace || !y) return fail(RESR_ERR_ARG, "generator_forward: null argument");
    Bufs b;
    carve(p, (char*)workspace, b);
    if (b.total > workspace_bytes) return fail(RESR_ERR_WORKSPACE, "generator_forward: workspace %zu < %zu", workspace_bytes, b.total);
    const size_t es = elem_size(d->dtype);
    const bool x2 = d->dtype == RESR_F16X2;
    const size_t wes = es * (x2 ? 3 : 1);   // bytes per element of the plain packed layout
    const char* pk = (const char*)packed;
    const int N = d->n, h = p.h, w = p.w;
    const int nws = (int)b.ws.size();
    if ((long)N * h * w * 32 * 16 > 0x7fffffffL) return fail(RESR_ERR_ARG, "generator: batch x resolution too large for 32-bit chunk strides");
    const int plane = N * h * w * 32;  // elements per 32-channel plane of the chunk-planar trunk tensors
    // exact16 inference with single-f16 growth planes (RESR_X2_PLAN_GROWTH_F16_INFER): o1..o4 are stored without a lo tensor and
    // read as two-stage chunks; a training forward keeps every pre-activation fp32-class (a rounded input flips LeakyReLU masks)
    const bool growth_single = x2 && !d->training && (d->x2_plan & RESR_X2_PLAN_GROWTH_F16_INFER);
    const bool growth_w16 = growth_single && (d->x2_plan & RESR_X2_PLAN_GROWTH_W16_INFER);   // ... and meet f16 weights: one stage per growth chunk
    // RESR_F16X2: element offset hi -> lo of a buffer holding `planes` 32-channel planes of `pl` elements
    auto LO = [&](long planes, long pl) -> int64_t { return x2 ? planes * pl : 0; };
    const int64_t lo_ws = LO(6, plane), lo_t = LO(2, plane), lo_xin = x2 ? (int64_t)N * h * w * p.ci_pad : 0;
    const int64_t lo_out1 = b.out1 == b.ws[0] ? lo_ws : LO(2, plane);
    auto W = [&](const ConvSpec& c) { return pk + c.pk_fwd * wes; };
    auto Bias = [&](const ConvSpec& c) { return params + c.b_off; };
    // RESR_X2_PLAN_MX_INFER: the q tensor of a buffer follows its lo tensor (element offset hi -> q = twice hi -> lo); a pass whose
    // pair chunks take the MX stage says so (in_q = its input's hi -> q offset), names its convolution's MX blocks, and emits the q
    // tensor of its own output when a later pass reads that output through an MX stage (out_q; 0: none)
    const bool mx = plan_mx(*d);
    const bool mxw = plan_mx_wgrad(*d);
    const bool mxt = plan_mx_tail(*d);     // the training forward also emits the q tensors of u1, u2, c3
    const char* pk_mx = mx ? pk + generator_mx_offset(d) : nullptr;
    auto MXP = [&](ResrConvDesc& cd, const ConvSpec& c, int64_t in_q, int64_t out_q) {
        if (!mx) return;
        cd.flags |= RESR_CONV_MX_PAIRS;
        cd.in0_q_offset = in_q; cd.out_q_offset = out_q;
        cd.w_mx_offset = (int64_t)((pk_mx + c.pk_fwd * 2) - W(c));
    };

    RUN(nchw_to_nhwc_q_dispatch(x, b.x_in, N, d->in_channels, d->h, d->w, p.r, p.ci_pad, d->dtype, nullptr, st, (long)lo_xin, nullptr, mx ? 2L * (long)lo_xin : 0L));
    {   // conv1 -> ws[0][0:64]                                           model.py:259
        const ConvSpec& c = p.convs[p.i_conv1];
        ResrConvDesc cd = conv_desc(p, N, h, w, p.ci_pad, p.ci_pad, p.ci_pad, 0, 64, 64, 32, 0);
        cd.out_chunk_stride = plane;
        cd.in0_lo_offset = lo_xin; cd.out_lo_offset = lo_ws;
        MXP(cd, c, 2 * lo_xin, 2 * lo_ws);
        if (mxw) cd.out_q_offset = 2 * lo_ws;    // training, MX_WGRAD: the stream's q tensor for the weight gradients' MX jobs (a plain three-stage pass emits it)
        RUN(conv3x3_dispatch(&cd, b.x_in, nullptr, W(c), Bias(c), nullptr, nullptr, nullptr, b.ws[0], nullptr, st));
        if (b.out1 != b.ws[0]) {  // inference: second copy of out1 (0.01 % of the FLOPs) instead of a pinned workspace
            cd.out_stride = b.out1_stride; cd.out_chunk_stride = 0; cd.out_lo_offset = lo_out1;
            cd.out_q_offset = 0;   // (read as a residual only)
            RUN(conv3x3_dispatch(&cd, b.x_in, nullptr, W(c), Bias(c), nullptr, nullptr, nullptr, b.out1, nullptr, st));
        }
    }
    for (int r = 0; r < p.nrdb; ++r) {
        char* cur = b.ws[r % nws];
        {   // conv1..conv4 (model.py:90-93): one chained launch where the kernel supports it, else four
            ResrConvDesc cds[4];
            const void* ws4[4];
            const float* bs4[4];
            void* outs4[4];
            void* signs4[4];
            for (int k = 1; k <= 4; ++k) {
                const ConvSpec& c = p.convs[p.i_trunk0 + r * 5 + k - 1];
                ResrConvDesc cd = conv_desc(p, N, h, w, c.cin, c.cin, 32, 0, 32, 32, 32, RESR_CONV_LRELU);
                cd.in0_chunk_stride = plane;
                cd.in0_lo_offset = lo_ws; cd.out_lo_offset = lo_ws;
                if (growth_single) { cd.x2_pair_chunks = 2; cd.flags |= RESR_CONV_OUT_SINGLE | (growth_w16 ? RESR_CONV_SINGLE_W16 : 0); }
                MXP(cd, c, 2 * lo_ws, 0);
                char* signs = nullptr;
                if (d->training) {   // the backward pass reads the 1-bit mask, not the activation
                    cd.flags |= RESR_CONV_WRITE_SIGNBITS;
                    signs = b.bits[r] + (size_t)(k - 1) * N * h * w * sizeof(uint32_t);
                }
                cds[k - 1] = cd; ws4[k - 1] = W(c); bs4[k - 1] = Bias(c);
                outs4[k - 1] = cur + (size_t)(2 + (k - 1)) * plane * es; signs4[k - 1] = signs;
            }
            // conv5 (model.py:94-96) goes with them: one chained launch of six jobs on small launches, else its own launch
            const ConvSpec& c = p.convs[p.i_trunk0 + r * 5 + 4];
            const bool last = r == p.nrdb - 1;
            char* dst = last ? b.trunk_out : b.ws[(r + 1) % nws];
            ResrConvDesc cd = conv_desc(p, N, h, w, 192, 192, 32, 0, 64, 64, 32, 0);
            cd.in0_chunk_stride = plane;
            cd.out_chunk_stride = plane;
            cd.in0_lo_offset = lo_ws; cd.out_lo_offset = last ? lo_t : lo_ws;
            if (growth_single) { cd.x2_pair_chunks = 2; if (growth_w16) cd.flags |= RESR_CONV_SINGLE_W16; }
            MXP(cd, c, 2 * lo_ws, last ? 2 * lo_t : 2 * lo_ws);
            if (mxw && !last) cd.out_q_offset = 2 * lo_ws;   // (the next block's stream planes; the trunk's output is no dense block's X)
            cd.s0 = 0.2f; cd.t0 = 1.f; cd.res0_stride = 32; cd.res0_chunk_stride = plane; cd.res0_lo_offset = lo_ws;  // model.py:95-96
            const char* res1 = nullptr;
            if (r % 3 == 2) {  // model.py:129-130
                res1 = b.ws[(r - 2) % nws];
                cd.s1 = 0.2f; cd.t1 = 1.f; cd.res1_stride = 32; cd.res1_chunk_stride = plane; cd.res1_lo_offset = lo_ws;
            }
            RUN(conv3x3_block_dispatch(4, cds, cur, nullptr, ws4, bs4, nullptr, outs4, signs4, &cd, W(c), Bias(c), cur, res1, dst, b.chain, b.chain_bytes, st));
        }
    }
    {   // conv2 + skip                                                   model.py:261-262
        const ConvSpec& c = p.convs[p.i_conv2];
        ResrConvDesc cd = conv_desc(p, N, h, w, 64, 64, 32, 0, 64, 64, 64, 0);
        cd.in0_chunk_stride = plane;
        cd.res0_stride = b.out1_stride;
        cd.res0_chunk_stride = b.out1 == b.ws[0] ? plane : 0;
        cd.out_stride = 32; cd.out_chunk_stride = plane;        // feat and the whole HR tail are chunk-planar [2][N,H,W,32] too
        cd.in0_lo_offset = lo_t; cd.res0_lo_offset = lo_out1; cd.out_lo_offset = lo_t;
        MXP(cd, c, 2 * lo_t, 2 * lo_t);
        RUN(conv3x3_dispatch(&cd, b.trunk_out, nullptr, W(c), Bias(c), b.out1, nullptr, nullptr, b.feat, nullptr, st));
    }
    {   // model.py:264
        const ConvSpec& c = p.convs[p.i_up1];
        ResrConvDesc cd = conv_desc(p, N, 2 * h, 2 * w, 64, 64, 32, 0, 64, 64, 32, RESR_CONV_LRELU | RESR_CONV_UPSAMPLE_IN);
        cd.in0_chunk_stride = plane; cd.out_chunk_stride = 4 * plane;
        cd.in0_lo_offset = lo_t; cd.out_lo_offset = LO(2, 4L * plane);
        MXP(cd, c, 2 * lo_t, 2 * LO(2, 4L * plane));
        if (mxt) cd.out_q_offset = 2 * LO(2, 4L * plane);
        RUN(conv3x3_dispatch(&cd, b.feat, nullptr, W(c), Bias(c), nullptr, nullptr, nullptr, b.u1, nullptr, st));
    }
    {   // model.py:265
        const ConvSpec& c = p.convs[p.i_up2];
        ResrConvDesc cd = conv_desc(p, N, 4 * h, 4 * w, 64, 64, 32, 0, 64, 64, 32, RESR_CONV_LRELU | RESR_CONV_UPSAMPLE_IN);
        cd.in0_chunk_stride = 4 * plane; cd.out_chunk_stride = 16 * plane;
        cd.in0_lo_offset = LO(2, 4L * plane); cd.out_lo_offset = LO(2, 16L * plane);
        if (d->training) cd.flags |= RESR_CONV_WRITE_SIGNBITS;
        MXP(cd, c, 2 * LO(2, 4L * plane), 2 * LO(2, 16L * plane));
        if (mxt) cd.out_q_offset = 2 * LO(2, 16L * plane);
        RUN(conv3x3_dispatch(&cd, b.u1, nullptr, W(c), Bias(c), nullptr, nullptr, nullptr, b.u2, b.bits_u2, st));
    }
    {   // model.py:267
        const ConvSpec& c = p.convs[p.i_conv3];
        ResrConvDesc cd = conv_desc(p, N, 4 * h, 4 * w, 64, 64, 32, 0, 64, 64, 32, RESR_CONV_LRELU);
        cd.in0_chunk_stride = 16 * plane; cd.out_chunk_stride = 16 * plane;
        cd.in0_lo_offset = LO(2, 16L * plane); cd.out_lo_offset = LO(2, 16L * plane);
        if (d->training) cd.flags |= RESR_CONV_WRITE_SIGNBITS;
        MXP(cd, c, 2 * LO(2, 16L * plane), 0);   // (conv4 -- 0.15 % of the FLOPs, the fp32 NCHW epilogue -- keeps its three f16 stages: no q tensor of c3)
        if (mxt) cd.out_q_offset = 2 * LO(2, 16L * plane);
        RUN(conv3x3_dispatch(&cd, b.u2, nullptr, W(c), Bias(c), nullptr, nullptr, nullptr, b.c3, b.bits_c3, st));
    }
    {   // model.py:268-270
        const ConvSpec& c = p.convs[p.i_conv4];
        ResrConvDesc cd = conv_desc(p, N, 4 * h, 4 * w, 64, 64, 32, 0, c.cout, c.cout_pad, 0,
                                    RESR_CONV_CLAMP01 | RESR_CONV_OUT_NCHW_F32);
        cd.in0_chunk_stride = 16 * plane;
        cd.in0_lo_offset = LO(2, 16L * plane);
        RUN(conv3x3_dispatch(&cd, b.c3, nullptr, W(c), Bias(c), nullptr, nullptr, nullptr, y, b.ymask, st));
    }
    return RESR_OK;
}

int generator_backward(const ResrGeneratorDesc* d, const float* gy, const float* params, const void* packed,
                       void* workspace, size_t workspace_bytes, float* grad, float* gx, hipStream_t st,
                       void* const* events, int n_events) {
    (void)params;
    Plan p;
    if (!build_plan(d, p)) return fail(RESR_ERR_ARG, "generator_backward: bad descriptor");
    if (n_events != 0 && (!events || n_events != d->n_blocks + 2))
        return fail(RESR_ERR_ARG, "generator_backward: grad_ready_events needs n_blocks + 2 = %d events, got %d", d->n_blocks + 2, n_events);
    // a range of the gradient arena is final once its weight-gradient reductions are enqueued: tell the caller's comm stream
    auto ready = [&](int i) -> int {
        if (n_events == 0) return RESR_OK;
        if (!events[i]) return fail(RESR_ERR_ARG, "generator_backward: null event %d", i);
        if (hipEventRecord((hipEvent_t)events[i], st) != hipSuccess) return fail(RESR_ERR_LAUNCH, "generator_backward: hipEventRecord");
        return RESR_OK;
    };
    if (!d->training) return fail(RESR_ERR_ARG, "generator_backward: forward was not run with training=1");
    if (!gy || !packed || !workspace || !grad) return fail(RESR_ERR_ARG, "generator_backward: null argument");
    Bufs b;
    carve(p, (char*)workspace, b);
    if (b.total > workspace_bytes) return fail(RESR_ERR_WORKSPACE, "generator_backward: workspace %zu < %zu", workspace_bytes, b.total);
    // RESR_X2_PLAN_F16_BACKWARD (opt-in): an exact16 FORWARD (every pre-activation fp32-class: reference-exact LeakyReLU masks, losses at
    // 2e-6) followed by fast mode's backward pass -- plain f16 on the hi tensors of the saved activations, the sign words the forward
    // wrote, single f16 gradient planes, f16 packed weights (`packed` is then a RESR_F16 packing of the same table).  The workspace is
    // the exact16 one (carve below keeps the caller's descriptor); every descriptor of the pass says `dt`.
    const bool f16bwd = d->dtype == RESR_F16X2 && (d->x2_plan & RESR_X2_PLAN_F16_BACKWARD);
    const int dt = f16bwd ? (int)RESR_F16 : d->dtype;
    Plan pb = p;
    pb.d.dtype = dt;
    const size_t es = elem_size(dt);
    const bool x2 = dt == RESR_F16X2;
    const size_t wes = es * (x2 ? 3 : 1);
    const int wm = x2 ? wgrad_x2_products() : 1;
    // exact16, RESR_X2_PLAN_GROWTH_GRAD_F16: the growth-plane gradients g_o1..g_o4 are READ as single f16 tensors -- two stages on
    // their chunks in every backward-data pass, two tap-products in conv1..conv4's weight gradients.  They are still STORED as
    // pairs: the bias gradient (a plain sum of G, which cancels where the weight products do not) takes hi + lo through the one
    // (x_hi chunk 0, g_lo) job per convolution that carries the bias sum -- with a single-f16 G the worst bias tensor of the
    // emulation reached 6.7e-4 at 1 x 128^2.
    // The 16-bit modes lift a small incoming gradient into f16's normal range: when max |g_y| < 2^6 the pass runs on g_y * 2^k with
    // max |g_y * 2^k| in [2^6, 2^7) and hands every result out times 2^-k (both exact; common.h grad_prescale).  The pass is linear in
    // g_y, and its f16 tensors -- the hi halves that the plan below reads alone most of all -- keep their 11 bits whatever loss scale
    // the caller works at.  Measured without it (tools/x2_plan_validate.py, L1 mean loss at 16 x 256^2: g_y = scale x 2e-8 per
    // element): worst gradient tensor of the default plan against the all-pairs plan 0.82 at loss scale 2^10, 3.3e-3 at a GradScaler's
    // initial 2^16, 1.2e-5 at 2^20 -- hi halves below 6e-5 are f16 subnormals.  The target leaves 2^9 of headroom to f16's maximum for
    // gradients that grow on their way back; $RESR_X2_GRAD_PRESCALE_LOG2 moves it, RESR_X2_NO_GRAD_PRESCALE=1 turns the lift off.
    const char* no_prescale = getenv("RESR_X2_NO_GRAD_PRESCALE");   // (read per call: A/B knobs)
    const char* pre_t = getenv("RESR_X2_GRAD_PRESCALE_LOG2");
    const int pre_log2 = pre_t ? atoi(pre_t) : 6;
    // fast mode (plain f16) takes the same lift: under the L1 loss at 16 x 256^2 its worst gradient tensor against exact16's all-pairs plan
    // reads 4.3e-3 at a GradScaler's initial 2^16 and 1.4e-3 from 2^20 on (median 7.7e-4 throughout; garbage at 2^10) -- the difference is
    // f16 underflow, not f16 arithmetic (tools/fast_loss_scale_probe.py).  strict (f32) needs none.
    const unsigned* gsc = (dt != RESR_F32 && !no_prescale) ? b.gscale : nullptr;
    // RESR_X2_PLAN_MX_BWD: the dense blocks' backward-data passes read every gradient chunk as a pair on an f16 + an MX stage
    const bool mxb = plan_mx_bwd(*d) && !f16bwd;
    const bool mxw = plan_mx_wgrad(*d) && mxb;
    const bool mxt = plan_mx_tail(*d) && mxw;
    const bool gg_single = x2 && (d->x2_plan & RESR_X2_PLAN_GROWTH_GRAD_F16);
    const bool gg_store_single = gg_single && (d->x2_plan & RESR_X2_PLAN_GROWTH_GRAD_STORE_F16);   // opt-in: no lo store, biases from hi alone
    // RESR_X2_PLAN_GROWTH_ACT_F16_WGRAD: the weight products of conv2..conv5 read the growth planes (X chunks 2..) as their hi tensor
    const int wx_pairs = (x2 && (d->x2_plan & RESR_X2_PLAN_GROWTH_ACT_F16_WGRAD)) ? 2 : 0;
    const char* pk = (const char*)packed;
    const int N = d->n, h = p.h, w = p.w;
    const int H4 = 4 * h, W4 = 4 * w, H2 = 2 * h, W2 = 2 * w;
    if ((long)N * h * w * 32 * 16 > 0x7fffffffL) return fail(RESR_ERR_ARG, "generator: batch x resolution too large for 32-bit chunk strides");
    const int plane = N * h * w * 32;  // elements per 32-channel plane of the chunk-planar trunk tensors (ws[], gS)
    const int pl2 = 4 * plane, pl4 = 16 * plane;   // planes of the 2x / 4x resolution tensors
    // RESR_F16X2: element offsets hi -> lo (the lo tensor follows the hi tensor of the same buffer)
    auto LO = [&](long planes, long pl) -> long { return x2 ? planes * pl : 0; };
    const long lo_ws = LO(6, plane), lo_t = LO(2, plane), lo_gs = LO(4, plane), lo_2 = LO(2, pl2), lo_4 = LO(2, pl4), lo_g4 = LO(1, pl4);
    const long lo_xin = x2 ? (long)N * h * w * p.ci_pad : 0;

    auto wconv = [&](const ConvSpec& c, const void* x0, int cin, int s0, const void* g, int gstride, float scale,
                     long x_lo, long g_lo) {
        WgradConv wc;
        wc.x0 = x0; wc.cin = cin; wc.in0_stride = s0; wc.cin_real = c.cin;
        wc.g = g; wc.cout = c.cout; wc.cout_pad = c.cout_pad; wc.g_stride = gstride;
        wc.x_chunk_stride = wc.g_chunk_stride = 0;
        wc.x_lo_off = x_lo; wc.g_lo_off = g_lo; wc.x_s2d_c = 0; wc.g_lo_bias_only = 0;
        wc.dw = grad + c.w_off; wc.db = grad + c.b_off; wc.scale = scale;
        wc.unscale = gsc;
        return wc;
    };
    auto wgrad_run = [&](const WgradConv* wc, int nconv, int hh, int ww, int flags) -> int {
        const int njobs = wgrad_batch_jobs(wc, nconv, dt);   // tap-products: fewer than wm per product where G or an X chunk is read single
        const int splits = splits_for(pb, njobs, hh, ww, x2 ? wgrad_batch_quads(wc, nconv, dt) : 0);
        if (wgrad_batch_partial_bytes(wc, nconv, splits, dt) > b.partial_bytes)
            return fail(RESR_ERR_WORKSPACE, "wgrad slab buffer too small");
        return wgrad_batch(wc, nconv, N, hh, ww, dt, flags, splits, b.partial, st);
    };
    auto wgrad = [&](const ConvSpec& c, int hh, int ww, const void* x0, int cin, int s0, const void* g, int gstride,
                     int flags, float scale, long x_chunk, long g_chunk, long x_lo, long g_lo, long x_q = 0, long g_q = 0) -> int {
        WgradConv wc = wconv(c, x0, cin, s0, g, gstride, scale, x_lo, g_lo);
        wc.x_chunk_stride = x_chunk; wc.g_chunk_stride = g_chunk;
        wc.x_q_off = x_q; wc.g_q_off = g_q;     // both != 0 (RESR_X2_PLAN_MX_TAIL): the correction tap-products as MX jobs
        return wgrad_run(&wc, 1, hh, ww, flags);
    };
    // backward-data pass descriptor: in0 (cin0 channels, lo offset lo0) [+ in1 (lo offset lo1)] -> out (lo offset lo_out)
    auto dgrad = [&](int hh, int ww, int cin0, int s0, int cin, int s1, int cout, int cout_pad, int out_stride, int flags,
                     long lo0, long lo1, long lo_out) {
        ResrConvDesc cd = conv_desc(pb, N, hh, ww, cin, cin0, s0, s1, cout, cout_pad, out_stride, flags | RESR_CONV_NO_BIAS);
        cd.in0_lo_offset = lo0; cd.in1_lo_offset = lo1; cd.out_lo_offset = lo_out;
        return cd;
    };

    // clamp_ backward + layout                                              model.py:270
    if (gsc) RUN(absmax_dispatch(gy, (long)N * d->out_channels * H4 * W4, b.gscale, pre_log2, st));
    RUN(nchw_to_nhwc_q_dispatch(gy, b.g4, N, d->out_channels, H4, W4, 1, 32, dt, b.ymask, st, lo_g4, gsc, mxt ? 2 * lo_g4 : 0L));
    const char* pk_mx = mxb ? pk + generator_mx_offset(d) : nullptr;
    auto MXT = [&](ResrConvDesc& cd, size_t pk_off, long in_q, long out_q) {   // a tail pass on one f16 + one MX stage per chunk of its gradient input
        if (!mxt) return;
        cd.flags |= RESR_CONV_MX_PAIRS;
        cd.in0_q_offset = in_q; cd.out_q_offset = out_q;
        cd.w_mx_offset = (int64_t)((pk_mx + pk_off * 2) - (pk + pk_off * wes));
    };
    {   // conv4                                                            model.py:268
        const ConvSpec& c = p.convs[p.i_conv4];
        RUN(wgrad(c, H4, W4, b.c3, 64, 32, b.g4, 32, 0, 1.f, pl4, 0, lo_4, lo_g4, mxt ? 2 * lo_4 : 0, mxt ? 2 * lo_g4 : 0));
        ResrConvDesc cd = dgrad(H4, W4, 32, 32, 32, 0, 64, 64, 32, RESR_CONV_MASK | RESR_CONV_MASK_BITS, lo_g4, 0, lo_4);
        cd.out_chunk_stride = pl4;
        MXT(cd, p.pk_bwd_conv4, 2 * lo_g4, 2 * lo_4);
        RUN(conv3x3_dispatch(&cd, b.g4, nullptr, pk + p.pk_bwd_conv4 * wes, nullptr, nullptr, nullptr, b.bits_c3, b.gA, nullptr, st));
    }
    {   // conv3                                                            model.py:267
        const ConvSpec& c = p.convs[p.i_conv3];
        RUN(wgrad(c, H4, W4, b.u2, 64, 32, b.gA, 32, 0, 1.f, pl4, pl4, lo_4, lo_4, mxt ? 2 * lo_4 : 0, mxt ? 2 * lo_4 : 0));
        ResrConvDesc cd = dgrad(H4, W4, 64, 32, 64, 0, 64, 64, 32, RESR_CONV_MASK | RESR_CONV_MASK_BITS, lo_4, 0, lo_4);
        cd.in0_chunk_stride = pl4; cd.out_chunk_stride = pl4;
        MXT(cd, p.pk_bwd_conv3, 2 * lo_4, 2 * lo_4);
        RUN(conv3x3_dispatch(&cd, b.gA, nullptr, pk + p.pk_bwd_conv3 * wes, nullptr, nullptr, nullptr, b.bits_u2, b.gB, nullptr, st));
    }
    if (debug_stop() == 1) return RESR_OK;
    {   // upsampling2                                                      model.py:265
        const ConvSpec& c = p.convs[p.i_up2];
        RUN(wgrad(c, H4, W4, b.u1, 64, 32, b.gB, 32, RESR_CONV_UPSAMPLE_IN, 1.f, pl2, pl4, lo_2, lo_4, mxt ? 2 * lo_2 : 0, mxt ? 2 * lo_4 : 0));
        ResrConvDesc cd = dgrad(H4, W4, 64, 32, 64, 0, 64, 64, 32, 0, lo_4, 0, lo_4);
        cd.in0_chunk_stride = pl4; cd.out_chunk_stride = pl4;
        MXT(cd, p.pk_bwd_up2, 2 * lo_4, 0);     // (its output feeds the sum-pool: no q tensor)
        RUN(conv3x3_dispatch(&cd, b.gB, nullptr, pk + p.pk_bwd_up2 * wes, nullptr, nullptr, nullptr, nullptr, b.gA, nullptr, st));
        for (int q = 0; q < 2; ++q)   // per 32-channel plane
            RUN(sumpool2x2_dispatch(b.gA + (size_t)q * pl4 * es, b.gM1 + (size_t)q * pl2 * es, b.u1 + (size_t)q * pl2 * es,
                                    N, H2, W2, 32, dt, 0.2f, st, lo_4, lo_2));
    }
    if (debug_stop() == 2) return RESR_OK;
    {   // upsampling1                                                      model.py:264
        const ConvSpec& c = p.convs[p.i_up1];
        RUN(wgrad(c, H2, W2, b.feat, 64, 32, b.gM1, 32, RESR_CONV_UPSAMPLE_IN, 1.f, plane, pl2, lo_t, lo_2));
        ResrConvDesc cd = dgrad(H2, W2, 64, 32, 64, 0, 64, 64, 32, 0, lo_2, 0, lo_2);
        cd.in0_chunk_stride = pl2; cd.out_chunk_stride = pl2;
        RUN(conv3x3_dispatch(&cd, b.gM1, nullptr, pk + p.pk_bwd_up1 * wes, nullptr, nullptr, nullptr, nullptr, b.gA, nullptr, st));
        for (int q = 0; q < 2; ++q)
            RUN(sumpool2x2_dispatch(b.gA + (size_t)q * pl2 * es, b.gF + (size_t)q * plane * es, nullptr, N, h, w, 32, dt, 0.2f, st,
                                    lo_2, lo_t));
    }
    if (debug_stop() == 3) return RESR_OK;
    int cur = 0;  // index into gT ring of the gradient wrt the current RDB's output chain
    {   // conv2                                                            model.py:261
        const ConvSpec& c = p.convs[p.i_conv2];
        RUN(wgrad(c, h, w, b.trunk_out, 64, 32, b.gF, 32, 0, 1.f, plane, plane, lo_t, lo_t));
        ResrConvDesc cd = dgrad(h, w, 64, 32, 64, 0, 64, 64, 32, 0, lo_t, 0, lo_t);
        cd.in0_chunk_stride = plane; cd.out_chunk_stride = plane;   // the gT ring (gradient wrt the RDB chain) is chunk-planar [2][N,h,w,32]
        if (mxb) cd.out_q_offset = 2 * lo_t;   // the first dense block's passes read gT[0] through MX stages: this (plain) pass emits its q tensor
        RUN(conv3x3_dispatch(&cd, b.gF, nullptr, pk + p.pk_bwd_conv2 * wes, nullptr, nullptr, nullptr, nullptr, b.gT[0], nullptr, st));
    }
    auto MXB = [&](ResrConvDesc& cd, size_t pk_off, long out_q) {   // gin (in0) and the slab gS (in1) with their q tensors, the pass's MX blocks
        if (!mxb) return;
        cd.flags |= RESR_CONV_MX_PAIRS;
        cd.x2_pair_chunks = 0;   // every chunk a pair: the growth-plane gradients enter with both halves
        cd.in0_q_offset = 2 * lo_t; cd.in1_q_offset = 2 * lo_gs; cd.out_q_offset = out_q;
        cd.w_mx_offset = (int64_t)((pk_mx + pk_off * 2) - (pk + pk_off * wes));
    };
    RUN(ready(0));   // conv4, conv3, upsampling2, upsampling1, conv2: the tail of the arena
    // trunk, mirrored dense blocks.  gT ring: e (grad wrt RRDB output) must survive its three RDBs.
    int e_idx = 0;
    // The weight gradients of the three dense blocks of an RRDB run as ONE batched launch pair behind the block's last
    // backward-data pass (15 convolutions, 78 products): a third of the slab writes and reductions of one launch pair per
    // dense block -- on small launches (the 64^2 training crops) the slabs are a third of the weight-gradient time.  Every
    // block of the RRDB therefore keeps its own gradient slab gS[pos]; the gT ring already keeps the three input gradients
    // (e, a, b) alive until the RRDB is done.  exact16 with three products per weight (234 jobs) stays per block.
    const bool batch_rrdb = 78 * wm <= kWgradMaxJobs && !getenv("RESR_WGRAD_PER_BLOCK");
    WgradConv wc[15];
    for (int r = p.nrdb - 1; r >= 0; --r) {
        const int pos = r % 3;  // 2: rdb3 (first in backward), 0: rdb1 (last)
        if (pos == 2) e_idx = cur;
        const char* gin = b.gT[cur];
        const char* act = b.ws[r];
        char* gS = b.gS[pos];
        const float fold = pos == 2 ? 0.04f : 0.2f;
        WgradConv* wcb = wc + (batch_rrdb ? 5 * pos : 0);
        wcb[4] = wconv(p.convs[p.i_trunk0 + r * 5 + 4], act, 192, 32, gin, 32, fold, lo_ws, lo_t);   // conv5: G = fold * gin
        wcb[4].x_chunk_stride = plane; wcb[4].g_chunk_stride = plane;
        wcb[4].x_pair_chunks = wx_pairs;
        wcb[4].x_single_g_hi = (wx_pairs && (d->x2_plan & RESR_X2_PLAN_GROWTH_ACT_G_HI_WGRAD)) ? 1 : 0;
        if (mxw) { wcb[4].x_q_off = 2 * lo_ws; wcb[4].g_q_off = 2 * lo_t; }   // MX jobs for the stream chunks' corrections (X: ws planes 0, 1; G: gin)
        ResrConvDesc cds[4];
        const void* ws4[4];
        const void* masks4[4];
        void* outs4[4];
        for (int ps = 0; ps < 4; ++ps) {   // g_o4, g_o3, g_o2, g_o1
            const int k = 4 - ps;           // conv index whose pre-activation gradient this pass yields
            const int cin = 64 + 32 * ps;
            ResrConvDesc cd = dgrad(h, w, 64, 32, cin, 32, 32, 32, 32, RESR_CONV_MASK | RESR_CONV_MASK_BITS, lo_t, lo_gs, lo_gs);
            cd.in0_chunk_stride = plane; cd.in1_chunk_stride = plane;
            if (gg_single) cd.x2_pair_chunks = 2;   // g_y (in0): pairs; the slab (in1) is read as single f16 chunks, written as a pair
            if (gg_store_single) cd.flags |= RESR_CONV_OUT_SINGLE;
            MXB(cd, p.pk_bwd_trunk[(size_t)r * 5 + ps], 2 * lo_gs);
            char* out = gS + (size_t)ps * plane * es;
            const char* mask = b.bits[r] + (size_t)(k - 1) * N * h * w * sizeof(uint32_t);   // sign plane of o_k
            cds[ps] = cd; ws4[ps] = pk + p.pk_bwd_trunk[(size_t)r * 5 + ps] * wes; masks4[ps] = mask; outs4[ps] = out;
            const ConvSpec& c = p.convs[p.i_trunk0 + r * 5 + k - 1];
            wcb[k - 1] = wconv(c, act, c.cin, 32, out, 32, 1.f, lo_ws, gg_store_single ? 0 : lo_gs);
            wcb[k - 1].x_chunk_stride = plane;
            wcb[k - 1].g_lo_bias_only = (gg_single && !gg_store_single) ? 1 : 0;
            wcb[k - 1].x_pair_chunks = wx_pairs;
            if (mxw) { wcb[k - 1].x_q_off = 2 * lo_ws; wcb[k - 1].g_q_off = 2 * lo_gs; }   // (the stream chunks' MX job carries (x_hi, g_lo) too; the growth chunks keep the plan's reads)
        }
        {   // the four mirrored cout-32 passes, then g_x = convT(all) + (skip terms): one chained launch where the kernel supports
            // it (g_x joins on small launches), else one launch per pass
            int nxt = (cur + 1) & 3;
            if (nxt == e_idx && pos != 2) nxt = (nxt + 1) & 3;
            ResrConvDesc cd = dgrad(h, w, 64, 32, 192, 32, 64, 64, 32, 0, lo_t, lo_gs, lo_t);
            cd.in0_chunk_stride = plane; cd.in1_chunk_stride = plane; cd.out_chunk_stride = plane;
            if (gg_single) cd.x2_pair_chunks = 2;
            const char* res0 = gin;
            const char* res1 = nullptr;
            cd.res0_stride = 32; cd.res0_chunk_stride = plane; cd.s0 = 1.f; cd.res0_lo_offset = lo_t;
            cd.t0 = pos == 2 ? 0.2f : 1.f;       // d(rdb3_out*0.2 + x)/d(rdb3_out) reaches x3 scaled
            if (pos == 0) { res1 = b.gT[e_idx]; cd.res1_stride = 32; cd.res1_chunk_stride = plane; cd.s1 = 1.f; cd.t1 = 1.f; cd.res1_lo_offset = lo_t; }
            MXB(cd, p.pk_bwd_trunk[(size_t)r * 5 + 4], 2 * lo_t);
            RUN(conv3x3_block_dispatch(4, cds, gin, gS, ws4, nullptr, masks4, outs4, nullptr, &cd,
                                       pk + p.pk_bwd_trunk[(size_t)r * 5 + 4] * wes, nullptr, res0, res1, b.gT[nxt], b.chain, b.chain_bytes, st));
            cur = nxt;
        }
        if (!batch_rrdb) RUN(wgrad_run(wc, 5, h, w, 0));   // all five weight gradients of the block in one launch pair (they read gin and gS, not g_x)
        else if (pos == 0) RUN(wgrad_run(wc, 15, h, w, 0));   // ... of the RRDB's three blocks
        if (pos == 0) RUN(ready(1 + (d->n_blocks - 1 - r / 3)));   // rdb3, rdb2, rdb1 of this RRDB are done
    }
    if (debug_stop() == 4) return RESR_OK;
    // gradient wrt out1 = trunk path + skip (model.py:262)
    RUN(add_inplace_dispatch(b.gT[cur], b.gF, (long)N * h * w * 64, dt, st, lo_t, lo_t));   // both chunk-planar [2][N,h,w,32]
    {   // conv1                                                            model.py:259
        const ConvSpec& c = p.convs[p.i_conv1];
        RUN(wgrad(c, h, w, b.x_in, p.ci_pad, p.ci_pad, b.gT[cur], 32, 0, 1.f, 0, plane, lo_xin, lo_t));
        RUN(ready(d->n_blocks + 1));
        if (gx) {
            ResrConvDesc cd = dgrad(h, w, 64, 32, 64, 0, p.ci_pad, p.ci_pad, p.ci_pad, 0, lo_t, 0, lo_xin);
            cd.in0_chunk_stride = plane;
            RUN(conv3x3_dispatch(&cd, b.gT[cur], nullptr, pk + p.pk_bwd_conv1 * wes, nullptr, nullptr, nullptr, nullptr, b.gxin, nullptr, st));
            RUN(nhwc_to_nchw_scaled_dispatch(b.gxin, gx, N, d->in_channels, d->h, d->w, p.r, p.ci_pad, dt, st, lo_xin, gsc));
        }
    }
    return RESR_OK;
}

}  // namespace resr
