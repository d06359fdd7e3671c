/*
 * resr.h -- C-ABI of libresr_hip.so: the MI355X (gfx950) Real-ESRGAN hot path.
 *
 * The reference (Lornatang/Real_ESRGAN-PyTorch) has no FFI: its device arithmetic is stock
 * PyTorch op call sites.  Each entry point below names the reference call site(s) it replaces
 * (file:line relative to the reference root).  All pointers are raw device pointers owned by the
 * caller (PyTorch's caching allocator); the library never allocates, frees or retains device
 * memory, never synchronises the stream or the device, and is graph-capture safe (chained dense-block launches inside a
 * captured graph: see resr_conv3x3_chain).  Test / measurement aids (resr_debug_*, resr_profile_*) are exported too but are NOT
 * part of this contract: they are declared in resr_debug.h.  What a launch needs beyond its operands -- scratch, slabs, the progress flags of the chained
 * dense-block launches -- is part of a workspace the caller passes in.  Every call enqueues on the
 * given hipStream_t (passed as void*) of the current device and returns 0 or a negative
 * resr_status; resr_last_error() returns a thread-local message.  No C++ exception crosses.
 *
 * Layouts.  Activations: NHWC ("pixel-major"), element type f16 (fast mode) or f32 (strict mode),
 * channels padded to a multiple of 32, addressed as base + pixel * pixel_stride + channel; a conv
 * reads a channel *prefix* of one or two such tensors, which is how the dense-block concat of
 * model.py:91-94 is served without materialising it.  Weights: packed MFMA A-fragments made by
 * resr_pack_weights from the module's OIHW fp32 parameters.
 */
#ifndef RESR_H_
#define RESR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: ResrWgradDesc gained x_chunk_stride / g_chunk_stride (round 4), ResrConvDesc.reserved_ became x2_pair_chunks and the struct gained mask_lo_offset,
 * ResrGeneratorDesc gained x2_plan.  A caller built against an older header passes shorter structs: compare resr_version() with the RESR_VERSION it was
 * compiled with and refuse on a mismatch.  New fields are appended (or replace reserved ones) and mean "as before" when zero, so
 * zero-initialise every descriptor (memset / = {0}) before filling it. */
/* 3 (round 6): ResrConvDesc gained in0_q_offset / in1_q_offset / out_q_offset / w_mx_offset (RESR_CONV_MX_PAIRS), ChainJob-level
 * counterparts follow from the per-job descriptors; resr_pack_weights_mx; RESR_X2_PLAN_MX_INFER. */
#define RESR_VERSION 3

typedef enum {
    RESR_OK = 0,
    RESR_ERR_ARG = -1,      /* bad descriptor / unsupported shape */
    RESR_ERR_LAUNCH = -2,   /* hipLaunchKernel failed */
    RESR_ERR_WORKSPACE = -3 /* workspace too small */
} resr_status;

/* RESR_F16X2 ("exact16" mode): every activation tensor is a PAIR of f16 tensors of identical shape and strides,
 *   value = hi + lo * 2^-12,   hi = f16(value),   lo = f16((value - hi) * 2^12)
 * (about 22 significand bits, the scaled lo keeps clear of the f16 subnormal range), addressed as `ptr` (hi) and
 * `ptr + *_lo_offset` elements (lo).  Packed weights hold three f16 blocks per 32-channel chunk -- f16(w*2^12),
 * its remainder, and f16(w) -- and a conv pass runs three v_mfma_f32_32x32x16_f16 per product into one fp32
 * accumulator (x_hi*W0 + x_hi*W1 + x_lo*W2, descaled by 2^-12 in the epilogue): fp32-class results on the f16
 * matrix pipe (the x_lo*w_lo term, 2^-22 relative, is dropped).  |w| must stay below 16. */
typedef enum { RESR_F16 = 0, RESR_F32 = 1, RESR_F16X2 = 2 } resr_dtype;

/* epilogue / gather flags of resr_conv3x3 */
enum {
    RESR_CONV_LRELU = 1 << 0,        /* v = v > 0 ? v : slope*v              (model.py:81,90-93,237,242,248) */
    RESR_CONV_UPSAMPLE_IN = 1 << 1,  /* input is nearest-upsampled x2 on load (model.py:264-265)            */
    RESR_CONV_CLAMP01 = 1 << 2,      /* v = min(max(v,0),1)                   (model.py:270)                */
    RESR_CONV_OUT_NCHW_F32 = 1 << 3, /* write planar fp32 [N,cout,H,W] (+ pass-mask bytes to aux)           */
    RESR_CONV_MASK = 1 << 4,         /* v *= (mask[p,c] > 0 ? 1 : slope): LeakyReLU backward (RESR_F16X2: mask_lo_offset) */
    RESR_CONV_NO_BIAS = 1 << 5,
    RESR_CONV_AUX_BEFORE_MASK = 1 << 6, /* aux_out (NHWC, out_stride) also receives v after bias, before the mask     */
    RESR_CONV_AUX_BEFORE_RES = 1 << 7,  /* aux_out (NHWC, out_stride) also receives v after LeakyReLU, before residuals */
    /* 1-bit LeakyReLU masks.  A sign tensor is uint32 [pixels][cout_pad/32]: bit c of word (p, m) = (out[p, 32m+c] > 0)
     * for the value as stored.  A forward pass can emit it (aux_out; no mask / residuals / other aux use in the same
     * pass); the matching backward-data pass reads it instead of re-reading the saved activation (64 B -> 4 B per
     * pixel and 32 channels in f16).  MASK_BITS goes together with RESR_CONV_MASK, without residuals. */
    RESR_CONV_WRITE_SIGNBITS = 1 << 8,
    RESR_CONV_MASK_BITS = 1 << 9,
    /* RESR_F16X2 only: the output is stored as ONE f16 tensor (the hi tensor; no lo tensor is written and out_lo_offset is
     * ignored).  A later pass reads such a tensor as a "single" chunk (x2_pair_chunks).  Ignored for other dtypes. */
    RESR_CONV_OUT_SINGLE = 1 << 10,
    /* RESR_F16X2 with x2_pair_chunks: the single chunks meet the f16 weights W0 alone -- ONE stage (x W0) instead of two (x W0 + x W1).
     * For operands that are the small ones of the sum (the growth planes of an inference forward next to the residual stream: forward
     * 1.4-2.1e-6 instead of 1.0-1.4e-6 at the reference's init, 2.4-2.9e-5 instead of 1.6-2.9e-5 with the dense weights x 4). */
    RESR_CONV_SINGLE_W16 = 1 << 11,
    /* RESR_F16X2 only: the PAIR chunks take TWO stages instead of three -- the main product x_hi W0 on the f16 matrix pipe, and BOTH
     * 2^-12-weighted corrections (x_hi W1 + x_lo W2) as ONE stage of nine v_mfma_scale_f32_32x32x64_f8f6f4 per output row on 8-bit
     * operands: B = the pixel's "q record" [bf8(x_hi) x 32 | bf8(x_lo) x 32] (64 B per pixel and chunk, K = 64 per tap), A = the
     * packed block [bf8(W1) | bf8(W2)] of resr_pack_weights_mx, unit scales (bf8 = e5m2 has f16's exponent range; the remainders
     * are stored times 2^12).  Every pair operand then needs its q tensor (in0_q_offset / in1_q_offset), the weights their MX blocks
     * (w_mx_offset); out_q_offset != 0 makes a pass EMIT the q tensor of its (pair) output for the passes behind it.  The
     * corrections need ~4 good bits, not 11: forward 0.8-1.2e-4 of the fp32 result over 23 blocks, 2/3 of the
     * matrix time of a pair chunk.  Inference epilogues only (bias / LeakyReLU / residuals). */
    RESR_CONV_MX_PAIRS = 1 << 12
};

/* One 3x3, stride 1, pad 1 convolution pass (forward conv or backward-data conv):
 *   v = sum_{tap,c} W[co][c][tap] * in[p + tap][c] + bias[co]
 *   v = mask / lrelu / (v*s0 + t0*res0[p,co]) / (v*s1 + t1*res1[p,co]) / clamp, in that order.
 * Replaces the F.conv2d + leaky_relu + torch.cat + mul/add call sites of model.py:87-98,
 * 123-132, 255-272 and their autograd backward-data counterparts. */
typedef struct {
    int32_t n, h, w;         /* output batch / height / width (input is h/2 x w/2 when UPSAMPLE_IN) */
    int32_t cin;             /* input channels, multiple of 32                                        */
    int32_t cin0;            /* channels [0,cin0) come from in0, [cin0,cin) from in1; multiple of 32  */
    int32_t in0_stride;      /* pixel strides in elements                                             */
    int32_t in1_stride;
    int32_t cout;            /* real output channels (<= cout_pad)                                    */
    int32_t cout_pad;        /* 32 or 64: rows of the packed weight tile                              */
    int32_t out_stride;      /* NHWC pixel stride of out (ignored for OUT_NCHW_F32)                   */
    int32_t res0_stride, res1_stride, mask_stride;
    int32_t dtype;           /* resr_dtype of activations and packed weights                          */
    int32_t flags;
    float s0, t0, s1, t1, slope;
    /* Chunk strides, in elements: distance between consecutive 32-channel chunks of an operand.  0 means 32, i.e. the
     * chunks of a pixel are interleaved (plain NHWC).  A chunk-planar tensor [C/32][N,H,W,32] has pixel stride 32 and
     * chunk stride N*H*W*32: every 64-byte (f16) piece a pass touches is then contiguous with its x-neighbours, so
     * HBM lines are consumed whole (the generator keeps its dense-block workspaces this way). */
    int32_t in0_chunk_stride, in1_chunk_stride, out_chunk_stride;
    int32_t res0_chunk_stride, res1_chunk_stride, mask_chunk_stride;
    /* RESR_F16X2 only: element offset from the hi tensor of an operand to its lo tensor (a mask given as an f16
     * activation: mask_lo_offset below; the aux tensor of AUX_BEFORE_* uses out's offset). */
    int64_t in0_lo_offset, in1_lo_offset, out_lo_offset, res0_lo_offset, res1_lo_offset;
    /* 4x4 / stride-2 convolutions (model.py:140-152) run as 3x3 convolutions over the 2x2 space-to-depth image with the
     * "virtual" kernel of ResrPackChunk.virtual4x4; 20 of its 36 (tap, sub-position) blocks are zero.  These two hints let
     * the kernel skip them (results are identical with or without; honoured for f16 / f16x2, 64-channel output groups and a
     * plain epilogue):  s2d_in_channels = C > 0: the input is the space-to-depth image, C channels per sub-position (cin = 4C,
     * forward pass);  s2d_out_channels = C > 0: the OUTPUT (all cout_groups * 64 = 4C channels of it) is the gradient of
     * such an image (backward-data pass).  0 = dense.
     * cout_groups = G > 1: ONE launch computes G consecutive 64-channel output groups of a convolution (f16 / f16x2, cout =
     * cout_pad = 64 per group, no bias, NHWC output): group g uses the packed weights at w_packed + g * (cin/32)*9*2*1024
     * elements (how resr_pack_weights lays out consecutive 64-row chunks tables) and writes / reads out, res0, res1, mask,
     * aux 64*g channels further inside the pixel.  The discriminator's 128..512-channel layers at 32^2..128^2 pixels
     * fill the 256 CUs only this way. */
    int32_t s2d_in_channels, s2d_out_channels, cout_groups;
    /* RESR_F16X2 only: P > 0 = only the FIRST P 32-channel input chunks are hi/lo pairs (three stages each: x_hi W0 + x_hi W1 +
     * x_lo W2); the chunks behind them are single f16 tensors -- no lo tensor is read, two stages each (x W0 + x W1: the weights
     * stay split, the activation carries 11 bits).  0 = every chunk is a pair.  The dense blocks use P = 2: the residual stream
     * (x0 x1 / g_y) as pairs, the growth planes (o1..o4 at inference, their gradients in backward) as single f16 -- the rungs of
     * DESIGN section 2 that keep the 1e-3 gate at 50 instead of 60 stages per block. */
    int32_t x2_pair_chunks;
    /* RESR_F16X2, RESR_CONV_MASK without _BITS: element offset from the mask's hi tensor to its lo tensor when the saved activation
     * is a pair of out's layout.  hi decides; the lo half is read only where hi rounded to zero (|value| < 2^-25: below f16's
     * subnormals), so that a LeakyReLU mask survives activations of any scale.  0 = the hi tensor alone (such values count as
     * "not positive"; the mask may then be any f16 tensor, e.g. a view of a larger batch). */
    int32_t reserved2_;
    int64_t mask_lo_offset;
    /* RESR_F16X2, RESR_CONV_MX_PAIRS (version 3): ELEMENT offsets (2 bytes each, like the lo offsets) from the hi tensor of a pair
     * operand to its q tensor -- same pixel / chunk strides as the hi tensor, 64 bytes per pixel and 32-channel chunk: byte c = bf8
     * (e5m2, round to nearest even) of hi[c], byte 32 + c = bf8 of lo[c].  out_q_offset != 0 (any pass with a pair output and a lean
     * epilogue): also write out's q tensor.  w_mx_offset: BYTE offset from w_packed to this convolution's MX blocks, one block of
     * 9 x cout_pad x 64 bytes per 32-channel input chunk in chunk order (resr_pack_weights_mx). */
    int64_t in0_q_offset, in1_q_offset, out_q_offset, w_mx_offset;
} ResrConvDesc;

int resr_conv3x3(const ResrConvDesc* d, const void* in0, const void* in1, const void* w_packed,
                 const float* bias, const void* res0, const void* res1, const void* mask,
                 void* out, void* aux_out, void* stream);

/* The growth convolutions of one dense block in ONE call (model.py:90-93: out_k = leaky_relu(conv_k(cat(x, out_1 ..
 * out_{k-1})))), or their mirrored backward-data passes: njobs (2..4) descriptors as for resr_conv3x3, all reading the
 * SAME in0 [/ in1] tensors -- job j the first descs[j].cin channels of them -- each with its own packed weights, bias (or
 * NULL), sign-word mask (or NULL), output and aux_out (or NULL); job j's output must be the tensor that job j+1 reads as
 * its last 32 channels.  Same results as njobs calls of resr_conv3x3 (to which it falls back); where the fast-mode kernel
 * supports it (f16 or f16x2, cout 32, chunk-planar operands, batch a multiple of 8, even width, LeakyReLU [+ sign words] or
 * the sign-word mask) the jobs run as one persistent launch that orders them through per-tile flags instead of kernel
 * boundaries.  bias / mask / aux_out may be NULL pointers to mean "no job has one".
 * chain_state: resr_conv3x3_chain_state_bytes(n, h, w) bytes of device memory OWNED BY THE CALLER (16-byte aligned), filled
 * with zeros once before its first use and then left alone: epoch, per-XCD tickets, per-tile flags and error words of the
 * chained launches live there (nothing is allocated, reset or synchronised by the library; consecutive launches on one
 * stream may share it, launches that can overlap in time must not).  NULL / too small: one launch per job.
 * A chained launch spins on its own workgroups, so only one runs at a time per device: a call from another stream first makes
 * that stream wait (hipStreamWaitEvent) for the previous chain.  A flag poll that gives up (~4 s: a lost workgroup) is never
 * silent -- the plane it waited for is read as NaNs, so the output (and a training loss) turns NaN, and resr_chain_errors()
 * counts it.
 * Under stream capture the call neither waits for, records, nor takes ownership of anything (none of it would mean anything at
 * replay time): inside one graph the chained launches are ordered by the capture itself, and the CALLER guarantees that a graph
 * holding chained launches is not replayed while another stream runs chained launches on the same device. */
size_t resr_conv3x3_chain_state_bytes(int32_t n, int32_t h, int32_t w);
int resr_conv3x3_chain(int32_t njobs, const ResrConvDesc* descs, const void* in0, const void* in1,
                       const void* const* packed_w, const float* const* bias, const void* const* mask,
                       void* const* out, void* const* aux_out, void* chain_state, size_t chain_state_bytes, void* stream);
/* Health of the chained launches on the current device: low 32 bits = flag polls that gave up, high 32 bits = workgroups an
 * XCD received beyond its share of the grid (tile ownership follows the XCD a workgroup really runs on and assumes the
 * dispatcher deals every XCD grid / 8 workgroups).  Both must be 0.  Reads two host-mapped words the kernels add to: no
 * synchronisation, cheap enough to call every few steps; it reflects the launches that have finished by then. */
int64_t resr_chain_errors(void);

/* Weight-gradient of the same convolution: dW[co][ci][tap] = sum_p G[p][co] * X[p + tap][ci]
 * (autograd backward of F.conv2d wrt weight, all conv call sites of model.py) and
 * db[co] = sum_p G[p][co].  Two launches: partial sums over pixel splits into `partial`
 * (fp32, resr_wgrad_partial_bytes), then a deterministic reduce that writes dW (OIHW fp32,
 * scaled by `scale`) and db.  cout_pad is 32 or 64 -- or, for RESR_F16 / RESR_F16X2, any multiple of 32: a convolution of
 * more than 80 (32-channel chunk x 32-channel tile) products (RESR_F16X2: more than 96 tap-products) runs as ONE "layer mode"
 * launch pair whose jobs follow from their grid position instead of a job table (the discriminator's 256..512-channel
 * layers, model.py:140-160). */
typedef struct {
    int32_t n, h, w;
    int32_t cin, cin0, in0_stride, in1_stride; /* X operand, same addressing as ResrConvDesc      */
    int32_t cin_real;                          /* rows of dW actually written (<= cin)            */
    int32_t cout, cout_pad, g_stride;          /* G operand: channels [0,cout_pad) of g           */
    int32_t dtype, flags;                      /* RESR_CONV_UPSAMPLE_IN honoured for X; RESR_CONV_OUT_SINGLE: G is a single f16 tensor */
    int32_t splits;                            /* pixel splits (partial slabs)                    */
    float scale;
    int64_t x_lo_offset, g_lo_offset;          /* RESR_F16X2: hi -> lo element offsets of X and G (both required).  A G that a
                                                * pass stored as ONE f16 tensor is selected explicitly by RESR_CONV_OUT_SINGLE in
                                                * flags (g_lo_offset is then ignored): dW = X_hi^T G + 2^-12 X_lo^T G; a zero
                                                * g_lo_offset without the flag is an argument error (version 3)                 */
    /* elements between consecutive 32-channel chunks of X / G (0 = 32: interleaved NHWC; a chunk-planar tensor
     * [C/32][N,H,W,32] has pixel stride 32 and chunk stride N*H*W*32 -- how the generator keeps its dense-block workspaces) */
    int64_t x_chunk_stride, g_chunk_stride;
} ResrWgradDesc;

size_t resr_wgrad_partial_bytes(const ResrWgradDesc* d);
int resr_conv3x3_wgrad(const ResrWgradDesc* d, const void* x0, const void* x1, const void* g,
                       float* partial, float* dw, float* db, void* stream);

/* Weight packing: a table of chunk descriptors turns the flat fp32 OIHW parameter arena into
 * MFMA A-fragment order (forward) or its transposed+flipped form (backward-data). */
typedef struct {
    int64_t src_off;     /* element offset of the source conv weight in the fp32 arena            */
    int64_t dst_off;     /* element offset of this chunk in the packed buffer                     */
    int32_t src_cout, src_cin;
    int32_t m_off, m_count; /* first row / number of real rows (rest zero) of this chunk's M      */
    int32_t k_off, k_count; /* first / number of real K channels of this chunk (<= 32)            */
    int32_t mt;          /* M tiles (1 or 2)                                                      */
    int32_t transposed;  /* 0: M = cout, K = cin;  1: M = cin, K = cout, taps flipped             */
    float scale;
    int32_t virtual4x4;  /* 1: source is a [cout][C=src_cin][4][4] stride-2 kernel seen as a 3x3 kernel
                            over the 2x2 space-to-depth image (4C virtual channels, (i*2+j)*C + c)       */
    const float* scale_ptr; /* optional device scalar multiplied in (1/sigma of spectral norm)           */
} ResrPackChunk;

int resr_pack_weights(const ResrPackChunk* chunks_dev, int32_t n_chunks, const float* arena,
                      void* packed, int32_t dtype, void* stream);
/* The MX blocks of RESR_CONV_MX_PAIRS from the SAME chunk table: chunk i becomes one block of 9 x (32 mt) x 64 bytes at
 * packed_mx + 2 * dst_off bytes (dst_off counts elements of the plain f16 layout, so the MX region mirrors a plain f16 packing byte for
 * byte): per tap and output row the bytes [bf8(W1[k]), k = 0..31 | bf8(W2[k])] of the exact16 split of resr_pack_weights (W0 =
 * f16(w 2^12), W1 = f16(w 2^12 - W0), W2 = f16(W0 2^-12)), in the A-fragment order of v_mfma_scale_f32_32x32x64_f8f6f4 (lane (row,
 * half h): K = 16 h .. 16 h + 15 of either block). */
int resr_pack_weights_mx(const ResrPackChunk* chunks_dev, int32_t n_chunks, const float* arena, void* packed_mx, void* stream);

/* Layout helpers around the generator (model.py:257 PixelUnshuffle, NCHW fp32 module surface). */
int resr_nchw_to_nhwc(const float* src, void* dst, int32_t n, int32_t c, int32_t h, int32_t w,
                      int32_t unshuffle, int32_t c_pad, int32_t dtype, const uint8_t* mask,
                      void* stream);
int resr_nhwc_to_nchw(const void* src, float* dst, int32_t n, int32_t c, int32_t h, int32_t w,
                      int32_t shuffle, int32_t src_stride, int32_t dtype, void* stream);
/* backward of nearest x2 upsample (+ optional LeakyReLU mask of the producer's output) */
int resr_sumpool2x2(const void* src, void* dst, const void* mask, int32_t n, int32_t h_out,
                    int32_t w_out, int32_t c, int32_t dtype, float slope, void* stream);

/* Whole-generator passes (model.py:255-272 and its autograd backward), enqueued natively so the
 * ~350 / ~1100 launches cost no Python time.  See real_esrgan-pytorch_amd/csrc/generator.hip. */
typedef struct {
    int32_t n, h, w;          /* input batch / height / width (before pixel-unshuffle)           */
    int32_t in_channels, out_channels, upscale; /* Generator(in, out, upscale) ctor args          */
    int32_t n_blocks;         /* RRDB count (23)                                                  */
    int32_t dtype;            /* RESR_F16 fast / RESR_F32 strict                                  */
    int32_t training;         /* keep activations for backward                                    */
    int32_t wgrad_splits;     /* 0 = auto                                                         */
    /* RESR_F16X2 only, bit set of RESR_X2_PLAN_*: which tensors of the dense blocks are single f16 instead of hi/lo pairs.  0 =
     * pairs everywhere (three stages per chunk in every pass: forward 1.8e-6, gradients 5.8e-6 vs float64). */
    int32_t x2_plan;
    int32_t reserved_;
} ResrGeneratorDesc;
enum {
    /* inference forward (training = 0) only: the growth planes o1..o4 of every dense block are single f16 tensors, the residual
     * stream and the HR tail stay pairs, weights stay split: 50 instead of 60 stages per block (forward 1e-6 at the reference's
     * init scale; a TRAINING forward ignores the bit -- a 2^-12 perturbation of a pre-activation flips LeakyReLU mask elements) */
    RESR_X2_PLAN_GROWTH_F16_INFER = 1,
    /* backward: the gradients of the growth planes (g_o1..g_o4) are READ as single f16 tensors (their hi tensor): their chunks
     * take two stages in the mirrored backward-data passes and conv1..conv4's weight gradients two tap-products instead of
     * three; they are still stored as pairs, and the BIAS gradients sum hi + lo (one extra tap-product per convolution).  Worst
     * gradient tensor 3-5e-4 vs float64 */
    RESR_X2_PLAN_GROWTH_GRAD_F16 = 2,
    /* with GROWTH_GRAD_F16: store the growth-plane gradients as single f16 tensors too (no lo store in the mirrored passes, no
     * bias job: 64 instead of 68 tap-products per block, ~4 % of an exact16 step).  The bias gradients of conv1..conv4 are then
     * sums of ROUNDED values: the emulation's worst bias tensor reaches 6.7e-4 at 1 x 128^2 (inside the 1e-3 gate, outside the
     * 5e-4 rule the default plan keeps): opt-in. */
    RESR_X2_PLAN_GROWTH_GRAD_STORE_F16 = 4,
    /* backward: the weight products read the growth planes o1..o4 (the X chunks behind the residual stream of conv2..conv5) as their
     * hi tensor: no (x_lo, g_hi) tap-product for them -- 14 fewer tap-products per dense block (54 instead of 68 next to
     * GROWTH_GRAD_F16).  The forward pass and backward-data still see the pairs.  What is dropped is a zero-mean residue of the
     * SMALL operand of conv5's products (the growth planes next to the stream): the worst gradient tensor does not move (emulation:
     * 3.0e-4 -> 3.0e-4 at the reference's init, 3.9e-4 -> 4.0e-4 with the dense weights x 4; conv5's own tensors 7e-7 -> 2.6e-5 / 1e-4) */
    RESR_X2_PLAN_GROWTH_ACT_F16_WGRAD = 8,
    /* with GROWTH_ACT_F16_WGRAD: conv5's products of the growth planes also take g_y's hi tensor alone -- one tap-product
     * (x_hi, g_hi) per growth chunk, 46 per dense block.  Both dropped residues belong to the small operand block of conv5's weight
     * tensor (emulation: conv5's tensors 2.6e-5 -> 3.8e-5, 1.0e-4 -> 1.5e-4 with the dense weights x 4; the worst tensor does not move) */
    RESR_X2_PLAN_GROWTH_ACT_G_HI_WGRAD = 16,
    /* with GROWTH_F16_INFER: the growth chunks of an inference forward take ONE stage (RESR_CONV_SINGLE_W16): 40 instead of 50 stages per
     * dense block */
    RESR_X2_PLAN_GROWTH_W16_INFER = 32,
    /* inference forward (training = 0) only, with GROWTH_F16_INFER + GROWTH_W16_INFER: the pair chunks (residual stream, HR tail) take
     * one f16 stage + one MX-fp8 stage (RESR_CONV_MX_PAIRS) instead of three f16 stages: 30 stage-equivalents per dense block
     * instead of 40.  The workspace grows by the q tensors, the packed weights by their MX region (resr_generator_packed_bytes,
     * resr_generator_mx_offset).  Forward 0.8-1.2e-4 of the fp32 oracle instead of 2e-6 (gate 2e-4). */
    RESR_X2_PLAN_MX_INFER = 64,
    /* backward (training = 1): the backward-data passes of the dense blocks (the four mirrored cout-32 passes and the g_x pass of every
     * block) read EVERY gradient chunk as a pair on one f16 stage + one MX stage (RESR_CONV_MX_PAIRS): 40 stage-equivalents per block
     * instead of 50 (GROWTH_GRAD_F16) or 60, and nothing is dropped -- the growth-plane gradients enter with both halves again: worst
     * gradient tensor 0.8-1.2e-4 vs float64 in the emulation (GROWTH_GRAD_F16: 3-5e-4).  The gradient planes gT / gS carry q tensors
     * (written by the passes that produce them), the packed buffer its MX region.  The forward pass and the weight gradients are
     * untouched (GROWTH_GRAD_F16 then only shapes the weight products); not together with GROWTH_GRAD_STORE_F16. */
    RESR_X2_PLAN_MX_BWD = 128,
    /* opt-in: the exact16 forward (all pairs: reference-exact masks, fp32-class losses) followed by FAST mode's backward pass -- plain f16
     * on the hi tensors, f16 weights: resr_generator_backward then takes a RESR_F16 packing of the same table as `packed`.  Most of fast
     * mode's gradient error is its own forward's mask flips, not its backward pass's roundings (emulation: median 4-6e-3 -> 5-7e-4 under the
     * L1 loss): a third operating point between fast and exact16.  Overrides the other backward bits. */
    RESR_X2_PLAN_F16_BACKWARD = 256,
    /* with MX_BWD: the weight gradients of the dense blocks take the two 2^-12-weighted tap-products (x_hi, g_lo) + (x_lo, g_hi) of every
     * STREAM chunk as ONE MX job -- K is pixels there: 8-bit transpose reads (ds_read_b64_tr_b8) of the staged q records feed one
     * v_mfma_scale_f32_32x32x64_f8f6f4 per output row and tap, block 0 = (g_lo, x_hi), block 1 = (g_hi, x_lo) -- half the staged bytes
     * and half the matrix time of the two f16 tap-products, and conv1..conv4 get their (x_hi, g_lo) term back (GROWTH_GRAD_F16 drops it).
     * The training forward then emits the q tensor of the residual stream (conv1 and every closing convolution: + 2 of 12 plane stores
     * per dense block), the workspace holds it, and g_lo's share of a bias gradient is summed by the MX job itself from the bf8 bytes of
     * its G fragments (a bias moves by 1-2e-5 against the f16 lo tensor's sum; gate 5e-5 in tests/test_gpu_mx.py). */
    RESR_X2_PLAN_MX_WGRAD = 512,
    /* with MX_WGRAD: the same treatment for the 4x-resolution tail (conv3, conv4, upsampling2 -- a third of an exact16 step's weight-gradient
     * work and a tenth of its backward-data work): the training forward emits the q tensors of u1, u2 and c3, the layout pass that brings the
     * incoming gradient in and the two masked tail passes emit those of the gradient tensors, the three tail passes that read them run one
     * f16 + one MX stage per chunk, and the three weight gradients take their correction tap-products as MX jobs.  upsampling1 keeps its f16
     * form (its gradient comes out of a 2 x 2 sum-pool). */
    RESR_X2_PLAN_MX_TAIL = 1024
};

size_t resr_generator_param_count(const ResrGeneratorDesc* d);
size_t resr_generator_packed_bytes(const ResrGeneratorDesc* d, int32_t backward);
/* RESR_F16X2: byte offset of the MX region (resr_pack_weights_mx; RESR_X2_PLAN_MX_INFER) inside a packed buffer of
 * resr_generator_packed_bytes(d, *) bytes: pack the same chunk table there after resr_pack_weights.  0 for other dtypes. */
size_t resr_generator_mx_offset(const ResrGeneratorDesc* d);
size_t resr_generator_workspace_bytes(const ResrGeneratorDesc* d);
/* The first resr_generator_chain_state_bytes(d) bytes of a generator workspace are the chain state of its dense-block
 * launches (resr_conv3x3_chain) and, behind it, the 256-byte slot of the backward pass's gradient pre-scale (below): the caller
 * fills them with zeros once, when the workspace is allocated. */
size_t resr_generator_chain_state_bytes(const ResrGeneratorDesc* d);
/* fills `chunks` (host memory, capacity in elements) and returns the count; forward table first,
 * then (if backward) the backward-data table; dst offsets are relative to one packed buffer */
int64_t resr_generator_pack_table(const ResrGeneratorDesc* d, int32_t backward,
                                  ResrPackChunk* chunks, int64_t capacity);
/* test aid: byte offsets inside the workspace of {x_in, ws[0], out1, trunk_out, feat, u1, u2, c3,
 * ymask, g4, gA, gB, gM1, gF, gT0, gT1, gT2, gT3, gS, gxin, partial}; -1 = not allocated */
int64_t resr_generator_buffer_offsets(const ResrGeneratorDesc* d, int64_t* out, int64_t capacity);
int resr_generator_forward(const ResrGeneratorDesc* d, const float* x_nchw, const float* params,
                           const void* packed, void* workspace, size_t workspace_bytes,
                           float* y_nchw, void* stream);
/* grad_ready_events (optional, n_events = n_blocks + 2 hipEvent_t handles, else NULL / 0): recorded on `stream` as soon as a
 * range of the gradient arena is final, in backward order -- [0] the tail (conv2, upsampling1/2, conv3, conv4: the END of the
 * arena), [1 + j] RRDB n_blocks-1-j, [n_blocks + 1] conv1 (everything).  A data-parallel caller makes its communication
 * stream wait on them and all-reduces each range while the rest of the backward pass still runs.
 * RESR_F16 / RESR_F16X2: the pass does not depend on the scale of gy -- when max |gy| < 2^6 it runs on gy * 2^k (max in [2^6, 2^7)) and hands
 * grad / gx out times 2^-k, both exact, so its f16 tensors never see subnormals at small loss scales; an inf / NaN in gy turns the
 * lift off and reaches grad (resr_discriminator_backward does the same).  $RESR_X2_GRAD_PRESCALE_LOG2, RESR_X2_NO_GRAD_PRESCALE=1.
 * The lift leaves 2^9 of headroom to f16's maximum.  A gradient that grows by more than that on its way back overflows whatever the
 * caller's loss scale is -- a GradScaler halving its scale cannot cure an overflow the lift re-creates every step (the symptom would
 * be found_inf on every step and a scale decaying to nothing) -- so the pass backs off by itself: a lifted pass that writes a non-finite
 * weight gradient sets a flag in the workspace's pre-scale slot, and every later pass on that workspace aims 2^4 lower per such event
 * (sticky, up to 2^40 = lift off: the caller's scale rules again).  The overflowed step is the GradScaler's to skip, as ever. */
int resr_generator_backward(const ResrGeneratorDesc* d, const float* gy_nchw, const float* params,
                            const void* packed, void* workspace, size_t workspace_bytes,
                            float* grad_params, float* gx_nchw, void* stream,
                            void* const* grad_ready_events, int32_t n_events);

/* ---- second-order degradation (imgproc.py device ops; call sites train_realesrnet.py:268-377) ----------
 * Images are planar fp32 [n,c,h,w] in [0,1].  No entry point synchronises or reads back. */

/* filter2d_torch (imgproc.py:1089-1121): reflect pad, correlation with a kh x kw kernel (odd sizes);
 * per_sample = 1: kernel is [n,kh,kw], else [kh,kw] shared. */
int resr_filter2d(const float* src, float* dst, const float* kernel, int32_t n, int32_t c, int32_t h, int32_t w,
                  int32_t kh, int32_t kw, int32_t per_sample, void* stream);
/* USMSharp.forward (imgproc.py:1526-1537) with the Gaussian given as its 1-D factor k1d[ksize];
 * tmp3 = 3*n*c*h*w floats of scratch, kept by a caller that will run resr_usm_sharp_bwd: [mask bytes / row pass | blur | soft].
 * ksize = 51 (USMSharp(50, 0), the only configuration the reference builds) runs as two fused launches -- blur + byte mask, soft
 * mask + combine: 26 B per value instead of the 60 B of six separate passes. */
int resr_usm_sharp(const float* src, float* dst, float* tmp3, const float* k1d, int32_t ksize, float weight,
                   float threshold, int32_t n, int32_t c, int32_t h, int32_t w, void* stream);
/* the same without the soft-mask store the backward pass needs (the degradation path, train_realesrnet.py:268: no graph) */
int resr_usm_sharp_forward_only(const float* src, float* dst, float* tmp3, const float* k1d, int32_t ksize, float weight,
                                float threshold, int32_t n, int32_t c, int32_t h, int32_t w, void* stream);
/* backward of resr_usm_sharp wrt its input (the GAN step differentiates through usm_sharpener(sr),
 * train_realesrgan.py:476): saved_tmp3 = the forward's tmp3 (kept), tmp2 = 2*n*c*h*w floats of scratch. */
int resr_usm_sharp_bwd(const float* x, const float* saved_tmp3, const float* g, float* gx, float* tmp2, const float* k1d,
                       int32_t ksize, float weight, int32_t n, int32_t c, int32_t h, int32_t w, void* stream);
/* F.interpolate (train_realesrnet.py:288,326-329,349-351,366-368): mode 0 area, 1 bilinear, 2 bicubic;
 * scale_h/scale_w > 0: the caller used scale_factor= (coordinates use 1/scale), else size= semantics. */
int resr_resize(const float* src, float* dst, int32_t n, int32_t c, int32_t h, int32_t w, int32_t oh, int32_t ow,
                int32_t mode, double scale_h, double scale_w, void* stream);
/* standard-normal field, Philox4x32-10 (the device RNG draws of imgproc.py:854,858) */
int resr_randn_fill(float* dst, int64_t count, uint64_t seed, uint64_t stream_id, void* stream);
/* random_add_gaussian_noise_torch (imgproc.py:1029-1057) with the draws given: sigma[n], gray[n] (0/1),
 * field_gray[h*w] (ONE field for the batch, may be NULL when no sample is gray), field_color[n*c*h*w]. */
int resr_noise_gaussian(const float* src, float* dst, const float* sigma, const float* gray, const float* field_gray,
                        const float* field_color, int32_t n, int32_t c, int32_t h, int32_t w, int32_t clip, void* stream);
/* random_add_poisson_noise_torch (imgproc.py:1060-1086): per-sample unique-value counts on the device,
 * Poisson draws from Philox; scale[n], gray[n]. */
size_t resr_noise_poisson_workspace_bytes(int32_t n);
int resr_noise_poisson(const float* src, float* dst, const float* scale, const float* gray, uint64_t seed, void* workspace,
                       int32_t n, int32_t c, int32_t h, int32_t w, int32_t clip, void* stream);
/* DiffJPEG(differentiable=False).forward (imgproc.py:1462-1494); quality[n] (not mutated);
 * coeffs (optional) receives the rounded coefficients [n][Y blocks | Cb blocks | Cr blocks][64]. */
int resr_jpeg(const float* src, float* dst, const float* quality, float* coeffs, int32_t n, int32_t h, int32_t w,
              int32_t flags /* bit0: clamp the input to [0,1] first (train_realesrnet.py:308) */, void* stream);
/* clamp(round(x*255))/255 on lr + random_crop of both (train_realesrnet.py:374-377, imgproc.py:1894-1934).  hr_out = NULL is
 * allowed when the HR window is the whole image (hr_size == hr_h == hr_w): the caller keeps using `hr` instead of a copy of it. */
int resr_quantize_crop(const float* lr, const float* hr, float* lr_out, float* hr_out, int32_t n, int32_t c, int32_t lr_h,
                       int32_t lr_w, int32_t hr_h, int32_t hr_w, int32_t hr_size, int32_t upscale, int32_t hr_top,
                       int32_t hr_left, void* stream);

/* ---- integer mode of blur / resize / JPEG (north_star: "blur/resize/JPEG bit-exact in integer mode") --------------
 * uint8 planar images [n,c,h,w], fixed-point taps, integer accumulation: the CPU restatement (oracle/imgproc_int_ref.py)
 * and these kernels agree bit for bit.  They shadow the reference's float ops (imgproc.py:1089-1121 filter2d_torch; the
 * F.interpolate call sites train_realesrnet.py:288,326-329,349-351,366-368), to which the distance is <= 1 LSB.
 * resr_filter2d_u8: taps_q14 = the kernel in Q14 (int32, [kh*kw] or [n][kh*kw] with per_sample), summing to 2^14:
 *   dst = clamp((sum q*src[reflect] + 2^13) >> 14, 0, 255).
 * resr_resize_u8: mode 0 area (integer window means, round half up; tables unused), 1 bilinear (2 taps), 2 bicubic
 *   (4 taps): per-axis tables idx_* [out][taps] (clamped source indices) and w_* [out][taps] (Q11, summing to 2^11),
 *   made by the caller from ATen's coordinate map; dst = clamp((sum_y wy * (sum_x wx*src) + 2^21) >> 22, 0, 255). */
int resr_filter2d_u8(const uint8_t* src, uint8_t* dst, const int32_t* taps_q14, int32_t n, int32_t c, int32_t h, int32_t w,
                     int32_t kh, int32_t kw, int32_t per_sample, void* stream);
int resr_resize_u8(const uint8_t* src, uint8_t* dst, int32_t n, int32_t c, int32_t h, int32_t w, int32_t oh, int32_t ow,
                   int32_t mode, const int32_t* idx_y, const int32_t* w_y, const int32_t* idx_x, const int32_t* w_x, void* stream);
/* resr_jpeg_u8: DiffJPEG(differentiable=False).forward (imgproc.py:1462-1494) as an integer round trip on uint8 RGB
 *   [n,3,h,w]: zero padding to multiples of 16, RGB -> YCbCr and the 8x8 DCT with Q20 constants and int64 accumulation,
 *   quantiser steps rint(table * factor * 2^20) (the reference's transposed tables, imgproc.py:40-49; factor from the
 *   float32 quality[n] as imgproc.py:1134-1139, evaluated in float64 on the device), round-half-even division (torch.round),
 *   inverse DCT, nearest chroma upsampling, YCbCr -> RGB, clamp.  `coeffs` (nullable): the quantised coefficients, int32,
 *   per image [Y blocks row-major | Cb blocks | Cr blocks] x 64 over the padded size -- the same layout as resr_jpeg's.
 *   `quality` is read, never written (the reference mutates its argument in place; the Python mirror does that). */
int resr_jpeg_u8(const uint8_t* src, uint8_t* dst, const float* quality, int32_t* coeffs, int32_t n, int32_t h, int32_t w,
                 void* stream);

/* ---- discriminator helpers (model.py:135-203) -------------------------------------------------------- */
/* 2x2 space-to-depth of an NHWC tensor [n,h,w,c] -> [n,h/2,w/2,4c] (inverse != 0: depth-to-space) */
int resr_space_to_depth(const void* src, void* dst, int32_t n, int32_t h, int32_t w, int32_t c, int32_t dtype,
                        int32_t inverse, void* stream);
/* F.interpolate(scale_factor=2, mode="bilinear", align_corners=False) on NHWC [n,h,w,c] (model.py:186,190,194);
 * backward != 0: src is the gradient [n,2h,2w,c], dst the input gradient [n,h,w,c] */
int resr_bilinear_up2x(const void* src, void* dst, int32_t n, int32_t h, int32_t w, int32_t c, int32_t dtype,
                       int32_t backward, void* stream);
/* out = (a + b) * (mask > 0 ? 1 : slope); b and mask optional (RESR_F16X2: a, b, out are pairs with the lo tensor `count`
 * elements behind the hi tensor; the mask is an activation, read from its hi tensor) */
int resr_add_mask(const void* a, const void* b, const void* mask, void* out, int64_t count, int32_t dtype, float slope,
                  void* stream);
/* partial[k] = workgroup k's share of sum |a - b| over `count` elements (nblocks workgroups, fixed order: deterministic); the
 * caller adds the partials and divides: F.l1_loss of two feature tensors (model.py:320-327) without fp32 copies of them.
 * RESR_F16X2: a and b are pairs with the lo tensor lo_offset elements behind the hi tensor */
int resr_l1_partial(const void* a, const void* b, int64_t count, int32_t dtype, int64_t lo_offset, float* partial, int32_t nblocks,
                    void* stream);
/* ---- scalar losses of the train steps, forward value and unit gradient in ONE launch each --------------------------------
 * scratch: resr_loss_scratch_bytes() bytes of device memory owned by the caller, zero-filled once (an arrival counter + one
 * partial sum per workgroup; the last workgroup adds them in a fixed order and re-arms the counter: deterministic values).
 * Launches that may overlap in time need their own scratch.
 * resr_bce_logits_const: loss[0] = weight * mean_i BCEWithLogits(logits_i, label) against a CONSTANT label -- nn.BCEWithLogitsLoss
 *   on torch.full(..., 1.0 / 0.0) (train_realesrgan.py:460-461,478,500,509) without the label tensor; grad (nullable, [count]) =
 *   d loss / d logits_i = weight / count * (sigmoid(logits_i) - label).
 * resr_l1_mean: loss[0] = weight * mean_i |a_i - b_i| -- nn.L1Loss (train_realesrnet.py:385, train_realesrgan.py:475); grad_a
 *   (nullable) = weight / count * sign(a_i - b_i).  All tensors fp32, 16-byte aligned. */
size_t resr_loss_scratch_bytes(void);
int resr_bce_logits_const(const float* logits, int64_t count, float label, float weight, float* loss, float* grad, float* scratch,
                          void* stream);
int resr_l1_mean(const float* a, const float* b, int64_t count, float weight, float* loss, float* grad_a, float* scratch, void* stream);
/* out[r] = coef_host[r] * sum_k partial[r * cols + k] for r < rows (<= 8), out[rows] = their total: the weighted feature
 * distances of the perceptual term (model.py:320-335) from resr_l1_partial's partial sums, one launch; coef_host is HOST memory. */
int resr_weighted_row_sums(const float* partial, int32_t rows, int32_t cols, const float* coef_host, float* out, void* stream);

/* torch.nn.utils.spectral_norm forward (model.py:140-168): W [rows][cols] fp32; training: one power iteration
 * updating u[rows], v[cols] in place; sigma2[0] = sigma, sigma2[1] = 1/sigma; tmp = rows + ceil(rows/32) * cols floats
 * (W^T u is summed in 32-row groups, in a fixed order: bit-identical on every data-parallel rank) */
int resr_spectral_norm(const float* w, float* u, float* v, int32_t rows, int32_t cols, int32_t training, float eps,
                       float* sigma2, float* tmp, void* stream);
/* gradient wrt W_orig from the gradient wrt W = W_orig/sigma; tmp1 = 512 floats (block partials of <G, W>, added in a fixed
 * order: no atomics, the result is bit-reproducible) */
int resr_spectral_norm_bwd(const float* g, const float* w, const float* u, const float* v, const float* sigma2, float* dst,
                           int32_t rows, int32_t cols, int32_t accumulate, float* tmp1, void* stream);
/* 2x2 stride-2 max pooling on NHWC [n,2*h_out,2*w_out,c] (VGG19 of ContentLoss, model.py:296-298).  RESR_F16X2: src and dst
 * are (hi, lo) pairs, each lo tensor directly behind its hi tensor (this holds for every helper of this group). */
int resr_maxpool2x2(const void* src, void* dst, int32_t n, int32_t h_out, int32_t w_out, int32_t c, int32_t dtype,
                    void* stream);
/* ... also recording which window position won (arg: uint8 [n,h_out,w_out,c], value dy * 2 + dx, the first maximum in that order
 * like ATen's max_pool2d; may be NULL), and the backward pass through it: gin [n,2*h_out,2*w_out,c] receives g at the recorded
 * position of every window and zeros elsewhere (the differentiable perceptual term, model.py:311-335). */
int resr_maxpool2x2_arg(const void* src, void* dst, uint8_t* arg, int32_t n, int32_t h_out, int32_t w_out, int32_t c, int32_t dtype,
                        void* stream);
int resr_maxpool2x2_bwd(const void* g, const uint8_t* arg, void* gin, int32_t n, int32_t h_out, int32_t w_out, int32_t c, int32_t dtype,
                        void* stream);
/* virtual [cout][4C][3][3] weight gradient of a space-to-depth conv -> real [cout][C][4][4] */
int resr_fold4x4(const float* dw3, float* dw4, int32_t cout, int32_t c, void* stream);

/* Whole-discriminator passes (model.py:135-203, torch.nn.utils.spectral_norm included, and the autograd backward), enqueued
 * natively like the generator's: see real_esrgan-pytorch_amd/csrc/disc_native.hip for the parameter / spectral-norm arena
 * layouts (the reference's named_parameters() / named_buffers() orders) and the workspace plan.  The workspace holds the
 * activations kept for backward, THIS call's spectral-norm vectors, sigmas and packed weights (the module is called three
 * times per GAN step, train_realesrgan.py:479,500,508) and all scratch: nothing is allocated per tensor. */
typedef struct {
    int32_t n, h, w;       /* input [n,3,h,w], h and w multiples of 8                                  */
    int32_t dtype;         /* RESR_F16, RESR_F16X2 (hi/lo pairs: fp32-class results) or RESR_F32       */
    int32_t training;      /* keep activations for resr_discriminator_backward                         */
    int32_t sn_training;   /* module in training mode: one power iteration, u / v updated in place     */
} ResrDiscriminatorDesc;
size_t resr_discriminator_param_count(void);
size_t resr_discriminator_uv_count(void);
/* The first 256 bytes of a discriminator workspace (1 / sigma of the normalised layers + the gradient pre-scale slot of
 * resr_discriminator_backward, see resr_generator_backward) are filled with zeros by the caller once, when the workspace is allocated. */
size_t resr_discriminator_workspace_bytes(const ResrDiscriminatorDesc* d);
/* chunk table for the pack launch inside resr_discriminator_forward (host memory; returns the count, `chunks` may be NULL to
 * query it).  1/sigma of the normalised layers is read on the device from the head of `workspace`: one table per workspace. */
int64_t resr_discriminator_pack_table(const ResrDiscriminatorDesc* d, const void* workspace, ResrPackChunk* chunks, int64_t capacity);
int resr_discriminator_forward(const ResrDiscriminatorDesc* d, const float* x_nchw, const float* params, float* uv,
                               const ResrPackChunk* table_dev, int32_t n_chunks, void* workspace, size_t workspace_bytes,
                               float* y_nchw, void* stream);
/* grad_params = NULL: gradient wrt the input only (the generator's adversarial term, discriminator frozen); gx_nchw optional */
int resr_discriminator_backward(const ResrDiscriminatorDesc* d, const float* gy_nchw, const float* params, void* workspace,
                                size_t workspace_bytes, float* grad_params, float* gx_nchw, void* stream);

/* EMA.update (model.py:43-48) over the flat parameter arena, one launch. */
int resr_ema_update(float* shadow, const float* params, int64_t count, double decay, void* stream);

const char* resr_last_error(void);
int resr_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RESR_H_ */
