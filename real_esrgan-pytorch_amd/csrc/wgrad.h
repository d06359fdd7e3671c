// wgrad.h -- host-side description of one convolution of a batched weight-gradient launch (wgrad.hip); shared with
// the whole-network planners (generator.hip, disc_native.hip).
#pragma once

namespace resr {

struct WgradConv {
    const void* x0; int cin, in0_stride, cin_real;     // X: channel prefix [0,cin) of x0
    const void* g; int cout, cout_pad, g_stride;       // G: channels [0,cout_pad) of g
    long x_chunk_stride, g_chunk_stride;               // elements between 32-channel chunks of X / G (0 = 32: interleaved)
    long x_lo_off, g_lo_off;                           // RESR_F16X2: element offsets hi -> lo tensor of X / G (else 0; g_lo_off = 0: G is single f16)
    int g_lo_bias_only;                                // RESR_F16X2: the (x_hi, g_lo) tap-product only for X chunk 0 -- the job that also sums
                                                       // the bias: dW takes G's hi tensor, db takes hi + lo (generator.hip, x2_plan bit 1)
    int x_pair_chunks = 0;                             // RESR_F16X2: > 0 = only the first P X chunks are read as pairs; the chunks behind them (a dense
                                                       // block's growth planes) enter the weight products as their hi tensor: no (x_lo, g_hi) tap-product
                                                       // for them (generator.hip, x2_plan bit 3).  0 = every chunk a pair
    int x_single_g_hi = 0;                             // with x_pair_chunks: the products of those single X chunks also take G's hi tensor alone -- no (x_hi, g_lo)
                                                       // tap-product for them either (conv5 of a dense block, x2_plan bit 4)
    long x_q_off = 0, g_q_off = 0;                     // RESR_F16X2, both != 0 (x2_plan bit 9): element offsets hi -> q tensor of X / G (64 B per pixel and
                                                       // chunk: bf8 of hi | bf8 of lo).  The two 2^-12-weighted tap-products (x_hi, g_lo) and (x_lo, g_hi) of every
                                                       // PAIR X chunk then run as ONE "MX" job -- 8-bit transpose reads of the staged q records feeding
                                                       // v_mfma_scale_f32_32x32x64_f8f6f4, K = 32 pixels twice -- and sums g_lo's share of the bias from the same fragments
    int x_s2d_c;                                       // > 0: X is a space-to-depth image with this many channels per sub-position
                                                       // (virtual kernel of a 4x4 / stride-2 conv): the zero taps are skipped
    float* dw; float* db; float scale;
    const unsigned* unscale = nullptr;                 // device pointer to the bits of max |g_y| of a pre-scaled backward pass (common.h:
                                                       // grad_prescale): dw / db leave times its inverse; one per launch (convs[0]'s)
};

constexpr int kWgradMaxJobs = 96;   // (X chunk, G tile) tap-products per weight-gradient launch (kernel arguments: 96 x 40 B + header < 4 KB)

}  // namespace resr
