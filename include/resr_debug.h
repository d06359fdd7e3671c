/*
 * resr_debug.h -- test and measurement aids of libresr_hip.so.  Exported, but NOT part of the drop-in contract of resr.h:
 * nothing in the product path (model.py, train.py, inference.py ...) needs them; tests/, tools/ and bench.py do.
 * Unlike the entry points of resr.h, resr_debug_chain_errors and resr_profile_end synchronise.
 */
#ifndef RESR_DEBUG_H_
#define RESR_DEBUG_H_

#include "resr.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Debug: per-workgroup timeline of the fast-mode conv kernel (32 workgroups x 2 roles x 64 uint64 stamps, 100 MHz).  The stamps
 * are compiled in by a trace build only (-DRESR_TRACE=1, tools/build_variant.py); in the product build the buffer stays untouched. */
int resr_debug_conv_trace(void* dev_buf);
/* Debug / test: resr_chain_errors() after a hipDeviceSynchronize() (every launch enqueued so far has reported).
 * RESR_CONV_NO_CHAIN=1 in the environment disables chaining. */
int64_t resr_debug_chain_errors(void);
/* Test aid: `workgroups` single-wave workgroups that each hold `lds_bytes` of LDS and spin for `micros` microseconds -- what a
 * collective of another stream looks like to a chained launch that wants every CU. */
int resr_debug_occupy(int32_t workgroups, int32_t lds_bytes, int32_t micros, void* stream);
/* Host logic of the f16 weight-gradient launch, no GPU needed: how the (X chunk, G tile) products of `nconv` convolutions
 * that read one channel-prefix workspace (conv i: the first cin[i] channels; its own cout_pad[i] gradient channels) are
 * grouped into 2x2 jobs of the quad kernel.  out[q*4 + p] = index of the product computed by slot p of job q (products are
 * numbered conv-major, then G tile, then X chunk), -1 = slot unused.  Returns the number of jobs (<= max_jobs) or < 0. */
int resr_debug_wgrad_plan(const int32_t* cin, const int32_t* cout_pad, int32_t nconv, int32_t* out, int32_t max_jobs);
/* Measurement aid (tools/energy.py): the weight-gradient launch pair the generator's backward pass issues per RRDB -- the five
 * convolutions of `nblocks` (1..3) dense blocks, 26 products each, as ONE batched launch -- on caller-made f16 operands.
 * x_ws[b]: chunk-planar [6][n,h,w,32] (conv_k reads the first 2 + (k - 1) planes); g_ws[b]: chunk-planar [6][n,h,w,32] (planes
 * 0,1 = the closing convolution's 64 gradient channels, plane 1 + k = conv_k's 32); dw: nblocks * 26624 * 9 floats;
 * partial: `partial_bytes` of slab scratch (26 * nblocks * splits * (9 * 1024 + 32) floats). */
int resr_debug_wgrad_dense_blocks(int32_t nblocks, const void* const* x_ws, const void* const* g_ws, int32_t n, int32_t h, int32_t w,
                                  int32_t splits, float* partial, size_t partial_bytes, float* dw, void* stream);
/* test probe: lane/element map of ds_read_b64_tr_b16 (256 floats out) */
int resr_debug_tr_probe(float* out256, void* stream);

/* What THIS board sustains once it sits at its power cap (bench.py's `roofline.vs_sustained`): 256 workgroups launched back to
 * back for `seconds` (last third timed) -- mode 1: an LDS-DMA stream over `src` (`bytes` >= 64 MB of readable device memory),
 * mode 2: eight waves per workgroup issuing v_mfma_f32_32x32x16_f16 on random f16 operands, mode 3: both at once.  Returns the
 * stream's TB/s and the matrix waves' executed PFLOP/s; counter8 = 8 bytes of device scratch.  Synchronises. */
int resr_debug_sustained(int32_t mode, double seconds, const void* src, size_t bytes, void* counter8, double* stream_tbs,
                         double* matrix_pflops, void* stream);

/* In-situ kernel timing for bench.py: between begin and end every conv3x3 / wgrad launch is bracketed by HIP events
 * on its launch stream.  kernel_id = dtype*10000 + MT*100 + NT*10 + NW for conv3x3_kernel<T,MT,NT,NW>,
 * 50000 + dtype*100 + RPW for wgrad_kernel<T,RPW>.  resr_profile_end synchronises the events (host-side, test/bench
 * only), fills up to `capacity` entries and returns the number recorded. */
typedef struct {
    int32_t kernel_id;
    float ms;
    double flop;  /* algorithmic FLOP of the launch: 2*9*cin*cout*pixels */
    double bytes; /* algorithmic HBM bytes of the launch: every operand plane read / written once (no halo, no re-reads) */
} ResrProfEntry;
int resr_profile_begin(void);
int64_t resr_profile_end(ResrProfEntry* out, int64_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* RESR_DEBUG_H_ */
