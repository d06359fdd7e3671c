// api.hip -- the extern "C" boundary of libresr_hip.so (declared in include/resr.h).
// Plain pointers and sizes only; no torch / ATen types; errors become negative status codes plus a
// thread-local message.  Every entry point enqueues on the caller's stream and returns.
#include <stdarg.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "common.h"

namespace resr {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

// ---- in-situ profiling -------------------------------------------------------------------------------------
// Launches come from the Python main thread (forward) and from autograd worker threads (backward): the record list is
// guarded by a mutex, the switch is atomic, and the "before" event of a launch lives in the launching thread.
namespace {
struct ProfRec { hipEvent_t e0, e1; int id; double flop, bytes; };
std::vector<ProfRec> g_prof;
std::mutex g_prof_mu;
std::atomic<bool> g_prof_on{false};
thread_local hipEvent_t t_prof_e0 = nullptr;
}  // namespace
bool prof_on() { return g_prof_on.load(std::memory_order_relaxed); }
void prof_before(hipStream_t st) {
    if (!prof_on()) return;
    (void)hipEventCreate(&t_prof_e0);
    (void)hipEventRecord(t_prof_e0, st);
}
void prof_after(hipStream_t st, int kernel_id, double flop, double bytes) {
    if (!prof_on() || !t_prof_e0) return;
    ProfRec r;
    r.e0 = t_prof_e0; r.id = kernel_id; r.flop = flop; r.bytes = bytes;
    (void)hipEventCreate(&r.e1);
    (void)hipEventRecord(r.e1, st);
    t_prof_e0 = nullptr;
    std::lock_guard<std::mutex> lock(g_prof_mu);
    g_prof.push_back(r);
}

// Every entry point runs on the device that owns the caller's stream, whatever the calling thread's current device is
// (the boundary is used from the Python main thread and from autograd worker threads, one process per GPU or not):
// hipStreamGetDevice names it, the scope switches to it and back.  The null stream means "the current device".
struct DeviceScope {
    int prev = -1;
    explicit DeviceScope(void* stream) {
        if (!stream) return;
        hipDevice_t dev = 0;
        int cur = 0;
        if (hipStreamGetDevice((hipStream_t)stream, &dev) != hipSuccess || hipGetDevice(&cur) != hipSuccess) return;
        if ((int)dev != cur && hipSetDevice((int)dev) == hipSuccess) prev = cur;
    }
    ~DeviceScope() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};
#define RESR_DEVICE_SCOPE(stream) DeviceScope resr_device_scope_(stream)

void conv_trace_set(void*);
long long conv3x3_chain_errors();
int conv3x3_dispatch(const ResrConvDesc*, const void*, const void*, const void*, const float*, const void*,
                     const void*, const void*, void*, void*, hipStream_t);
int conv3x3_chain_dispatch(int, const ResrConvDesc*, const void*, const void*, const void* const*, const float* const*,
                           const void* const*, void* const*, void* const*, void*, size_t, hipStream_t);
size_t conv3x3_chain_state_bytes(int, int, int);
int wgrad_dispatch(const ResrWgradDesc*, const void*, const void*, const void*, float*, float*, float*, hipStream_t);
size_t wgrad_partial_bytes(const ResrWgradDesc*);
int wgrad_debug_plan(const int*, const int*, int, int*, int);
int wgrad_debug_dense_blocks(int, const void* const*, const void* const*, int, int, int, int, float*, size_t, float*, hipStream_t);
int pack_dispatch(const ResrPackChunk*, int, const float*, void*, int, hipStream_t);
int pack_mx_dispatch(const ResrPackChunk*, int, const float*, void*, hipStream_t);
size_t generator_mx_offset(const ResrGeneratorDesc*);
int ema_dispatch(float*, const float*, long, double, hipStream_t);
int nchw_to_nhwc_dispatch(const float*, void*, int, int, int, int, int, int, int, const uint8_t*, hipStream_t, long);
int nhwc_to_nchw_dispatch(const void*, float*, int, int, int, int, int, int, int, hipStream_t, long);
int sumpool2x2_dispatch(const void*, void*, const void*, int, int, int, int, int, float, hipStream_t, long, long);
size_t generator_param_count(const ResrGeneratorDesc*);
size_t generator_packed_bytes(const ResrGeneratorDesc*, int);
size_t generator_workspace_bytes(const ResrGeneratorDesc*);
size_t generator_chain_state_bytes(const ResrGeneratorDesc*);
int64_t generator_pack_table(const ResrGeneratorDesc*, int, ResrPackChunk*, int64_t);
int64_t generator_buffer_offsets(const ResrGeneratorDesc*, int64_t*, int64_t);
int generator_forward(const ResrGeneratorDesc*, const float*, const float*, const void*, void*, size_t, float*, hipStream_t);
int generator_backward(const ResrGeneratorDesc*, const float*, const float*, const void*, void*, size_t, float*, float*,
                       hipStream_t, void* const*, int);

size_t discriminator_param_count();
size_t discriminator_uv_count();
size_t discriminator_workspace_bytes(const ResrDiscriminatorDesc*);
int64_t discriminator_pack_table(const ResrDiscriminatorDesc*, const void*, ResrPackChunk*, int64_t);
int discriminator_forward(const ResrDiscriminatorDesc*, const float*, const float*, float*, const ResrPackChunk*, int, void*, size_t, float*,
                          hipStream_t);
int discriminator_backward(const ResrDiscriminatorDesc*, const float*, const float*, void*, size_t, float*, float*, hipStream_t);

int filter2d_dispatch(const float*, float*, const float*, int, int, int, int, int, int, int, hipStream_t);
int usm_dispatch(const float*, float*, float*, const float*, int, float, float, int, int, int, int, hipStream_t, int);
int resize_dispatch(const float*, float*, int, int, int, int, int, int, int, double, double, hipStream_t);
int usm_bwd_dispatch(const float*, const float*, const float*, float*, float*, const float*, int, float, int, int, int, int, hipStream_t);
int randn_dispatch(float*, long, uint64_t, uint64_t, hipStream_t);
int gauss_noise_dispatch(const float*, float*, const float*, const float*, const float*, const float*, int, int, int, int, int,
                         hipStream_t);
int poisson_noise_dispatch(const float*, float*, const float*, const float*, uint64_t, void*, int, int, int, int, int, hipStream_t);
int jpeg_dispatch(const float*, float*, const float*, float*, int, int, int, int, hipStream_t);
int quantize_crop_dispatch(const float*, const float*, float*, float*, int, int, int, int, int, int, int, int, int, int, hipStream_t);

int filter2d_u8_dispatch(const uint8_t*, uint8_t*, const int32_t*, int, int, int, int, int, int, int, hipStream_t);
int jpeg_u8_dispatch(const uint8_t*, uint8_t*, const float*, int32_t*, int, int, int, hipStream_t);
int resize_u8_dispatch(const uint8_t*, uint8_t*, int, int, int, int, int, int, int, const int32_t*, const int32_t*, const int32_t*,
                       const int32_t*, hipStream_t);

int s2d_dispatch(const void*, void*, int, int, int, int, int, int, hipStream_t);
int bilinear_up_dispatch(const void*, void*, int, int, int, int, int, int, hipStream_t, long, long);
int add_mask_dispatch(const void*, const void*, const void*, void*, long, int, float, hipStream_t);
int l1_partial_dispatch(const void*, const void*, long, int, long, float*, int, hipStream_t);
int bce_logits_const_dispatch(const float*, long, float, float, float*, float*, float*, hipStream_t);   // loss.hip
int l1_mean_dispatch(const float*, const float*, long, float, float*, float*, float*, hipStream_t);
int weighted_rows_dispatch(const float*, int, int, const float*, float*, hipStream_t);
int sustained_run(int, double, const void*, size_t, void*, double*, double*, hipStream_t);   // sustained.hip
int spectral_norm_dispatch(const float*, float*, float*, int, int, int, float, float*, float*, hipStream_t);
int spectral_norm_bwd_dispatch(const float*, const float*, const float*, const float*, const float*, float*, int, int, int, float*,
                               hipStream_t);
int fold4x4_dispatch(const float*, float*, int, int, hipStream_t);
int maxpool2x2_dispatch(const void*, void*, int, int, int, int, int, hipStream_t, long, long, uint8_t*);
int maxpool2x2_bwd_dispatch(const void*, const uint8_t*, void*, int, int, int, int, int, hipStream_t, long, long);

// probe used by tests: what does ds_read_b64_tr_b16 hand to (lane, element)?  LDS holds the element
// index at every position; lane l supplies byte address l*8.
typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 fp16x4_t;
__global__ void tr_probe_kernel(float* out) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[256];
    const int l = threadIdx.x;
    for (int i = l; i < 256; i += 64) lds[i] = (_Float16)i;
    __syncthreads();
    auto p = reinterpret_cast<__attribute__((address_space(3))) fp16x4_t*>(
        (__attribute__((address_space(3))) _Float16*)(lds + l * 4));
    const fp16x4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4f16(p);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (float)r[j];
}

// test aid: `workgroups` workgroups that each hold `lds_bytes` of LDS and spin for `micros` microseconds -- what a collective
// kernel of another stream looks like to a chained launch that needs every CU (tests/test_gpu_chain.py)
__global__ void occupy_kernel(unsigned long long ticks, unsigned* sink) {
    extern __shared__ unsigned occ_lds[];
    occ_lds[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
    if (sink && occ_lds[threadIdx.x] == 0xffffffffu) *sink = 1;
}

}  // namespace resr

using namespace resr;

extern "C" {

int resr_version(void) { return RESR_VERSION; }
const char* resr_last_error(void) { return err_buf(); }

int resr_conv3x3(const ResrConvDesc* d, const void* in0, const void* in1, const void* w_packed, const float* bias,
                 const void* res0, const void* res1, const void* mask, void* out, void* aux_out, void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return conv3x3_dispatch(d, in0, in1, w_packed, bias, res0, res1, mask, out, aux_out, (hipStream_t)stream);
}

int resr_conv3x3_chain(int32_t njobs, const ResrConvDesc* descs, const void* in0, const void* in1,
                       const void* const* packed_w, const float* const* bias, const void* const* mask,
                       void* const* out, void* const* aux_out, void* chain_state, size_t chain_state_bytes, void* stream) {
    RESR_DEVICE_SCOPE(stream);
    if (!descs || !in0 || !packed_w || !out) return fail(RESR_ERR_ARG, "conv3x3_chain: null argument");
    return conv3x3_chain_dispatch(njobs, descs, in0, in1, packed_w, bias, mask, out, aux_out, chain_state, chain_state_bytes, (hipStream_t)stream);
}

size_t resr_conv3x3_chain_state_bytes(int32_t n, int32_t h, int32_t w) { return conv3x3_chain_state_bytes(n, h, w); }

size_t resr_wgrad_partial_bytes(const ResrWgradDesc* d) { return d ? wgrad_partial_bytes(d) : 0; }

int resr_conv3x3_wgrad(const ResrWgradDesc* d, const void* x0, const void* x1, const void* g, float* partial,
                       float* dw, float* db, void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return wgrad_dispatch(d, x0, x1, g, partial, dw, db, (hipStream_t)stream);
}

int resr_pack_weights(const ResrPackChunk* chunks_dev, int32_t n_chunks, const float* arena, void* packed,
                      int32_t dtype, void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return pack_dispatch(chunks_dev, n_chunks, arena, packed, dtype, (hipStream_t)stream);
}

int resr_pack_weights_mx(const ResrPackChunk* chunks_dev, int32_t n_chunks, const float* arena, void* packed_mx, void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return pack_mx_dispatch(chunks_dev, n_chunks, arena, packed_mx, (hipStream_t)stream);
}

int resr_nchw_to_nhwc(const float* src, void* dst, int32_t n, int32_t c, int32_t h, int32_t w, int32_t unshuffle,
                      int32_t c_pad, int32_t dtype, const uint8_t* mask, void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return nchw_to_nhwc_dispatch(src, dst, n, c, h, w, unshuffle, c_pad, dtype, mask, (hipStream_t)stream, -1L);
}

int resr_nhwc_to_nchw(const void* src, float* dst, int32_t n, int32_t c, int32_t h, int32_t w, int32_t shuffle,
                      int32_t src_stride, int32_t dtype, void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return nhwc_to_nchw_dispatch(src, dst, n, c, h, w, shuffle, src_stride, dtype, (hipStream_t)stream, -1L);
}

int resr_sumpool2x2(const void* src, void* dst, const void* mask, int32_t n, int32_t h_out, int32_t w_out, int32_t c,
                    int32_t dtype, float slope, void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return sumpool2x2_dispatch(src, dst, mask, n, h_out, w_out, c, dtype, slope, (hipStream_t)stream, -1L, -1L);
}

size_t resr_generator_param_count(const ResrGeneratorDesc* d) { return generator_param_count(d); }
size_t resr_generator_packed_bytes(const ResrGeneratorDesc* d, int32_t backward) { return generator_packed_bytes(d, backward); }
size_t resr_generator_mx_offset(const ResrGeneratorDesc* d) { return generator_mx_offset(d); }
size_t resr_generator_workspace_bytes(const ResrGeneratorDesc* d) { return generator_workspace_bytes(d); }
size_t resr_generator_chain_state_bytes(const ResrGeneratorDesc* d) { return generator_chain_state_bytes(d); }
int64_t resr_generator_pack_table(const ResrGeneratorDesc* d, int32_t backward, ResrPackChunk* chunks, int64_t capacity) {
    return generator_pack_table(d, backward, chunks, capacity);
}

int64_t resr_generator_buffer_offsets(const ResrGeneratorDesc* d, int64_t* out, int64_t capacity) {
    return generator_buffer_offsets(d, out, capacity);
}

int resr_generator_forward(const ResrGeneratorDesc* d, const float* x_nchw, const float* params, const void* packed,
                           void* workspace, size_t workspace_bytes, float* y_nchw, void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return generator_forward(d, x_nchw, params, packed, workspace, workspace_bytes, y_nchw, (hipStream_t)stream);
}

int resr_generator_backward(const ResrGeneratorDesc* d, const float* gy_nchw, const float* params, const void* packed,
                            void* workspace, size_t workspace_bytes, float* grad_params, float* gx_nchw, void* stream,
                            void* const* grad_ready_events, int32_t n_events) {
    RESR_DEVICE_SCOPE(stream);
    return generator_backward(d, gy_nchw, params, packed, workspace, workspace_bytes, grad_params, gx_nchw,
                              (hipStream_t)stream, grad_ready_events, n_events);
}

size_t resr_discriminator_param_count(void) { return discriminator_param_count(); }
size_t resr_discriminator_uv_count(void) { return discriminator_uv_count(); }
size_t resr_discriminator_workspace_bytes(const ResrDiscriminatorDesc* d) { return discriminator_workspace_bytes(d); }
int64_t resr_discriminator_pack_table(const ResrDiscriminatorDesc* d, const void* workspace, ResrPackChunk* chunks, int64_t capacity) {
    return discriminator_pack_table(d, workspace, chunks, capacity);
}

int resr_discriminator_forward(const ResrDiscriminatorDesc* d, const float* x_nchw, const float* params, float* uv,
                               const ResrPackChunk* table_dev, int32_t n_chunks, void* workspace, size_t workspace_bytes,
                               float* y_nchw, void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return discriminator_forward(d, x_nchw, params, uv, table_dev, n_chunks, workspace, workspace_bytes, y_nchw, (hipStream_t)stream);
}

int resr_discriminator_backward(const ResrDiscriminatorDesc* d, const float* gy_nchw, const float* params, void* workspace,
                                size_t workspace_bytes, float* grad_params, float* gx_nchw, void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return discriminator_backward(d, gy_nchw, params, workspace, workspace_bytes, grad_params, gx_nchw, (hipStream_t)stream);
}

int resr_ema_update(float* shadow, const float* params, int64_t count, double decay, void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return ema_dispatch(shadow, params, (long)count, decay, (hipStream_t)stream);
}

int resr_filter2d(const float* src, float* dst, const float* kernel, int32_t n, int32_t c, int32_t h, int32_t w, int32_t kh,
                  int32_t kw, int32_t per_sample, void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return filter2d_dispatch(src, dst, kernel, n, c, h, w, kh, kw, per_sample, (hipStream_t)stream);
}

int resr_usm_sharp(const float* src, float* dst, float* tmp3, const float* k1d, int32_t ksize, float weight, float threshold,
                   int32_t n, int32_t c, int32_t h, int32_t w, void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return usm_dispatch(src, dst, tmp3, k1d, ksize, weight, threshold, n, c, h, w, (hipStream_t)stream, 1);
}

int resr_usm_sharp_forward_only(const float* src, float* dst, float* tmp3, const float* k1d, int32_t ksize, float weight, float threshold,
                                int32_t n, int32_t c, int32_t h, int32_t w, void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return usm_dispatch(src, dst, tmp3, k1d, ksize, weight, threshold, n, c, h, w, (hipStream_t)stream, 0);
}

int resr_usm_sharp_bwd(const float* x, const float* saved_tmp3, const float* g, float* gx, float* tmp2, const float* k1d,
                       int32_t ksize, float weight, int32_t n, int32_t c, int32_t h, int32_t w, void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return usm_bwd_dispatch(x, saved_tmp3, g, gx, tmp2, k1d, ksize, weight, n, c, h, w, (hipStream_t)stream);
}

int resr_resize(const float* src, float* dst, int32_t n, int32_t c, int32_t h, int32_t w, int32_t oh, int32_t ow, int32_t mode,
                double scale_h, double scale_w, void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return resize_dispatch(src, dst, n, c, h, w, oh, ow, mode, scale_h, scale_w, (hipStream_t)stream);
}

int resr_randn_fill(float* dst, int64_t count, uint64_t seed, uint64_t stream_id, void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return randn_dispatch(dst, (long)count, seed, stream_id, (hipStream_t)stream);
}

int resr_noise_gaussian(const float* src, float* dst, const float* sigma, const float* gray, const float* field_gray,
                        const float* field_color, int32_t n, int32_t c, int32_t h, int32_t w, int32_t clip, void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return gauss_noise_dispatch(src, dst, sigma, gray, field_gray, field_color, n, c, h, w, clip, (hipStream_t)stream);
}

size_t resr_noise_poisson_workspace_bytes(int32_t n) { return (size_t)n * (512 * sizeof(unsigned) + 2 * sizeof(float)); }

int resr_noise_poisson(const float* src, float* dst, const float* scale, const float* gray, uint64_t seed, void* workspace,
                       int32_t n, int32_t c, int32_t h, int32_t w, int32_t clip, void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return poisson_noise_dispatch(src, dst, scale, gray, seed, workspace, n, c, h, w, clip, (hipStream_t)stream);
}

int resr_jpeg(const float* src, float* dst, const float* quality, float* coeffs, int32_t n, int32_t h, int32_t w, int32_t flags,
              void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return jpeg_dispatch(src, dst, quality, coeffs, n, h, w, flags, (hipStream_t)stream);
}

int resr_quantize_crop(const float* lr, const float* hr, float* lr_out, float* hr_out, int32_t n, int32_t c, int32_t lr_h,
                       int32_t lr_w, int32_t hr_h, int32_t hr_w, int32_t hr_size, int32_t upscale, int32_t hr_top,
                       int32_t hr_left, void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return quantize_crop_dispatch(lr, hr, lr_out, hr_out, n, c, lr_h, lr_w, hr_h, hr_w, hr_size, upscale, hr_top, hr_left,
                                  (hipStream_t)stream);
}

int resr_filter2d_u8(const uint8_t* src, uint8_t* dst, const int32_t* taps_q14, int32_t n, int32_t c, int32_t h, int32_t w,
                     int32_t kh, int32_t kw, int32_t per_sample, void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return filter2d_u8_dispatch(src, dst, taps_q14, n, c, h, w, kh, kw, per_sample, (hipStream_t)stream);
}

int resr_resize_u8(const uint8_t* src, uint8_t* dst, int32_t n, int32_t c, int32_t h, int32_t w, int32_t oh, int32_t ow,
                   int32_t mode, const int32_t* idx_y, const int32_t* w_y, const int32_t* idx_x, const int32_t* w_x, void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return resize_u8_dispatch(src, dst, n, c, h, w, oh, ow, mode, idx_y, w_y, idx_x, w_x, (hipStream_t)stream);
}

int resr_jpeg_u8(const uint8_t* src, uint8_t* dst, const float* quality, int32_t* coeffs, int32_t n, int32_t h, int32_t w,
                 void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return jpeg_u8_dispatch(src, dst, quality, coeffs, n, h, w, (hipStream_t)stream);
}

int resr_space_to_depth(const void* src, void* dst, int32_t n, int32_t h, int32_t w, int32_t c, int32_t dtype, int32_t inverse,
                        void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return s2d_dispatch(src, dst, n, h, w, c, dtype, inverse, (hipStream_t)stream);
}

int resr_bilinear_up2x(const void* src, void* dst, int32_t n, int32_t h, int32_t w, int32_t c, int32_t dtype, int32_t backward,
                       void* stream) {
    RESR_DEVICE_SCOPE(stream);
    // RESR_F16X2: src and dst are (hi, lo) pairs, each lo tensor directly behind its hi tensor
    const long px = (long)n * h * w * c;
    const bool x2 = dtype == RESR_F16X2;
    return bilinear_up_dispatch(src, dst, n, h, w, c, dtype, backward, (hipStream_t)stream, x2 ? (backward ? 4 * px : px) : 0L, x2 ? (backward ? px : 4 * px) : 0L);
}

int resr_add_mask(const void* a, const void* b, const void* mask, void* out, int64_t count, int32_t dtype, float slope,
                  void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return add_mask_dispatch(a, b, mask, out, (long)count, dtype, slope, (hipStream_t)stream);
}

int resr_l1_partial(const void* a, const void* b, int64_t count, int32_t dtype, int64_t lo_offset, float* partial, int32_t nblocks, void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return l1_partial_dispatch(a, b, (long)count, dtype, (long)lo_offset, partial, nblocks, (hipStream_t)stream);
}

int resr_debug_sustained(int32_t mode, double seconds, const void* src, size_t bytes, void* counter8, double* stream_tbs, double* matrix_pflops,
                         void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return sustained_run(mode, seconds, src, bytes, counter8, stream_tbs, matrix_pflops, (hipStream_t)stream);
}

size_t resr_loss_scratch_bytes(void) { return (1 + 1024) * sizeof(float); }

int resr_bce_logits_const(const float* logits, int64_t count, float label, float weight, float* loss, float* grad, float* scratch,
                          void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return bce_logits_const_dispatch(logits, (long)count, label, weight, loss, grad, scratch, (hipStream_t)stream);
}

int resr_l1_mean(const float* a, const float* b, int64_t count, float weight, float* loss, float* grad_a, float* scratch, void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return l1_mean_dispatch(a, b, (long)count, weight, loss, grad_a, scratch, (hipStream_t)stream);
}

int resr_weighted_row_sums(const float* partial, int32_t rows, int32_t cols, const float* coef_host, float* out, void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return weighted_rows_dispatch(partial, rows, cols, coef_host, out, (hipStream_t)stream);
}

int resr_spectral_norm(const float* w, float* u, float* v, int32_t rows, int32_t cols, int32_t training, float eps, float* sigma2,
                       float* tmp, void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return spectral_norm_dispatch(w, u, v, rows, cols, training, eps, sigma2, tmp, (hipStream_t)stream);
}

int resr_spectral_norm_bwd(const float* g, const float* w, const float* u, const float* v, const float* sigma2, float* dst,
                           int32_t rows, int32_t cols, int32_t accumulate, float* tmp1, void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return spectral_norm_bwd_dispatch(g, w, u, v, sigma2, dst, rows, cols, accumulate, tmp1, (hipStream_t)stream);
}

int resr_maxpool2x2(const void* src, void* dst, int32_t n, int32_t h_out, int32_t w_out, int32_t c, int32_t dtype, void* stream) {
    RESR_DEVICE_SCOPE(stream);
    const long px = (long)n * h_out * w_out * c;
    const bool x2 = dtype == RESR_F16X2;
    return maxpool2x2_dispatch(src, dst, n, h_out, w_out, c, dtype, (hipStream_t)stream, x2 ? 4 * px : 0L, x2 ? px : 0L, nullptr);
}

int resr_maxpool2x2_arg(const void* src, void* dst, uint8_t* arg, int32_t n, int32_t h_out, int32_t w_out, int32_t c, int32_t dtype,
                        void* stream) {
    RESR_DEVICE_SCOPE(stream);
    const long px = (long)n * h_out * w_out * c;
    const bool x2 = dtype == RESR_F16X2;
    return maxpool2x2_dispatch(src, dst, n, h_out, w_out, c, dtype, (hipStream_t)stream, x2 ? 4 * px : 0L, x2 ? px : 0L, arg);
}

int resr_maxpool2x2_bwd(const void* g, const uint8_t* arg, void* gin, int32_t n, int32_t h_out, int32_t w_out, int32_t c, int32_t dtype,
                        void* stream) {
    RESR_DEVICE_SCOPE(stream);
    const long px = (long)n * h_out * w_out * c;
    const bool x2 = dtype == RESR_F16X2;
    return maxpool2x2_bwd_dispatch(g, arg, gin, n, h_out, w_out, c, dtype, (hipStream_t)stream, x2 ? px : 0L, x2 ? 4 * px : 0L);
}

int resr_fold4x4(const float* dw3, float* dw4, int32_t cout, int32_t c, void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return fold4x4_dispatch(dw3, dw4, cout, c, (hipStream_t)stream);
}

int resr_profile_begin(void) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    for (auto& r : g_prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    g_prof.clear();
    g_prof_on = true;
    return RESR_OK;
}

int64_t resr_profile_end(ResrProfEntry* out, int64_t capacity) {
    g_prof_on = false;
    std::lock_guard<std::mutex> lock(g_prof_mu);
    int64_t n = 0;
    for (auto& r : g_prof) {
        (void)hipEventSynchronize(r.e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.e0, r.e1);
        if (out && n < capacity) { out[n].kernel_id = r.id; out[n].ms = ms; out[n].flop = r.flop; out[n].bytes = r.bytes; }
        ++n;
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    g_prof.clear();
    return n;
}

// host logic probe (no GPU): 2x2 grouping of a weight-gradient launch's products
int resr_debug_wgrad_dense_blocks(int32_t nblocks, const void* const* x_ws, const void* const* g_ws, int32_t n, int32_t h, int32_t w, int32_t splits,
                                  float* partial, size_t partial_bytes, float* dw, void* stream) {
    RESR_DEVICE_SCOPE(stream);
    return wgrad_debug_dense_blocks(nblocks, x_ws, g_ws, n, h, w, splits, partial, partial_bytes, dw, (hipStream_t)stream);
}

int resr_debug_wgrad_plan(const int32_t* cin, const int32_t* cout_pad, int32_t nconv, int32_t* out, int32_t max_jobs) {
    return wgrad_debug_plan(cin, cout_pad, nconv, out, max_jobs);
}

// debug: device buffer of 32*2*64 uint64 receiving s_memrealtime stamps of the next conv launches (null = off)
int resr_debug_conv_trace(void* dev_buf) {
    resr::conv_trace_set(dev_buf);
    return RESR_OK;
}

int64_t resr_chain_errors(void) { return (int64_t)resr::conv3x3_chain_errors(); }

int64_t resr_debug_chain_errors(void) {
    (void)hipDeviceSynchronize();   // test / end-of-run entry: every launch enqueued so far has reported
    return (int64_t)resr::conv3x3_chain_errors();
}

int resr_debug_occupy(int32_t workgroups, int32_t lds_bytes, int32_t micros, void* stream) {
    RESR_DEVICE_SCOPE(stream);
    if (workgroups <= 0 || lds_bytes < 256 || lds_bytes > 160 * 1024 || micros <= 0) return fail(RESR_ERR_ARG, "debug_occupy: bad argument");
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(occupy_kernel, dim3(workgroups), dim3(64), (size_t)lds_bytes, (hipStream_t)stream, (unsigned long long)micros * 100ull, (unsigned*)nullptr);
    RESR_CHECK_LAUNCH("occupy_kernel");
    return RESR_OK;
}

int resr_debug_tr_probe(float* out256, void* stream) {
    RESR_DEVICE_SCOPE(stream);
    hipLaunchKernelGGL(tr_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out256);
    RESR_CHECK_LAUNCH("tr_probe_kernel");
    return RESR_OK;
}

}  // extern "C"
