// Shared host/device helpers for libresr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "resr.h"
#include "resr_debug.h"

namespace resr {

typedef _Float16 half_t;
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
typedef float float16v __attribute__((ext_vector_type(16)));

// thread-local error text for resr_last_error()
char* err_buf();
int fail(int code, const char* fmt, ...);

#define RESR_CHECK_LAUNCH(name)                                                        \
    do {                                                                               \
        hipError_t e_ = hipGetLastError();                                             \
        if (e_ != hipSuccess) return fail(RESR_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e_)); \
    } while (0)

// optional in-situ kernel timing (resr_profile_begin/end): HIP events around individual launches on the launch stream
bool prof_on();
void prof_before(hipStream_t st);
void prof_after(hipStream_t st, int kernel_id, double flop, double bytes = 0.0);

// bytes per stored element (RESR_F16X2: of one f16 tensor of the hi/lo pair) / tensors per activation
inline size_t elem_size(int dtype) { return dtype == RESR_F32 ? 4 : 2; }
inline size_t act_tensors(int dtype) { return dtype == RESR_F16X2 ? 2 : 1; }
// RESR_F16X2: value = hi + lo * kLoInv, lo = (value - hi) * kLoScale; packed weights carry the factor kLoScale
constexpr float kLoScale = 4096.f, kLoInv = 1.f / 4096.f;
constexpr int kMaxDevices = 16;   // per-device caches (occupancy, zero pages, CU counts) are indexed by hipGetDevice()
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Power-of-two gradient pre-scale of a backward pass (generator.hip, RESR_F16X2).  absmax_dispatch (layout.hip) leaves the bits of
// m = max |g_y| * 2^-t; when m < 1 the pass runs on g_y * s with s = 2^-floor(log2 m) -- max |g_y * s| in [2^t, 2^(t+1)): f16's
// normal range however small the caller's loss scale is -- and every result leaves through * 1/s.  Both factors are exact.  Gradients
// that are large enough already (m >= 1) are left alone (s = 1): a GradScaler that keeps doubling its scale still meets f16's
// overflow and settles below it, as it does without the pre-scale.  Zero, denormal or non-finite maxima give s = 1.
// The slot is a block of words in memory the caller zero-fills once (the head of a generator / discriminator workspace):
//   [0] the bits of m, rewritten by every pass (absmax_dispatch)
//   [1] back-off b (sticky): the lift aims 2^b lower -- m is stored times 2^b -- once a LIFTED pass has produced a non-finite weight
//       gradient: a gradient that grows by more than the 2^9 of headroom on its way back overflows f16 at EVERY loss scale when the lift
//       re-raises it each step, so a GradScaler backing off could no longer cure it (found_inf would fire until its scale decays to
//       nothing).  Every such pass lowers the target by 2^4 (up to 2^40: then the lift is effectively off and the caller's scale rules,
//       as before round 5); the pass that overflowed is the GradScaler's to skip, the next one runs with headroom.
//   [2] flag: set by the weight-gradient reductions of a lifted pass that wrote a non-finite value; consumed by the next absmax_dispatch
constexpr int kPrescaleBackoffStep = 4, kPrescaleBackoffMax = 40;
__host__ __device__ __forceinline__ bool grad_prescale_lifted(unsigned amax_bits) {
    const unsigned e = (amax_bits >> 23) & 0xffu;
    return e != 0u && e < 127u;
}
__host__ __device__ __forceinline__ float grad_prescale(unsigned amax_bits, bool inverse) {
    const unsigned e = (amax_bits >> 23) & 0xffu;          // biased exponent of m
    if (e == 0u || e >= 127u) return 1.f;
    const unsigned f = (inverse ? e : 254u - e) << 23;     // 2^(e - 127) or 2^(127 - e)
    return __builtin_bit_cast(float, f);
}

// RESR_F16X2: "is this saved activation positive" for a LeakyReLU-backward mask read from a (hi, lo) pair.  hi decides unless it
// rounded to zero (|v| < 2^-25, below f16's subnormals); then the lo tensor (v * 2^12) carries the sign.
__device__ __forceinline__ bool pair_positive(half_t hi, half_t lo) {
    return (float)hi > 0.f || ((float)hi == 0.f && (float)lo > 0.f);
}

// MI355X: blocks are dealt round-robin to the 8 XCDs (block b -> XCD b % 8).  Give every XCD a
// contiguous range of tiles so neighbouring tiles (shared halo rows, same weights) meet in one L2.
// Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = b & 7, idx = b >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

}  // namespace resr
