// Shared pieces of the 3x3 convolution kernels (conv3x3.hip: register-staged tiles; conv3x3_ws.hip: persistent,
// producer/consumer wave-specialised tiles).
#pragma once
#include "common.h"

namespace resr {

struct ConvArgs {
    const char* in0;
    const char* in1;
    const char* w;
    const float* bias;
    const char* res0;
    const char* res1;
    const char* mask;
    char* out;
    uint8_t* aux;
    int n, h, w_, hs, ws;
    int cin, cin0;
    int in0_stride_b, in1_stride_b;  // bytes per pixel
    int cout;
    int out_stride, res0_stride, res1_stride, mask_stride;  // elements
    size_t in0_chunk_b, in1_chunk_b;                        // bytes between consecutive 32-channel chunks
    int out_chunk, res0_chunk, res1_chunk, mask_chunk;      // elements between consecutive 32-channel chunks
    int flags;
    float s0, t0, s1, t1, slope;
    int tiles_x, tiles_y;
    // RESR_F16X2: byte offsets hi -> lo of the two input segments, element offsets hi -> lo of out / residuals
    size_t in0_lo_b, in1_lo_b;
    long out_lo, res0_lo, res1_lo;   // out_lo = 0: the output is a single f16 tensor (RESR_CONV_OUT_SINGLE), no lo store
    long mask_lo;                    // a mask given as a saved activation (RESR_CONV_MASK without _BITS): its hi -> lo offset
                                     // (ResrConvDesc.mask_lo_offset); read only where a hi value is zero (common.h pair_positive); 0 = no lo tensor
    // RESR_F16X2: the first pair_chunks 32-channel input chunks are hi/lo pairs (three stages each), the chunks behind them single
    // f16 tensors (two stages: x W0 + x W1); >= cin / 32 = every chunk a pair (ResrConvDesc.x2_pair_chunks)
    int pair_chunks, out_single;
    int single_stages;               // RESR_F16X2: 1 = the single chunks take ONE stage (x W0 alone: RESR_CONV_SINGLE_W16), else two
    // RESR_CONV_MX_PAIRS (kernel instantiations X2 = 2): a pair chunk is an f16 stage (x_hi W0) + an MX stage on 8-bit operands
    // (conv3x3_ws.h).  Byte offsets hi tensor -> q tensor of the two input segments (same strides as the hi tensor), element offset
    // hi -> q of the output (0: no q tensor is written), this convolution's MX weight blocks (one WBUF-sized block per real chunk)
    int mx;
    size_t in0_q_b, in1_q_b;
    long out_q;
    const char* w_mx;
    // sparse taps of a 4x4 / stride-2 convolution run over the space-to-depth image (conv3x3_ws.h, SP): channels per
    // sub-position of the input (forward), sub-position of this launch's output group (backward-data)
    int s2d_c, tap_c;
    // output groups per launch (cout 64 shape): group g reads its packed weights at w + g * w_group_b and writes / reads
    // its epilogue operands 64 channels further inside the pixel; no bias, no sign-bit tensors, NHWC output
    int ngroups;
    size_t w_group_b;
    const char* zero;  // 16 zero bytes in device memory (source of out-of-image LDS-DMA lanes)
    unsigned long long* trace;  // debug: per-workgroup s_memrealtime stamps (resr_debug_conv_trace), else null
};

// A chain of dependent cout-32 convolutions of one dense block in ONE persistent launch (conv3x3_ws.h, CH): job j reads
// the plane prefix of the shared operands and writes its own plane, which is the LAST 32-channel chunk of job j+1's input.
// Everything a job does not share with the others:
struct ChainJob {
    const char* w;        // packed weights
    const float* bias;    // or null
    char* out;            // the job's output plane
    void* aux;            // sign words written (forward) / read as the mask (backward-data)
    int cin;              // input channels: a prefix of the shared in0 [+ in1] planes
    int dep;              // the job whose output plane is this job's LAST input chunk (-1: none)
    // kind 3: one 32-channel half of the block's closing convolution (conv5, model.py:94-96, or its mirrored backward-data
    // pass) instead of a growth convolution: no LeakyReLU / sign words / mask, v = v * s0 + t0 * res0 [, v * s1 + t1 * res1]
    int kind;             // 0: the launch's own epilogue kind (EPI 0 / 16 / 33), 3: residual half
    int w_mt, w_m;        // packed weights are laid out for w_mt (1 or 2) output tiles; this job multiplies tile w_m
    int pad_;
    const char* res0;     // kind 3: residual planes (pixel stride 32 elements) and their scales
    const char* res1;     // or null
    float s0, t0, s1, t1;
    long out_lo, res0_lo, res1_lo;   // RESR_F16X2: element offsets hi -> lo tensor of out / res0 / res1
    const char* w_mx;     // RESR_CONV_MX_PAIRS: the job's MX weight blocks (laid out for w_mt tiles like w)
    long out_q;           // ... and the element offset hi -> q tensor of its output (0: none)
};
constexpr int kMaxChain = 6;
constexpr int kMaxBiasGroups = 8;   // output-group launches WITH a bias keep <= 8 x 64 bias values in LDS (conv3x3_ws.h, GB_OFF)
// Device-side state of the chained launches: memory the CALLER owns (part of its workspace, zero-filled once before the
// first use; include/resr.h resr_conv3x3_chain_state_bytes) -- the library neither allocates nor keeps it.  Words:
//   [0] epoch      flag value of "job j of the current launch done" = epoch + j + 1; the last workgroup of a launch to
//                  finish adds 8 (flags only ever grow: nothing is reset between launches; near the wrap of the signed
//                  comparison that workgroup zeroes the flags and starts over)
//   [1] finished   workgroups of the current launch that have finished (the last one resets it and the tickets)
//   [2] time-outs  flag polls that gave up            [3] workgroups an XCD received beyond its share of the grid
//   [8..15]        per-XCD workgroup tickets of the current launch
//   [16 ...]       flags[tile]: per-tile progress
constexpr int kChainHdr = 16;
struct ChainArgs {
    int njobs;
    unsigned cap;            // flag words behind the header
    unsigned* state;         // see above
    unsigned* host_errors;   // [2] host-mapped copies of words 2 / 3: the host reads them without synchronising
    const char* nan16;       // 16 bytes of f16 NaNs in device memory: a dependent halo whose poll timed out is read from here,
                             // so a broken launch poisons its output (the loss turns NaN) instead of computing on stale planes
    // CH 3 (pinned pipeline, experiment): the workgroups of an XCD with index [split[j], split[j+1]) run job j only
    int split[kMaxChain + 2];
    ChainJob job[kMaxChain];
};
struct ChainNone {};

// Algorithmic HBM bytes of one pass: input channels + output (+ mask, residuals, aux) once per pixel.
// (es = 4 for RESR_F16X2 pairs: single f16 chunks / a single f16 output count 2 bytes per element)
inline double conv_algorithmic_bytes(const ConvArgs& a, size_t es) {
    const double px_out = (double)a.n * a.h * a.w_, px_in = (double)a.n * a.hs * a.ws;
    const int pair_ch = (es == 4 && a.pair_chunks > 0 && a.pair_chunks * 32 < a.cin) ? a.pair_chunks * 32 : a.cin;
    double b = px_in * (pair_ch * (double)es + (a.cin - pair_ch) * 2.0);
    const bool nchw = a.flags & RESR_CONV_OUT_NCHW_F32;
    b += px_out * a.cout * (nchw ? 4 : (es == 4 && a.out_single) ? 2 : es);
    if (a.flags & RESR_CONV_MASK) b += px_out * ((a.flags & RESR_CONV_MASK_BITS) ? ((a.cout + 31) / 32) * 4.0 : a.cout * (double)es);
    if (a.res0) b += px_out * a.cout * es;
    if (a.res1) b += px_out * a.cout * es;
    if (a.out_q) b += px_out * a.cout * 2.0;   // the q tensor of a pair output: one byte per hi and per lo value
    if (a.aux) b += px_out * ((a.flags & RESR_CONV_WRITE_SIGNBITS) ? ((a.cout + 31) / 32) * 4.0 : nchw ? a.cout : a.cout * (double)es);
    return b;
}

template <int SPP>
__device__ __forceinline__ int swz(int hx) {
    // SPP slots per pixel; 16/SPP consecutive pixels fill one 256-byte bank row.
    if constexpr (SPP == 4) return (hx >> 2) & 3;
    else return (hx >> 1) & 7;
}

template <typename T>
struct Frag;
template <>
struct Frag<half_t> {
    static __device__ __forceinline__ float16v mma(const uint4& a, const uint4& b, float16v c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a),
                                                      __builtin_bit_cast(half8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ float ld(const char* p, int idx) {
        return (float)reinterpret_cast<const half_t*>(p)[idx];
    }
};
template <>
struct Frag<float> {
    static __device__ __forceinline__ float16v mma(const uint4& a, const uint4& b, float16v c) {
        const float4v fa = __builtin_bit_cast(float4v, a), fb = __builtin_bit_cast(float4v, b);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0], fb[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1], fb[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[2], fb[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[3], fb[3], c, 0, 0, 0);
        return c;
    }
    static __device__ __forceinline__ float ld(const char* p, int idx) {
        return reinterpret_cast<const float*>(p)[idx];
    }
};

template <typename T>
__device__ __forceinline__ void load4(const char* base, size_t idx, float v[4]) {
    if constexpr (sizeof(T) == 2) {
        const half4 h = *reinterpret_cast<const half4*>(base + idx * 2);
        v[0] = (float)h[0]; v[1] = (float)h[1]; v[2] = (float)h[2]; v[3] = (float)h[3];
    } else {
        const float4v f = *reinterpret_cast<const float4v*>(base + idx * 4);
        v[0] = f[0]; v[1] = f[1]; v[2] = f[2]; v[3] = f[3];
    }
}

template <typename T>
__device__ __forceinline__ void store4(char* base, size_t idx, const float v[4]) {
    if constexpr (sizeof(T) == 2) {
        half4 h;
        h[0] = (half_t)v[0]; h[1] = (half_t)v[1]; h[2] = (half_t)v[2]; h[3] = (half_t)v[3];
        *reinterpret_cast<half4*>(base + idx * 2) = h;
    } else {
        float4v f;
        f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
        *reinterpret_cast<float4v*>(base + idx * 4) = f;
    }
}


}  // namespace resr
