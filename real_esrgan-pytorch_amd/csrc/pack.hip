// pack.hip -- parameter-side kernels: weight packing into MFMA A-fragment order, EMA update.
//
// resr_pack_weights: the module keeps the reference's OIHW fp32 parameters (state_dict surface,
// SURVEY.md §8b).  Every optimiser step they are re-packed (one launch for all 351 convs, both the
// forward and the backward-data forms) into the order conv3x3.hip streams them:
//     chunk (32 K-channels) -> tap (9) -> k-step -> M-tile -> lane (kh*32 + m) -> 16 bytes
// so that a wave's A-fragment is one contiguous, coalesced 1 KiB load.
// Backward-data chunks are gathered transposed (M = cin, K = cout) with flipped taps and an optional
// scale (the 0.2 residual scalings of model.py:95,129 folded into the weights).
#include "common.h"

namespace resr {

//
// RESR_F16X2 (X2 = true): a chunk becomes THREE consecutive f16 blocks of the plain chunk's size, in the stage order of
// the conv kernel: W0 = f16(w * 2^12) (multiplies x_hi), W1 = f16(w * 2^12 - W0) (x_hi again), W2 = f16(W0 * 2^-12)
// (multiplies x_lo, which is stored times 2^12).  The table's dst_off counts elements of the *plain* layout; the
// kernel triples it.
template <typename T, bool X2 = false>
__global__ __launch_bounds__(256) void pack_kernel(const ResrPackChunk* __restrict__ chunks,
                                                   const float* __restrict__ arena, T* __restrict__ packed) {
    constexpr int E = 16 / (int)sizeof(T);
    constexpr int KS = 32 / E / 2;
    const ResrPackChunk c = chunks[blockIdx.x];
    const int total = 9 * c.mt * 1024;
    T* dst = packed + c.dst_off * (X2 ? 3 : 1);
    const float* src = arena + c.src_off;
    const float sc = c.scale * (c.scale_ptr ? *c.scale_ptr : 1.f);
    for (int idx = threadIdx.x; idx < total; idx += 256) {
        int r = idx;
        const int e = r % E; r /= E;
        const int m = r % 32; r /= 32;
        const int kh = r % 2; r /= 2;
        const int mt = r % c.mt; r /= c.mt;
        const int ks = r % KS; r /= KS;
        const int tap = r;
        const int k = (ks * 2 + kh) * E + e;
        const int mm = mt * 32 + m;
        float v = 0.f;
        if (mm < c.m_count && k < c.k_count) {
            const int co = c.transposed ? c.k_off + k : c.m_off + mm;
            const int ci = c.transposed ? c.m_off + mm : c.k_off + k;
            const int t = c.transposed ? 8 - tap : tap;
            if (!c.virtual4x4) {
                v = src[((size_t)co * c.src_cin + ci) * 9 + t];
            } else {                                    // 4x4 stride-2 kernel as a 3x3 kernel over space-to-depth
                const int C = c.src_cin, sub = ci / C, ch = ci % C;
                const int ky = 2 * (t / 3) + (sub >> 1) - 1, kx = 2 * (t % 3) + (sub & 1) - 1;
                if (ky >= 0 && ky < 4 && kx >= 0 && kx < 4) v = src[((size_t)co * C + ch) * 16 + ky * 4 + kx];
            }
            v *= sc;
        }
        if constexpr (X2) {
            const float t = v * kLoScale;
            const T w0 = (T)t;
            dst[idx] = w0;
            dst[total + idx] = (T)(t - (float)w0);
            dst[2 * total + idx] = (T)((float)w0 * kLoInv);
        } else {
            dst[idx] = (T)v;
        }
    }
}

int pack_dispatch(const ResrPackChunk* chunks_dev, int n_chunks, const float* arena, void* packed,
                  int dtype, hipStream_t stream) {
    if (!chunks_dev || !arena || !packed || n_chunks <= 0) return fail(RESR_ERR_ARG, "pack_weights: bad argument");
    if (dtype == RESR_F16)
        hipLaunchKernelGGL(pack_kernel<half_t>, dim3(n_chunks), dim3(256), 0, stream, chunks_dev, arena, (half_t*)packed);
    else if (dtype == RESR_F32)
        hipLaunchKernelGGL(pack_kernel<float>, dim3(n_chunks), dim3(256), 0, stream, chunks_dev, arena, (float*)packed);
    else if (dtype == RESR_F16X2)
        hipLaunchKernelGGL((pack_kernel<half_t, true>), dim3(n_chunks), dim3(256), 0, stream, chunks_dev, arena, (half_t*)packed);
    else
        return fail(RESR_ERR_ARG, "pack_weights: dtype=%d", dtype);
    RESR_CHECK_LAUNCH("pack_kernel");
    return RESR_OK;
}

// f16 -> bf8 (e5m2), round to nearest even: e5m2 is f16's sign, exponent and two leading significand bits, so the conversion rounds
// the 16-bit pattern to its upper byte (carries run into the exponent as they should; f16 subnormals become e5m2 subnormals).
// The conv epilogue makes its q records with v_cvt_scalef32_pk_bf8_f16 at unit scale: the same function (tests/test_gpu_mx.py).
__host__ __device__ __forceinline__ unsigned f16_bits_to_bf8(unsigned h) {
    return ((h + 0x7fu + ((h >> 8) & 1u)) >> 8) & 0xffu;
}

// resr_pack_weights_mx: the MX block of a chunk (RESR_CONV_MX_PAIRS; conv3x3_ws.h, X2 = 2) -- per tap and output row 64 bytes
// [bf8(W1[k]) k = 0..31 | bf8(W2[k])] of the exact16 split above, as A fragments of v_mfma_scale_f32_32x32x64_f8f6f4: 16-byte piece
// ((tap * 2 + p) * mt + tile) * 64 + h * 32 + row holds K = 16 h .. 16 h + 15 of block p (0: W1, 1: W2).  One block = 9 x mt x 2 KB
// = the bytes of one plain f16 block, at packed_mx + 2 * dst_off.
__global__ __launch_bounds__(256) void pack_mx_kernel(const ResrPackChunk* __restrict__ chunks, const float* __restrict__ arena,
                                                      unsigned char* __restrict__ packed) {
    const ResrPackChunk c = chunks[blockIdx.x];
    const int total = 9 * c.mt * 2048;   // bytes
    unsigned char* dst = packed + c.dst_off * 2;
    const float* src = arena + c.src_off;
    const float sc = c.scale * (c.scale_ptr ? *c.scale_ptr : 1.f);
    for (int idx = threadIdx.x; idx < total; idx += 256) {
        int r = idx;
        const int e = r % 16; r /= 16;
        const int m = r % 32; r /= 32;
        const int kh = r % 2; r /= 2;
        const int mt = r % c.mt; r /= c.mt;
        const int part = r % 2; r /= 2;
        const int tap = r;
        const int k = kh * 16 + e;
        const int mm = mt * 32 + m;
        float v = 0.f;
        if (mm < c.m_count && k < c.k_count) {
            const int co = c.transposed ? c.k_off + k : c.m_off + mm;
            const int ci = c.transposed ? c.m_off + mm : c.k_off + k;
            const int t = c.transposed ? 8 - tap : tap;
            if (!c.virtual4x4) {
                v = src[((size_t)co * c.src_cin + ci) * 9 + t];
            } else {
                const int C = c.src_cin, sub = ci / C, ch = ci % C;
                const int ky = 2 * (t / 3) + (sub >> 1) - 1, kx = 2 * (t % 3) + (sub & 1) - 1;
                if (ky >= 0 && ky < 4 && kx >= 0 && kx < 4) v = src[((size_t)co * C + ch) * 16 + ky * 4 + kx];
            }
            v *= sc;
        }
        const float t = v * kLoScale;
        const half_t w0 = (half_t)t;
        const half_t q = part == 0 ? (half_t)(t - (float)w0) : (half_t)((float)w0 * kLoInv);
        dst[idx] = (unsigned char)f16_bits_to_bf8((unsigned)__builtin_bit_cast(unsigned short, q));
    }
}

int pack_mx_dispatch(const ResrPackChunk* chunks_dev, int n_chunks, const float* arena, void* packed_mx, hipStream_t stream) {
    if (!chunks_dev || !arena || !packed_mx || n_chunks <= 0) return fail(RESR_ERR_ARG, "pack_weights_mx: bad argument");
    hipLaunchKernelGGL(pack_mx_kernel, dim3(n_chunks), dim3(256), 0, stream, chunks_dev, arena, (unsigned char*)packed_mx);
    RESR_CHECK_LAUNCH("pack_mx_kernel");
    return RESR_OK;
}

// two rounded products, one rounded sum: the empty asm makes the products opaque so the compiler
// cannot contract them into an FMA (HIP's default -ffp-contract=fast would)
__device__ __forceinline__ float ema_step(float one_minus, float p, float decay, float s) {
    float a = one_minus * p, b = decay * s;
    asm volatile("" : "+v"(a), "+v"(b));
    return a + b;
}

// EMA.update (reference model.py:43-48): shadow = (1 - decay) * p + decay * shadow, evaluated as two
// rounded products and one rounded sum (no FMA contraction) so it is bit-identical to the reference.
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ shadow, const float* __restrict__ p,
                                                  long count, float one_minus, float decay) {
    const long stride = (long)gridDim.x * 256 * 4;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < count; i += stride) {
        if (i + 4 <= count) {
            const float4v a = *reinterpret_cast<const float4v*>(p + i);
            float4v s = *reinterpret_cast<const float4v*>(shadow + i);
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] = ema_step(one_minus, a[j], decay, s[j]);
            *reinterpret_cast<float4v*>(shadow + i) = s;
        } else {
            for (long j = i; j < count; ++j) shadow[j] = ema_step(one_minus, p[j], decay, shadow[j]);
        }
    }
}

__global__ __launch_bounds__(256) void ema_scalar_kernel(float* __restrict__ shadow, const float* __restrict__ p,
                                                         long count, float one_minus, float decay) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < count) shadow[i] = ema_step(one_minus, p[i], decay, shadow[i]);
}

int ema_dispatch(float* shadow, const float* params, long count, double decay, hipStream_t stream) {
    if (!shadow || !params || count <= 0) return fail(RESR_ERR_ARG, "ema_update: bad argument");
    if ((reinterpret_cast<uintptr_t>(shadow) | reinterpret_cast<uintptr_t>(params)) & 15) {
        // unaligned per-tensor call: peel to the scalar tail path by treating every element as tail
        hipLaunchKernelGGL(ema_scalar_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, shadow,
                           params, count, (float)(1.0 - decay), (float)decay);
        RESR_CHECK_LAUNCH("ema_scalar_kernel");
        return RESR_OK;
    }
    long blocks = (count / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(ema_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, shadow, params, count,
                       (float)(1.0 - decay), (float)decay);
    RESR_CHECK_LAUNCH("ema_kernel");
    return RESR_OK;
}

}  // namespace resr
