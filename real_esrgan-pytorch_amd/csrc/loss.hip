// loss.hip -- the scalar losses of the train steps as ONE launch each, forward value and unit gradient together.
//
// Reference call sites: nn.L1Loss (train_realesrnet.py:385; train_realesrgan.py:475 on the USM-sharpened sr) and
// nn.BCEWithLogitsLoss against torch.full(..., 1.0 / 0.0) labels (train_realesrgan.py:460-461,478,500,509).  As stock ATen ops
// a GAN step spent ~56 launches / 1.8 ms on them (label fills, sub / abs / mean, the weight and loss-scale multiplies and each
// one's backward node): on 16 x 256^2 values every one of those launches is latency, not bandwidth.
//
// One kernel per loss: every workgroup walks its share of the values (float4), writes the UNIT gradient d(loss)/d(x) -- weight
// and 1/N folded in -- and one partial sum; the LAST workgroup to arrive (a device-scope counter in the caller's scratch) adds
// the partials in a fixed order, so the value is deterministic and bit-reproducible from run to run, and re-arms the counter.
// HBM-bound: 4 B read (+ 4 B for the second operand of L1) and 4 B written per value.
#include "common.h"

namespace resr {

namespace {

constexpr int kLossThreads = 256;

__device__ __forceinline__ float block_sum(float acc) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    __shared__ float wsum[kLossThreads / 64];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    return (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);   // valid in thread 0
}

// scratch: [0] arrival counter (as unsigned), [1 ...] one partial per workgroup.  Returns true in thread 0 of the last workgroup
// to arrive, after which every partial is visible to it.
__device__ __forceinline__ bool publish_partial(float* scratch, float value) {
    __shared__ bool last;
    if (threadIdx.x == 0) {
        scratch[1 + blockIdx.x] = value;
        __threadfence();
        const unsigned old = atomicAdd(reinterpret_cast<unsigned*>(scratch), 1u);
        last = old == gridDim.x - 1;
    }
    __syncthreads();
    return last;
}

// the last workgroup: partials in a fixed order (thread t takes t, t + 256, ...; then the block tree), counter re-armed
__device__ __forceinline__ void finish(float* scratch, float scale, float* loss) {
    __threadfence();
    float acc = 0.f;
    for (unsigned i = threadIdx.x; i < gridDim.x; i += kLossThreads)
        acc += __hip_atomic_load(scratch + 1 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const float total = block_sum(acc);
    if (threadIdx.x == 0) {
        loss[0] = total * scale;
        *reinterpret_cast<unsigned*>(scratch) = 0u;
    }
}

// F.binary_cross_entropy_with_logits(x, full_like(x, label)) * weight, reduction "mean":
//   l_i = max(x_i, 0) - x_i * label + log1p(exp(-|x_i|));   d l_i / d x_i = sigmoid(x_i) - label
__global__ __launch_bounds__(kLossThreads) void bce_logits_const_kernel(const float* __restrict__ x, long count, float label, float gscale,
                                                                        float lscale, float* __restrict__ loss, float* __restrict__ grad,
                                                                        float* __restrict__ scratch) {
    float acc = 0.f;
    const long n4 = count >> 2;
    for (long i = (long)blockIdx.x * kLossThreads + threadIdx.x; i < n4; i += (long)gridDim.x * kLossThreads) {
        const float4v v = reinterpret_cast<const float4v*>(x)[i];
        float4v g;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t = __expf(-fabsf(v[e]));
            acc += fmaxf(v[e], 0.f) - v[e] * label + log1pf(t);
            const float s = v[e] >= 0.f ? 1.f / (1.f + t) : t / (1.f + t);   // sigmoid, no overflow
            g[e] = (s - label) * gscale;
        }
        if (grad) reinterpret_cast<float4v*>(grad)[i] = g;
    }
    if (blockIdx.x == 0 && threadIdx.x < (count & 3)) {   // tail (counts that are not a multiple of four)
        const long i = (n4 << 2) + threadIdx.x;
        const float v = x[i], t = __expf(-fabsf(v));
        acc += fmaxf(v, 0.f) - v * label + log1pf(t);
        if (grad) grad[i] = ((v >= 0.f ? 1.f / (1.f + t) : t / (1.f + t)) - label) * gscale;
    }
    const float part = block_sum(acc);
    if (publish_partial(scratch, part)) finish(scratch, lscale, loss);
}

// F.l1_loss(a, b) * weight, reduction "mean"; d/d a_i = sign(a_i - b_i) (0 at a tie, like ATen's sgn)
__global__ __launch_bounds__(kLossThreads) void l1_mean_kernel(const float* __restrict__ a, const float* __restrict__ b, long count, float gscale,
                                                               float lscale, float* __restrict__ loss, float* __restrict__ grad,
                                                               float* __restrict__ scratch) {
    float acc = 0.f;
    const long n4 = count >> 2;
    for (long i = (long)blockIdx.x * kLossThreads + threadIdx.x; i < n4; i += (long)gridDim.x * kLossThreads) {
        const float4v va = reinterpret_cast<const float4v*>(a)[i], vb = reinterpret_cast<const float4v*>(b)[i];
        float4v g;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = va[e] - vb[e];
            acc += fabsf(d);
            g[e] = d > 0.f ? gscale : (d < 0.f ? -gscale : 0.f);
        }
        if (grad) reinterpret_cast<float4v*>(grad)[i] = g;
    }
    if (blockIdx.x == 0 && threadIdx.x < (count & 3)) {
        const long i = (n4 << 2) + threadIdx.x;
        const float d = a[i] - b[i];
        acc += fabsf(d);
        if (grad) grad[i] = d > 0.f ? gscale : (d < 0.f ? -gscale : 0.f);
    }
    const float part = block_sum(acc);
    if (publish_partial(scratch, part)) finish(scratch, lscale, loss);
}

// out[r] = coef[r] * sum_k partial[r][k] for r < rows, out[rows] = sum_r out[r]: the five weighted feature distances of the
// perceptual term (model.py:320-335) and their total from the partial sums of resr_l1_partial, one launch.
__global__ __launch_bounds__(256) void weighted_rows_kernel(const float* __restrict__ partial, int rows, int cols,
                                                            float c0, float c1, float c2, float c3, float c4, float c5, float c6, float c7,
                                                            float* __restrict__ out) {
    const float coef[8] = {c0, c1, c2, c3, c4, c5, c6, c7};
    __shared__ float rowv[8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int r = wave; r < rows; r += 4) {
        float acc = 0.f;
        for (int k = lane; k < cols; k += 64) acc += partial[(size_t)r * cols + k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if (lane == 0) rowv[r] = acc * coef[r];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float total = 0.f;
        for (int r = 0; r < rows; ++r) { out[r] = rowv[r]; total += rowv[r]; }
        out[rows] = total;
    }
}

}  // namespace

static int grid_for(long count) {
    const long want = (count / 4 + kLossThreads * 4 - 1) / (kLossThreads * 4);   // ~16 values per thread
    return (int)(want < 1 ? 1 : (want > 1024 ? 1024 : want));
}

int bce_logits_const_dispatch(const float* x, long count, float label, float weight, float* loss, float* grad, float* scratch, hipStream_t st) {
    if (!x || !loss || !scratch || count <= 0) return fail(RESR_ERR_ARG, "bce_logits_const: bad argument");
    if (((size_t)x & 15) || (grad && ((size_t)grad & 15))) return fail(RESR_ERR_ARG, "bce_logits_const: tensors must be 16-byte aligned");
    const float inv = weight / (float)count;
    hipLaunchKernelGGL(bce_logits_const_kernel, dim3(grid_for(count)), dim3(kLossThreads), 0, st, x, count, label, inv, inv, loss, grad, scratch);
    RESR_CHECK_LAUNCH("bce_logits_const_kernel");
    return RESR_OK;
}

int l1_mean_dispatch(const float* a, const float* b, long count, float weight, float* loss, float* grad, float* scratch, hipStream_t st) {
    if (!a || !b || !loss || !scratch || count <= 0) return fail(RESR_ERR_ARG, "l1_mean: bad argument");
    if (((size_t)a & 15) || ((size_t)b & 15) || (grad && ((size_t)grad & 15))) return fail(RESR_ERR_ARG, "l1_mean: tensors must be 16-byte aligned");
    const float inv = weight / (float)count;
    hipLaunchKernelGGL(l1_mean_kernel, dim3(grid_for(count)), dim3(kLossThreads), 0, st, a, b, count, inv, inv, loss, grad, scratch);
    RESR_CHECK_LAUNCH("l1_mean_kernel");
    return RESR_OK;
}

int weighted_rows_dispatch(const float* partial, int rows, int cols, const float* coef, float* out, hipStream_t st) {
    if (!partial || !coef || !out || rows <= 0 || rows > 8 || cols <= 0) return fail(RESR_ERR_ARG, "weighted_rows: bad argument");
    float c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = 0; r < rows; ++r) c[r] = coef[r];
    hipLaunchKernelGGL(weighted_rows_kernel, dim3(1), dim3(256), 0, st, partial, rows, cols, c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], out);
    RESR_CHECK_LAUNCH("weighted_rows_kernel");
    return RESR_OK;
}

}  // namespace resr
