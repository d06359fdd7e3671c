// conv3x3_ws_chain.h -- launchers of the chained dense-block passes: the four cout-32 convolutions of a dense block (forward: LeakyReLU, EPI 0, + sign words in
// training, EPI 16; mirrored backward-data: sign-word mask, EPI 33) as one persistent launch of the producer/consumer kernel
// (conv3x3_ws.h, CH) -- fast mode (f16) and, in conv3x3_ws_chain_x2.hip, exact16 (hi/lo pairs).
#pragma once
#include "conv3x3_ws.h"

namespace resr {

template <int NT, int EPI, int CH, int X2>
static int launch_chain(const ConvArgs& a, const ChainArgs& cj, double flop, double bytes, hipStream_t stream) {
    using C = WsCfg<half_t, 1, NT, 8>;
    auto kern = conv3x3_ws_kernel<half_t, 1, NT, 8, EPI, X2, 0, CH>;
    ConvArgs args = a;
    args.tiles_x = (a.w_ + 31) / 32;
    args.tiles_y = (a.h + C::TH - 1) / C::TH;
    const size_t lds = C::LDS_BYTES + C::LUT_BYTES + C::CHAIN_BYTES;
    static int resident_dev[kMaxDevices] = {0};
    static const char* zero_dev[kMaxDevices] = {nullptr};
    int cur_dev = 0;
    if (hipGetDevice(&cur_dev) != hipSuccess || cur_dev < 0 || cur_dev >= kMaxDevices) return fail(RESR_ERR_LAUNCH, "conv3x3_chain: hipGetDevice");
    int& resident = resident_dev[cur_dev];
    const char*& zero = zero_dev[cur_dev];
    if (!resident) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int per_cu = 0;
        hipDeviceProp_t prop;
        void* zp = nullptr;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, C::NTHR, lds) != hipSuccess ||
            hipGetDeviceProperties(&prop, cur_dev) != hipSuccess || per_cu <= 0 ||
            hipGetSymbolAddress(&zp, HIP_SYMBOL(g_conv_zero16)) != hipSuccess || !zp)
            return fail(RESR_ERR_LAUNCH, "conv3x3_chain: occupancy / zero-page query failed");
        zero = (const char*)zp;
        resident = per_cu * prop.multiProcessorCount;
    }
    args.zero = zero;
    args.trace = g_conv_trace;
    const int ntiles = args.tiles_x * args.tiles_y * a.n;   // a multiple of 8 (n is)
    // every workgroup must be resident (the flags are waited for inside the launch): the grid never exceeds what the device
    // holds at once, and is a multiple of 8 so that each XCD gets the same number of workgroups
    unsigned grid = (unsigned)((ntiles < resident ? ntiles : resident) & ~7);
    // $RESR_CHAIN_CUS_PER_XCD = k (1..32, read per call): at most k workgroups per XCD, i.e. 32 - k CUs of every XCD stay free for a
    // co-resident kernel of another stream -- an RCCL collective overlapped with the backward pass (RESR_DP_OVERLAP=1) then runs NEXT
    // to the chained launches instead of delaying their polls.  Costs the launches tiles / (8 k) instead of tiles / 256 rounds:
    // ~3 % at the headline geometry with k = 31, a whole round at one tile per CU (the 64^2 crops).  Results do not depend on it.
    if (const char* e = getenv("RESR_CHAIN_CUS_PER_XCD")) {
        const int k = atoi(e);
        if (k >= 1 && k < 32 && grid > (unsigned)k * 8u) grid = (unsigned)k * 8u;
    }
    if (grid == 0) return fail(RESR_ERR_LAUNCH, "conv3x3_chain: empty grid");
    prof_before(stream);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NTHR), lds, stream, args, cj);
    // 24xxx / 26xxx: a chain of the <f16,1,NT,8> / <f16x2,1,NT,8> kernel; FLOP / bytes = the sum over its jobs
    prof_after(stream, (X2 ? 26000 : 24000) + 100 + NT * 10 + 8, flop, bytes);
    RESR_CHECK_LAUNCH("conv3x3_ws_kernel (chain)");
    return RESR_OK;
}

// kind: 0 = forward at inference (LeakyReLU), 1 = forward in training (LeakyReLU + sign words), 2 = mirrored backward-data;
// jobs of kind "residual half" (the closing convolution inside the chain) take the instantiation with the per-job switches
template <int NT, int CH, int X2>
static int launch_chain_kind(const ConvArgs& a, const ChainArgs& cj, int kind, double flop, double bytes, hipStream_t stream) {
    if constexpr (X2 == 2) {   // MX stages: inference forward chains (LeakyReLU) and the mirrored backward-data chains (sign-word mask)
        if (kind == 1) return fail(RESR_ERR_ARG, "conv3x3_chain: a training forward keeps every pair chunk on three f16 stages (no RESR_CONV_MX_PAIRS)");
        if (kind == 2) return launch_chain<NT, 33, CH, X2>(a, cj, flop, bytes, stream);
        return launch_chain<NT, 0, CH, X2>(a, cj, flop, bytes, stream);
    } else {
        if (kind == 2) return launch_chain<NT, 33, CH, X2>(a, cj, flop, bytes, stream);
        return kind == 1 ? launch_chain<NT, 16, CH, X2>(a, cj, flop, bytes, stream) : launch_chain<NT, 0, CH, X2>(a, cj, flop, bytes, stream);
    }
}

template <int X2>
static int chain_launch_t(const ConvArgs& a, const ChainArgs& cj, int tile_rows, int kind, double flop, double bytes, hipStream_t stream) {
    if constexpr (!X2) {
        if (cj.split[1] > 0) return launch_chain_kind<1, 3, 0>(a, cj, kind, flop, bytes, stream);   // pinned pipeline (experiment): 8-row tiles
    }
    bool mixed = false;
    for (int j = 0; j < cj.njobs; ++j) mixed = mixed || cj.job[j].kind == 3;
    if constexpr (X2) {   // exact16 with the closing convolution's halves: the 16-row shape exceeds its 128-register budget (it spills)
        if (mixed) return launch_chain_kind<1, 2, X2>(a, cj, kind, flop, bytes, stream);
        if (tile_rows >= 16) return launch_chain_kind<2, 1, X2>(a, cj, kind, flop, bytes, stream);
        return launch_chain_kind<1, 1, X2>(a, cj, kind, flop, bytes, stream);
    } else {
        if (tile_rows >= 16)
            return mixed ? launch_chain_kind<2, 2, 0>(a, cj, kind, flop, bytes, stream) : launch_chain_kind<2, 1, 0>(a, cj, kind, flop, bytes, stream);
        return mixed ? launch_chain_kind<1, 2, 0>(a, cj, kind, flop, bytes, stream) : launch_chain_kind<1, 1, 0>(a, cj, kind, flop, bytes, stream);
    }
}

}  // namespace resr
