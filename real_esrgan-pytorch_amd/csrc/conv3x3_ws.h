// conv3x3_ws.h -- the fast-mode (f16) 3x3 convolution: persistent workgroups, 4 producer waves + NWC consumer waves.
//
// Same GEMM view and LDS tile layout as conv3x3.hip (reference call sites: model.py:87-98, :123-132, :255-272 and
// their autograd backward-data passes).  What differs is who moves the data:
//
//   producer waves  LDS-DMA (global_load_lds, 16 B per lane) of the next stage's (TH+2) x 34 x 32ch halo tile and of the
//                   chunk's packed weights into free LDS buffers; they wait for the copy and meet the consumers at the
//                   workgroup barrier.
//   consumer waves  never touch global memory before the epilogue: weight fragments (a 3- or 6-unit register ring) and
//                   halo rows both come from LDS and feed the MFMAs.
//
// On gfx950 the vector-memory counter retires in order.  In the one-role kernel a wave's weight loads queued behind
// its own (HBM-latency) halo loads, so every chunk stalled two taps in until its prefetch had landed: memory time
// and MFMA time added up instead of overlapping.  Splitting the roles gives each wave a counter that only tracks one
// kind of traffic.  Workgroups are persistent (grid = one residency wave; tiles b, b+G, ...) so the producers run
// ahead across tile boundaries and the fill/drain phases are paid once per launch, not once per tile.
//
// A stage = (tile, 32-channel chunk).  Barrier protocol, one s_barrier per stage for every wave (two halo buffers):
//   producer:  for s: DMA(s -> buf s&1); wait vmcnt(0); barrier_s
//   consumer:  for s: barrier_s; multiply(buf s&1)
// barrier_{s+1} is passed only when all consumers are done with stage s, so DMA(s+2) may overwrite buf s&1.
// With three halo buffers (16-row tiles of cout <= 32) the halo of stage s+2 is requested right after barrier_s, behind
// the weights of stage s+1, and `s_waitcnt vmcnt(<instructions of that halo>)` on the in-order counter says that
// everything older has landed.
//
// Consumer inner loop: per (k-step, dx) group the NT+2 halo rows are read once from LDS and reused by the three dy
// taps (row t+dy of the tile is row t of tap dy): (NT+2) LDS reads per 3*NT*MT MFMAs instead of 3*NT.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "conv3x3.h"

// Timing experiment only (tools/winograd_bound.sh; never defined in the product build): issue the MFMAs of only the first N of
// the nine taps of every k-step -- everything else (LDS-DMA, LDS fragment reads, barriers, epilogue) unchanged, results wrong.
// N = 4 is the matrix work of a Winograd F(2x2,3x3) formulation (16 products per 2x2 outputs instead of 36) on this kernel's
// memory side: an upper bound of what that formulation could gain, before its transforms and its 16/9 larger weights.
// RESR_TIMING_VALU = V adds V packed-f16 vector instructions per remaining MFMA to the consumer waves: the input transform
// B^T d B of that formulation is 32 adds per 4x4 patch and channel = 8 v_pk_add_f16 per MFMA of a 16-channel k-step.
// RESR_TIMING_NO_W / RESR_TIMING_NO_H (chained launches): the producers skip the weight / halo requests after the first stages
// (tools/build_variant.py ... -DRESR_TIMING_NO_H=1): what a 64^2 launch costs with an idle memory pipe.
// RESR_TIMING_RESIDENT (chained launches, round 6): the memory side of an LDS-RESIDENT dense block -- a workgroup's own tile of the six
// planes stays in LDS across the jobs, only what it cannot have travels: job 0 fetches its two chunks whole (as today), a later job
// requests NOTHING for the planes it already holds and only the one-pixel RING of its newest plane (84 pixels x 64 B = six full LDS-DMA
// instructions instead of 22, behind the usual poll: the ring is the neighbours' data).  Flags, polls, barriers, weights, MFMAs,
// epilogues and stores are today's; the LDS contents are wrong.  An upper bound of what the restructuring could gain (it would
// still owe the epilogue's ds_writes into the next job's operand slots and 130 KB of resident planes next to the weight ring).
#ifndef RESR_TIMING_TAPS
#define RESR_TIMING_TAPS 9
#endif
// Stamps per traced wave of a timeline (trace builds, -DRESR_TRACE=1).  RESR_TRACE=2 ("budget" build, tools/chain_budget.py): 192 stamps per
// wave, so that all 26 stages of a six-job chained launch fit (six producer stamps and three consumer stamps per stage).
#if defined(RESR_TRACE) && RESR_TRACE == 2
#define RESR_TRACE_STAMPS 192
#else
#define RESR_TRACE_STAMPS 64
#endif
#ifndef RESR_TIMING_VALU
#define RESR_TIMING_VALU 0
#endif

namespace resr {

static __device__ uint4 g_conv_zero16 = {0, 0, 0, 0};  // one zero page per translation unit

extern unsigned long long* g_conv_trace;  // debug timeline buffer [32 workgroups][2 roles][64 stamps] (conv3x3_ws.hip)

template <typename T, int MT, int NT, int NWC>
struct WsCfg {
    static constexpr int E = 16 / (int)sizeof(T);
    static constexpr int SPP = 32 / E;              // 16-byte slots per pixel per chunk
    static constexpr int KS = SPP / 2;              // k-steps per chunk
    static constexpr int PB = 32 * (int)sizeof(T);  // bytes per pixel per chunk
    static constexpr int TH = NWC * NT, TW = 32, HH = TH + 2, HW = TW + 2;
    static constexpr int NSLOT = HH * HW * SPP;
    static constexpr int NI = (NSLOT + 63) / 64;    // LDS-DMA instructions per stage (1 KB each)
    static constexpr int BUF = NSLOT * 16;          // one LDS buffer (the last DMA instruction's lanes beyond NSLOT are masked)
    static constexpr int NG = KS * 3;               // (k-step, dx) weight groups per chunk
    static constexpr int WTAP = KS * MT * 1024;     // packed weight bytes per (chunk, tap)
    // 4 consumer + 4 producer waves: a workgroup's waves are dealt to the SIMDs round-robin from SIMD 0, so every SIMD
    // gets one consumer and one producer; two workgroups per CU then need <= 128 VGPRs (4 waves per SIMD).
    // producer waves (a single wave issues ~1 KB of LDS-DMA per 70 ns); 8 where 16 waves fit the register file (<= 128
    // VGPRs): with three buffers more requests can be outstanding, and 8 waves hold them (835 -> 880 TFLOP/s)
    static constexpr int NP = (MT == 1 && NT * NWC <= 16) ? 8 : 4;
    static constexpr int NIP = (NI + NP - 1) / NP;  // LDS-DMA instructions per producer wave per stage
    static constexpr int NTHR = 64 * (NWC + NP);
    // The chunk's packed weights (18 KB per 32 output channels) go through LDS as well: one LDS-DMA copy per stage shared
    // by all consumers instead of one register stream from L2 per consumer wave (for cout 64 those streams were 4/5 of
    // the workgroup's load traffic and set the stage time; for cout 32 they queued behind the halo requests).
    // 16-row tiles of cout <= 32: three 39 KB halo buffers, so the request for stage s+2 is in flight while stage s+1
    // lands and stage s is multiplied (with two buffers the memory pipe drains at every stage of a memory-bound launch:
    // measured 4.2 us of back-pressured issue, then 1.1 us with nothing in flight, per 5.5 us stage of a 32-row tile).
    static constexpr int NHB = (MT == 1 && NT * NWC <= 16) ? 3 : 2;
    static constexpr int WBUF = 9 * KS * MT * 1024;  // packed weight bytes per chunk
    static constexpr int NWI = WBUF / 1024;          // LDS-DMA instructions per chunk of weights
    static constexpr int NWIP = (NWI + NP - 1) / NP;
    static constexpr int WOFF = NHB * BUF + 256;     // after the halo buffers and the bias
    // weight buffers: as many as fit next to the halo buffers (at least the ring of two).  A launch whose chunks all fit
    // keeps them resident (buffer = chunk, loaded during the first tile only); otherwise the two-buffer ring by stage parity.
    static constexpr int NWB = (160 * 1024 - WOFF) / WBUF < 2 ? 2 : (160 * 1024 - WOFF) / WBUF;
    static constexpr int LDS_BYTES = WOFF + NWB * WBUF;
    // LeakyReLU-backward multipliers by 4 mask bits (16 x float4), behind the weight buffers (EPI 33 of the lean epilogue)
    static constexpr int LUT_OFF = LDS_BYTES, LUT_BYTES = 256;
    // chain launches: the biases of kMaxChain jobs + the consumers' arrival counter, behind the table
    // (+ 1 KB for the debug timeline of a traced workgroup: stamps go to LDS and are copied out at the end of the kernel -- a stamp
    // written straight to global memory is a store on the in-order vector-memory counter, and the very waits the timeline is
    // supposed to show then also wait for its acknowledgement)
    static constexpr int CHAIN_OFF = LUT_OFF + LUT_BYTES, CHAIN_TRACE = kMaxChain * 128 + 64, CHAIN_BYTES = CHAIN_TRACE + 2 * RESR_TRACE_STAMPS * 8;
    static_assert(LDS_BYTES + LUT_BYTES + CHAIN_BYTES <= 160 * 1024, "LDS");
    // output-group launches WITH a bias (cout 64 shape; VGG19's 128..512-channel layers): the biases of up to kMaxBiasGroups
    // groups, 64 floats each, where the chained launches (cout 32 shape only) keep their state
    static constexpr int GB_OFF = CHAIN_OFF, GB_BYTES = MT == 2 ? kMaxBiasGroups * 256 : 0;
    static_assert(LDS_BYTES + LUT_BYTES + GB_BYTES <= 160 * 1024, "LDS");
};

__device__ __forceinline__ void conv_glds16(const char* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// LDS-DMA with a uniform 64-bit base (SGPR pair) + 32-bit per-lane byte offset: no per-lane 64-bit address math.
// Written as asm because the builtin only takes a per-lane 64-bit pointer; the producer waits with an explicit
// s_waitcnt vmcnt(0) before the stage barrier (the compiler does not count these loads).
__device__ __forceinline__ void conv_glds16_s(const char* sbase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(sbase)
                 : "memory", "m0");
}

// v_permlane32_swap: lanes 32..63 of `a` trade places with lanes 0..31 of `b`.  (asm: the builtin of this toolchain
// returns its first result twice.)  The nops cover the VALU-write -> permlane-read wait states the compiler would
// otherwise insert itself.
__device__ __forceinline__ void permlane32_swap(float& a, float& b) {
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}


// ---- pieces of the lean epilogue (FAST instantiations of conv3x3_ws_kernel) ----
typedef float float2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef short short2v __attribute__((ext_vector_type(2)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

// v_max_f32 without the canonicalising self-max the compiler puts in front of fmaxf under IEEE mode
__device__ __forceinline__ float vmax_f32(float a, float b) {
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// t * (float)half of a packed f16 pair, one VALU instruction (v_fma_mix_f32 reads the f16 operand directly; + 0 keeps
// the product's single rounding)
__device__ __forceinline__ float mixmul_lo(unsigned h2, float t) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(t));
    return d;
}
__device__ __forceinline__ float mixmul_hi(unsigned h2, float t) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(t));
    return d;
}
// c + t * (float)half of a packed f16 pair (the lo part of an exact16 residual joins the product of its hi part)
__device__ __forceinline__ float mixfma_lo(unsigned h2, float t, float c) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(t), "v"(c));
    return d;
}
__device__ __forceinline__ float mixfma_hi(unsigned h2, float t, float c) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(t), "v"(c));
    return d;
}
// exact16: the lo halves of a converted pair -- f16((v - hi) * 2^12) for both values, given hi2 = the packed f16 pair
// cvt(v0, v1), v4 = (v0, v1) * 2^12 and m = -2^12: one fused multiply-add per value, rounded once (v - hi is exact)
__device__ __forceinline__ unsigned lo_pair(unsigned hi2, float v4_0, float v4_1, float m) {
    unsigned d = 0;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, %2, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "+v"(d) : "v"(hi2), "v"(m), "v"(v4_0), "v"(v4_1));
    return d;
}
// four lane-half exchanges in one block: (a_r, b_r) of lanes lx / lx+32 -> 8 consecutive output channels per lane.
// One pair of wait-state nops for the block (the swaps touch disjoint registers); volatile: stays behind the
// MFMA-result wait states the epilogue issues first.
__device__ __forceinline__ void permlane32_swap4(float& a0, float& a1, float& a2, float& a3, float& b0, float& b1, float& b2, float& b3) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %4\n\tv_permlane32_swap_b32 %1, %5\n\tv_permlane32_swap_b32 %2, %6\n\t"
        "v_permlane32_swap_b32 %3, %7\n\ts_nop 1"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
}
// bit r of the result (times 128) <-> f16 half r of (d0, d1) is > 0, r = 0..3, weights = one byte per half.
// A half is positive iff it is positive as a 16-bit integer; 0 - h (saturating) then carries that in its sign bit.
__device__ __forceinline__ unsigned signs4_x128(unsigned d0, unsigned d1, unsigned weights, unsigned acc) {
    const short2v z = {0, 0};
    const unsigned s0 = __builtin_bit_cast(unsigned, __builtin_elementwise_sub_sat(z, __builtin_bit_cast(short2v, d0)));
    const unsigned s1 = __builtin_bit_cast(unsigned, __builtin_elementwise_sub_sat(z, __builtin_bit_cast(short2v, d1)));
    const unsigned p = __builtin_amdgcn_perm(s1, s0, 0x07050301u);   // the four high bytes
    return __builtin_amdgcn_udot4(p & 0x80808080u, weights, acc, false);
}

// EPI: epilogue features of the instantiation -- bit 0 LeakyReLU-mask multiply, bit 1 residual 0, bit 2 residual 1 (each
// unconditional when set, absent when clear), bit 3 the rest (aux tensors, NCHW fp32 output, clamp; with bit 3 the
// other features are run-time flags), bit 4 also emit the 1-bit sign tensor (RESR_CONV_WRITE_SIGNBITS), bit 5 the mask is
// such a sign tensor (RESR_CONV_MASK_BITS).  The dispatcher instantiates the combinations the networks use.
//
// X2 (RESR_F16X2, "exact16"): every real 32-channel chunk is three ordinary stages -- (x_hi, W0 = f16(w*2^12)),
// (x_hi, W1 = remainder of W0), (x_lo, W2 = f16(w)) with x_lo stored times 2^12 -- accumulated into the same fp32
// registers; the packed weights are laid out in that stage order, so only the producers' chunk -> address map and the
// epilogue (descale, hi/lo split of the result, hi + lo residuals) know about the mode.
//
// SP (sparse taps): the discriminator's 4x4 / stride-2 convolutions (model.py:140-152) run as 3x3 convolutions over the
// 2x2 space-to-depth image with a "virtual" kernel (pack.hip); of the 9 taps of a 32-channel chunk only 2 x 2 are
// non-zero -- which ones depends on the chunk's sub-position (i, j) in the 2x2 cell.  SP = 1 (forward: the INPUT is the
// space-to-depth image, sub-position per input chunk = chunk*32 / a.s2d_c) and SP = 2 (backward-data: the OUTPUT group
// of this launch lies in sub-position a.tap_sub, taps flipped) skip the other 5 taps: 16 instead of 36 tap-products, the
// FLOPs of the dense 4x4 kernel.  Same tiles, buffers and barrier protocol as the dense loop.
//
// CH (chain; 2 = some jobs are residual halves of the closing convolution, switched per job at run time -- its own
// instantiation so that the homogeneous chains of the full-size batch do not pay for the switches): the launch runs cj.njobs dependent convolutions of one dense block back to back (ChainJob, conv3x3.h) --
// job j+1's last input chunk is job j's output plane -- without kernel boundaries: every workgroup walks its tiles of job
// 0, then of job 1, ...; the producers run ahead ACROSS jobs (the prefix chunks of job j+1 do not depend on job j), and
// only the halo of the dependent chunk waits, per tile, for the (up to) nine neighbouring tiles of job j (flags[] in
// device memory, published by the consumers once their stores are acknowledged).  Each XCD owns whole images (tiles
// xcd * T/8 ... of the image-major numbering, n % 8 == 0 host-checked), so writer and reader of a plane always share one
// L2: plain stores and loads are coherent there, and the flags only order them.  Deadlock-free with all workgroups
// resident: a tile of job j never waits for anything of job j+1, consumers publish a tile one stage after its epilogue
// without waiting for their producers, and producers only poll (for stage s+2) after barrier s.
//
// X2 = 2 (MX, RESR_CONV_MX_PAIRS): a pair chunk is TWO stages -- part 0 (x_hi, W0) as above, and part 3: BOTH corrections
// x_hi W1 + x_lo W2 as nine v_mfma_scale_f32_32x32x64_f8f6f4 per output row on 8-bit operands with unit scales.  B = the pixel's q
// record [bf8(x_hi) x 32 | bf8(x_lo) x 32] -- 64 B per pixel and chunk, the f16 record's size, so halo buffers, slot maps and the
// LDS-DMA pipeline are the f16 stage's; a lane (pixel, half h) wants K = 16 h .. 16 h + 15 of either 32-wide block, i.e. the 16-byte
// pieces h and 2 + h of the record: exactly the two k-step reads of an f16 stage.  A = [bf8(W1) | bf8(W2)] per (tap, output row) in
// the same fragment order (resr_pack_weights_mx), one WBUF-sized block per chunk.  bf8 = e5m2 shares f16's exponent range and both
// remainders are stored times 2^12: no scale bytes anywhere.  The corrections carry ~3 good bits of a 2^-11-weighted term.
template <typename T, int MT, int NT, int NWC, int EPI, int X2 = 0, int SP = 0, int CH = 0>
__global__ __launch_bounds__(64 * (NWC + WsCfg<T, MT, NT, NWC>::NP), (NWC + WsCfg<T, MT, NT, NWC>::NP) / 4) void conv3x3_ws_kernel(const ConvArgs a, const std::conditional_t<CH != 0, ChainArgs, ChainNone> cj) {
    static_assert(!CH || (MT == 1 && SP == 0 && (EPI == 0 || EPI == 16 || EPI == 33) && WsCfg<T, MT, NT, NWC>::NHB == 3), "chain: cout-32 dense-block passes only");
    static_assert(!(CH == 3 && X2), "the pinned-pipeline experiment is fast mode only");
    constexpr bool MX = X2 == 2;
    static_assert(!MX || (sizeof(T) == 2 && SP == 0), "MX stages: f16 storage, dense taps");
    using C = WsCfg<T, MT, NT, NWC>;
    constexpr int SPP = C::SPP, KS = C::KS, PB = C::PB, TH = C::TH, TW = C::TW, HW = C::HW, BUF = C::BUF;
    constexpr int NI = C::NI, NG = C::NG, NP = C::NP, NIP = C::NIP;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Output groups: a launch may cover a.ngroups consecutive 64-channel output groups of one convolution (the
    // discriminator's 128..512-channel layers at 32^2..128^2 pixels would otherwise be one under-filled launch per group);
    // tile index = group * spatial tiles + spatial tile, each group with its own packed weights (cout 64 shape only).
    const int ntiles_sp = a.tiles_x * a.tiles_y * a.n;
    const int ntiles = ntiles_sp * (MT == 2 ? a.ngroups : 1);
    const int G = gridDim.x;
    // CH: XCD x = blockIdx & 7 walks tiles [x * T/8, (x+1) * T/8) with its G/8 workgroups (whole images per XCD)
    int first = xcd_remap(blockIdx.x, G);
    int tile_end = ntiles, tile_step = G;
    int pin_job = 0;   // CH 3
    unsigned chain_epoch = 0;   // CH: flag value of "job j of this launch done" = chain_epoch + j + 1
    if constexpr (CH) {
        // Ownership follows the XCD the workgroup REALLY runs on (XCC_ID), not blockIdx: the dispatcher deals workgroups
        // to the XCDs round-robin, but where a dispatch starts is not fixed (measured: launches from a second stream
        // start elsewhere).  Index inside the XCD = a ticket; the last workgroup of a launch to finish resets the tickets
        // (end of the kernel), so every launch counts from 0 as long as it gives each XCD grid / 8 workgroups -- checked here.
        unsigned* tk = reinterpret_cast<unsigned*>(smem + C::CHAIN_OFF + kMaxChain * 128 + 16);
        if (threadIdx.x == 0) {
            const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u;   // HW_REG_XCC_ID[3:0]
            const unsigned per = (unsigned)G >> 3;
            // (requested before the ticket so that the two round trips overlap)
            const unsigned epoch = __hip_atomic_load(cj.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // this launch's epoch (bumped by the previous launch's last workgroup)
            unsigned idx = atomicAdd(cj.state + 8 + (xcc & 7u), 1u);
            if (xcc >= 8u || idx >= per) {   // an XCD with more than its share: counted (the host fails loudly), kept in range
                atomicAdd(cj.state + 3, 1u);
                __hip_atomic_fetch_add(cj.host_errors + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                idx %= per;
            }
            tk[0] = xcc & 7u;
            tk[1] = idx;
            tk[2] = epoch;
        }
        __syncthreads();
        chain_epoch = tk[2];
        const int t8 = ntiles >> 3, xcd = (int)tk[0];
        first = xcd * t8 + (int)tk[1];
        tile_end = (xcd + 1) * t8;
        tile_step = G >> 3;
        if constexpr (CH == 3) {   // pinned pipeline: this workgroup runs ONE job over every pin_nw-th tile of the XCD's range
            const int idx = (int)tk[1];
            while (idx >= cj.split[pin_job + 1]) ++pin_job;
            first = xcd * t8 + (idx - cj.split[pin_job]);
            tile_step = cj.split[pin_job + 1] - cj.split[pin_job];
        }
    }
    // X2 stage map: the first a.pair_chunks real chunks are hi/lo pairs -- three stages (x_hi W0, x_hi W1, x_lo W2) --, the chunks
    // behind them single f16 tensors -- two stages (x W0, x W1), or ONE (x W0: a.single_stages = 1, the growth planes of an inference
    // forward against f16 weights).  Packed weights keep three blocks per real chunk whatever its kind (block = chunk * 3 + part; a
    // single chunk never reads its third, a one-stage chunk only its first).
    // MX: a pair chunk is two stages -- part 0 (x_hi, W0) and part 3 (the q record against the chunk's MX block).
    constexpr int PS = MX ? 2 : 3;   // stages of a pair chunk
    const int x2_p = X2 ? a.pair_chunks : 0, x2_p3 = PS * x2_p;
    const bool x2_one = X2 && a.single_stages == 1;
    auto stages_of = [&](int cin) { const int c = cin >> 5; return !X2 ? c : (c <= x2_p ? PS * c : x2_p3 + (x2_one ? c - x2_p : 2 * (c - x2_p))); };
    auto st_chunk = [&](int ck) { return !X2 ? ck : (ck < x2_p3 ? ck / PS : x2_p + (x2_one ? ck - x2_p3 : ((ck - x2_p3) >> 1))); };      // real chunk of stage ck
    auto st_part = [&](int ck) { return !X2 ? 0 : (ck < x2_p3 ? (MX ? 3 * (ck & 1) : ck % 3) : (x2_one ? 0 : ((ck - x2_p3) & 1))); };   // its part: 0 / 1 on x (hi), 2 on x_lo, 3 on the q record
    auto st_wblock = [&](int ck) { return !X2 ? ck : st_chunk(ck) * 3 + st_part(ck); };                        // its packed weight block (parts 0..2)
    auto last_stages = [&](int cin) { return !X2 ? 1 : ((cin >> 5) <= x2_p ? PS : (x2_one ? 1 : 2)); };        // stages of a job's LAST real chunk
    // does the NEXT stage multiply the halo of stage ck again (part 0 of a chunk that has a part 1)?
    auto st_keeps_halo = [&](int ck) { return X2 && st_part(ck) == 0 && !(x2_one && ck >= x2_p3) && !(MX && ck < x2_p3); };
    int nchunks = stages_of(a.cin);   // stages per tile (CH: of the current job)
    if constexpr (CH) nchunks = stages_of(cj.job[CH == 3 ? pin_job : 0].cin);
    int tk = 0;
    auto stamp = [&](int role) {
#ifndef RESR_TRACE   // The timeline hooks are compiled in only by a trace build (python tools/build_variant.py trace -DRESR_TRACE=1; the
        (void)role;    // timeline tools load it through RESR_LIB_PATH): switched off at run time they still cost the 64^2
        return;        // configurations 2 % (six uniform branches per stage on the producers' serial path -- measured).
#endif
        // traced workgroups: blockIdx 16k .. 16k+0 for k < 32 (a sample across the whole grid)
        if (a.trace && (blockIdx.x & 15) == 0 && blockIdx.x < 512 && lane == 0 && tk < RESR_TRACE_STAMPS && wave <= NWC) {
            if constexpr (CH != 0) reinterpret_cast<unsigned long long*>(smem + C::CHAIN_OFF + C::CHAIN_TRACE)[role * RESR_TRACE_STAMPS + tk++] = __builtin_amdgcn_s_memrealtime();
            else a.trace[((blockIdx.x >> 4) * 2 + role) * RESR_TRACE_STAMPS + tk++] = __builtin_amdgcn_s_memrealtime();
        }
    };
    auto stamp_dump = [&](int role) {   // chained launches: the wave's stamps, LDS -> the trace buffer
        if constexpr (CH != 0) {
            if (a.trace && (blockIdx.x & 15) == 0 && blockIdx.x < 512 && lane == 0 && wave <= NWC)
                for (int i = 0; i < tk; ++i)
                    a.trace[((blockIdx.x >> 4) * 2 + role) * RESR_TRACE_STAMPS + i] = reinterpret_cast<const unsigned long long*>(smem + C::CHAIN_OFF + C::CHAIN_TRACE)[role * RESR_TRACE_STAMPS + i];
        }
    };

    if (wave == 0 || wave == NWC) stamp(wave == 0 ? 1 : 0);  // kernel entry

    // The bias lives in LDS for the whole launch: re-read from global memory per tile it would queue behind the
    // previous tile's stores on the in-order memory counter (measured: ~4 us per tile waiting for store acks).
    float* bias_lds = reinterpret_cast<float*>(smem + C::NHB * BUF);
    if (wave == 0) {
        const bool f_bias = !(a.flags & RESR_CONV_NO_BIAS) && a.bias != nullptr;
        bias_lds[lane] = (f_bias && lane < a.cout) ? a.bias[lane] * (X2 ? kLoScale : 1.f) : 0.f;
    }
    if constexpr (MT == 2 && CH == 0) {   // output groups with a bias: every group's 64 values (the tile loop points bias_cur at its group's)
        // (a grouped launch WITHOUT a bias -- the discriminator's layers, up to 16 groups -- keeps the zeros of bias_lds)
        if (a.ngroups > 1 && !(a.flags & RESR_CONV_NO_BIAS) && a.bias != nullptr) {
            float* gb = reinterpret_cast<float*>(smem + C::GB_OFF);
            for (int i = tid; i < a.ngroups * 64 && i < kMaxBiasGroups * 64; i += C::NTHR)
                gb[i] = a.bias[i] * (X2 ? kLoScale : 1.f);
        }
    }
    if constexpr (CH) {   // the biases of all jobs (kMaxChain x 32 floats) and the consumers' arrival counter
        float* cb = reinterpret_cast<float*>(smem + C::CHAIN_OFF);
        if (wave < kMaxChain && lane < 32) {
            const float* bp = wave < cj.njobs ? cj.job[wave].bias : nullptr;
            cb[wave * 32 + lane] = (bp && !(a.flags & RESR_CONV_NO_BIAS) && lane < a.cout) ? bp[lane] * (X2 ? kLoScale : 1.f) : 0.f;
        }
        if (wave == 0 && lane == 0) *reinterpret_cast<unsigned*>(smem + C::CHAIN_OFF + kMaxChain * 128) = 0u;
    }
    // FAST: instantiations with the lean epilogue (the generator's hot forms: plain / LeakyReLU, residuals, sign-bit
    // output, sign-bit mask); everything else keeps the general one
    constexpr bool FAST = EPI == 0 || EPI == 2 || EPI == 6 || EPI == 16 || EPI == 33;
    if constexpr (FAST && EPI == 33) {
        if (wave == 1 && lane < 16) {
            float4v m;
#pragma unroll
            for (int r = 0; r < 4; ++r) m[r] = (((lane >> r) & 1) ? 1.f : a.slope) * (X2 ? kLoInv : 1.f);   // X2: the accumulators carry 2^12
            *reinterpret_cast<float4v*>(smem + C::LUT_OFF + lane * 16) = m;
        }
    }
    // The set-up barrier (bias / table visible to the consumers) is passed by the producers AFTER they have requested the
    // first stage: the bias load's round trip runs under the first halo's instead of in front of it (a launch of one tile
    // per CU -- the 64^2 training crops -- spends a quarter of its time waiting for that first stage).
    if (wave >= NWC) {
        const int pw = wave - NWC;  // producer index: this wave issues DMA instructions pw, pw + NP, ...
        // =============================== producer ===============================
        if constexpr (C::NHB == 3) {
            // ---- three halo buffers, weights through LDS: H(s+2) is requested right after barrier s, behind W(s+1) ----
            // Every wave issues a static number of LDS-DMA instructions per stage (lanes outside the image read the zero
            // page through a per-lane address select, lanes beyond the tile are masked off inside the asm), so "W(s+1) and
            // H(s+1) have landed" is `s_waitcnt vmcnt(<my instructions of H(s+2)>)` on the in-order counter.
            const int ups3 = (a.flags & RESR_CONV_UPSAMPLE_IN) ? 1 : 0;
            constexpr int REM = NI - (NIP - 1) * NP;   // waves pw < REM own NIP halo instructions, the rest NIP-1
            static_assert(NIP < 64, "vmcnt is 6 bits");
            unsigned cst[NIP], pix[NIP];
            unsigned long long val[NIP], inb[NIP];   // per instruction: lanes that own a slot / whose pixel is inside the image
            const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
            // One instruction per slot group whatever the tile (static count).  The common case -- every lane inside the
            // image -- is the lean scalar-base form (the producers share their SIMDs' issue ports with the MFMA waves:
            // every VALU instruction here queues behind them); edge tiles select the zero page per lane.
            auto dma_s = [&](const char* sbase, unsigned voff, unsigned dst, unsigned long long mask) {
                unsigned long long save;
                asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %3, %4\n\ts_mov_b64 exec, %0"
                             : "=&s"(save) : "s"(mask), "s"(dst), "v"(voff), "s"(sbase) : "memory", "m0");
            };
            auto dma_v = [&](const char* src, unsigned dst, unsigned long long mask) {
                unsigned long long save;
                asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %3, off\n\ts_mov_b64 exec, %0"
                             : "=&s"(save) : "s"(mask), "s"(dst), "v"(src) : "memory", "m0");
            };
#pragma unroll
            for (int i = 0; i < NIP; ++i) {
                const unsigned s = (i * NP + pw) * 64 + lane;
                const unsigned hp = s / SPP, cp = s % SPP;
                const unsigned hy = (hp * 61681u) >> 21;  // hp / 34, exact below 100000
                const unsigned hx = hp - hy * HW;
                cst[i] = (i * NP + pw < NI && s < (unsigned)C::NSLOT) ? (hy << 8 | hx | ((cp ^ swz<SPP>((int)hx)) << 20)) : ~0u;
                val[i] = __ballot(cst[i] != ~0u);
            }
            // Per-lane byte offsets are computed once per tile (and pixel stride), not per stage: the producers share their
            // SIMDs' VALU port with the MFMA waves, and a stage's requests should be scalar + vector-memory instructions only.
            unsigned voff[NIP];
            unsigned voff_stride = 0;   // pixel stride the offsets were computed for (0: stale)
            int pix_tile = -1;   // the tile pix[] / inb[] were worked out for: a chained launch of ONE tile per workgroup comes back to the same tile for
                                 // every job, and the 39 per-lane pixel indices are the producers' own serial work (tools/chain_budget.py: "advance")
            auto tile_pix = [&](int tile) {
                if (tile == pix_tile) return;
                pix_tile = tile;
                const int tx = tile % a.tiles_x;
                const int t2 = tile / a.tiles_x;
                const int ty = t2 % a.tiles_y;
                const int n = t2 / a.tiles_y;
                const int x0 = tx * TW - 1, y0 = ty * TH - 1;
                const unsigned nbase = (unsigned)n * a.hs * a.ws;
#pragma unroll
                for (int i = 0; i < NIP; ++i) {
                    const unsigned c = cst[i];
                    const int iy = y0 + (int)((c >> 8) & 0xff), ix = x0 + (int)(c & 0xff);
                    const bool ok = c != ~0u && (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w_;
                    pix[i] = ok ? nbase + (unsigned)(iy >> ups3) * a.ws + (unsigned)(ix >> ups3) : ~0u;
                    inb[i] = __ballot(ok);
                }
                voff_stride = 0;
            };
            const char* const zero = a.zero;
            const char* poison_src = nullptr;   // CH: set by a poll that gave up, consumed by the halo request that follows it
            auto issue_h = [&](int ck, int hb) {   // halo of (current pix, chunk ck) -> halo buffer hb
                const int ckr = st_chunk(ck);   // X2: stage ck = (real chunk, part); part 2 reads the lo tensor
                const int c0 = ckr * 32;
                const bool seg1 = c0 >= a.cin0;
                const char* base = seg1 ? a.in1 + (size_t)((c0 - a.cin0) >> 5) * a.in1_chunk_b : a.in0 + (size_t)(c0 >> 5) * a.in0_chunk_b;
                if (X2 && st_part(ck) == 2) base += seg1 ? a.in1_lo_b : a.in0_lo_b;
                if (MX && st_part(ck) == 3) base += seg1 ? a.in1_q_b : a.in0_q_b;
                const unsigned stride_b = seg1 ? a.in1_stride_b : a.in0_stride_b;
                if (stride_b != voff_stride) {
#pragma unroll
                    for (int i = 0; i < NIP; ++i) voff[i] = __umul24(pix[i], stride_b) + ((cst[i] >> 16) & 0xfff0u);   // tensor < 4 GB (host-checked)
                    voff_stride = stride_b;
                }
                if constexpr (CH != 0) {
                    if (poison_src) {   // the poll for this plane timed out: NaNs instead of a plane that may be stale
#pragma unroll
                        for (int i = 0; i < NIP; ++i)
                            if (i * NP + pw < NI) dma_v(poison_src, lds_base + hb * BUF + (i * NP + pw) * 1024, val[i]);
                        return;
                    }
                }
#pragma unroll
                for (int i = 0; i < NIP; ++i) {
                    if (i * NP + pw < NI) {   // wave-uniform; false only for i = NIP-1 of the waves pw >= REM
                        const unsigned dst = lds_base + hb * BUF + (i * NP + pw) * 1024;
                        if (inb[i] == val[i]) {
                            dma_s(base, voff[i], dst, val[i]);
                        } else {
                            // (Tried, late in round 6: the inside lanes of a group that straddles the image border in the scalar-base form and
                            // 16 zero bytes stored by the outside lanes themselves, the zero page only for groups that lie outside entirely -- at
                            // 64^2 every 16 x 32 tile touches the border.  Correct, and no faster: config 3 1770 -> 1794 images/s together with the
                            // cached pixel indices above, which alone give 1788.  A timing build that simply masked the outside lanes off ran 24 %
                            // faster -- on NaNs: stale LDS at the borders, a NaN loss, and a chip whose clocks rise when its operands stop toggling.)
                            const char* src = pix[i] != ~0u ? base + voff[i] : zero;
                            dma_v(src, dst, val[i]);
                        }
                    }
                }
            };
            auto issue_w = [&](int ck, int par) {   // stage ck's packed weights, lane-linear = fragment order
                const char* wbase = (MX && st_part(ck) == 3) ? a.w_mx + (size_t)st_chunk(ck) * C::WBUF : a.w + (size_t)st_wblock(ck) * C::WBUF;
#pragma unroll
                for (int i = 0; i < C::NWIP; ++i) {
                    const int idx = i * NP + pw;
                    if (idx < C::NWI) conv_glds16_s(wbase, (unsigned)(idx * 1024) + ((unsigned)lane << 4), lds_base + C::WOFF + par * C::WBUF + idx * 1024);
                }
            };
            auto wait_all_but_h = [&]() {   // everything but the halo stage issued last has landed
                constexpr int K1 = NIP, K0 = NIP - 1;
                if (pw < REM) __builtin_amdgcn_s_waitcnt((K1 & 15) | 0x0F70 | ((K1 >> 4) << 14));
                else __builtin_amdgcn_s_waitcnt((K0 & 15) | 0x0F70 | ((K0 >> 4) << 14));
            };
            if constexpr (CH) {
                // ---- chain: the same pipeline over the stages of ALL jobs; ring weights; the dependent chunk polls first ----
                const int njobs = cj.njobs;
                int job = CH == 3 ? pin_job : 0, nch = nchunks;   // job / chunk count of the stage whose halo was requested last
                int nlast = last_stages(cj.job[job].cin);         // ... and the stages of that job's last (dependent) real chunk
                int it = first, ick = 0, hb = 0;
                bool tile_settled = false;   // the current tile's neighbourhood has shown the progress its last chunk needs
                // job jb's weights of chunk ck: fragment idx of this kernel's 32-channel output tile is fragment idx * w_mt + w_m of
                // a buffer packed for w_mt tiles (the closing convolution's halves read the cout-64 packing in place)
                // (Tried: the job's fields cached in registers per job instead of re-read from the argument segment per stage, and the
                // launch's input arguments pinned in SGPRs -- 1.2 % slower on the 64^2 configurations; the scalar loads are not the cost.)
                auto issue_wj = [&](int jb, int ck, int par) {
                    int wmt = 1, wm = 0;
                    if constexpr (CH == 2) { wmt = cj.job[jb].w_mt; wm = cj.job[jb].w_m; }
                    const char* wbase = (MX && st_part(ck) == 3) ? cj.job[jb].w_mx + (size_t)st_chunk(ck) * C::WBUF * wmt + (size_t)wm * 1024
                                                                 : cj.job[jb].w + (size_t)st_wblock(ck) * C::WBUF * wmt + (size_t)wm * 1024;
#pragma unroll
                    for (int i = 0; i < C::NWIP; ++i) {
                        const int idx = i * NP + pw;
                        if (idx < C::NWI) conv_glds16_s(wbase, (unsigned)(idx * wmt * 1024) + ((unsigned)lane << 4), lds_base + C::WOFF + par * C::WBUF + idx * 1024);
                    }
                };
                // lanes 0..8 watch the tile's 3 x 3 neighbourhood inside its image; `need` = flag value of "previous job done"
                // Waits until every neighbour's progress is >= need; returns whether it is already >= all_need (then the tile
                // needs no further polls in this job).  (Polling with scalar loads -- their own counter instead of the in-order
                // vector-memory queue behind the LDS-DMA requests -- was tried: eight waves per workgroup re-reading the same L2
                // lines at scalar-load rate saturate the channel the flags live in, 2.4x slower; one polling wave + an LDS
                // sequence number for the others, still 5 % slower than this.)
                auto poll = [&](int tile, unsigned need, unsigned all_need) -> bool {
                    const int tx = tile % a.tiles_x, ty = (tile / a.tiles_x) % a.tiles_y;
                    const int dx = lane % 3 - 1, dy = lane / 3 - 1;
                    const bool watch = lane < 9 && (unsigned)(tx + dx) < (unsigned)a.tiles_x && (unsigned)(ty + dy) < (unsigned)a.tiles_y;
                    const unsigned* fp = cj.state + kChainHdr + (watch ? tile + dy * a.tiles_x + dx : tile);
                    bool all_ok = false;
                    for (int spin = 0;; ++spin) {
                        const unsigned v = watch ? __hip_atomic_load(fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : all_need;
                        if (__ballot((int)(v - need) < 0) == 0ull) {
                            all_ok = __ballot((int)(v - all_need) < 0) == 0ull;
                            break;
                        }
                        // never hang the device: give up after ~4 s (a co-resident kernel of another stream -- a collective waiting
                        // for a late peer -- may legitimately hold a CU this launch needs for a while), and once any poll on this
                        // state has given up every later one does so within a millisecond.  Giving up is never silent: the plane
                        // is replaced by NaNs (poison_src) and the counters -- device and host-mapped -- say so.
                        if (spin > (1 << 21) || ((spin & 1023) == 1023 && __hip_atomic_load(cj.state + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                            if (lane == 0) {
                                atomicAdd(cj.state + 2, 1u);
                                __hip_atomic_fetch_add(cj.host_errors, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            }
                            poison_src = cj.nan16;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    return all_ok;
                };
                auto advance = [&]() -> bool {
                    if (++ick == nch) {
                        ick = 0;
                        it += tile_step;
                        if (it >= tile_end) {
                            it = first;
                            if constexpr (CH == 3) job = njobs;   // pinned: this workgroup's only job is done
                            else if (++job < njobs) { nch = stages_of(cj.job[job].cin); nlast = last_stages(cj.job[job].cin); }
                        }
                    }
                    return job < njobs;
                };
                tile_pix(it);
                issue_wj(job, 0, 0);
                issue_h(0, 0);
                __syncthreads();           // the set-up barrier
                // X2: stage (chunk, part 1) multiplies the x_hi halo of part 0 again with the second weight block: no halo of its own
                auto needs_hc = [&](int ck) { return !X2 || st_part(ck) != 1; };
                bool have_next = advance();   // chunk 1 of the same tile (every job has >= 2 chunks, host-checked)
                int ck_next = ick, job_next = job;
                if (have_next && needs_hc(ick)) {
                    if (ick == 0) tile_pix(it);
                    hb = 1;
                    issue_h(ick, hb);
                    wait_all_but_h();
                } else {
                    __builtin_amdgcn_s_waitcnt(0x0F70);
                }
                __syncthreads();           // barrier 0
                for (int s = 0; have_next; ++s) {   // stage s is being multiplied; H(s+1) is in flight
                    stamp(0);
#ifndef RESR_TIMING_NO_W   // (timing experiments, never in the product build: the ring keeps the first stages' weights / halos -- results wrong)
                    issue_wj(job_next, ck_next, (s + 1) & 1);
#endif
                    stamp(0);
                    const bool have_next2 = advance();
                    if (have_next2 && needs_hc(ick)) {
                        if (ick == 0) tile_pix(it);
                        // Chunk c >= 2 of the shared inputs is the output plane of job c - 2 of THIS launch: its halo may only be
                        // requested once that job has finished the tile's 3 x 3 neighbourhood.  A workgroup that WALKS the jobs
                        // only has to poll for its job's last chunk (the output of job `dep`): it requested the earlier planes
                        // of this very tile's neighbourhood last job, behind that job's poll, and progress never decreases.  A
                        // PINNED workgroup (CH 3) never ran the previous job on this tile: it also waits, before chunk 2, for
                        // job - 2 (which covers chunks 2 .. nch - 2).
                        int dep = job - 1;   // homogeneous chains and pinned pipelines: the previous job
                        if constexpr (CH == 2) dep = cj.job[job].dep;
                        if constexpr (CH == 3) {
                            if (ick == 0) tile_settled = false;
                            if (ick == 2 && dep >= 1) tile_settled = poll(it, chain_epoch + (unsigned)dep, chain_epoch + (unsigned)dep + 1u);
                        }
                        // (X2: the dependent real chunk is the job's last three -- a single f16 plane: two -- stages; its first halo request polls)
                        stamp(0);
                        if (ick == nch - nlast && dep >= 0 && !(CH == 3 && tile_settled)) poll(it, chain_epoch + (unsigned)dep + 1u, chain_epoch + (unsigned)dep + 1u);
                        stamp(0);
                        hb = hb == 2 ? 0 : hb + 1;
#if defined(RESR_TIMING_RESIDENT)
                        {
                            const bool newest = ick >= nch - nlast;     // the job's dependent chunk: its ring is the neighbours' data
                            if (job == 0) { issue_h(ick, hb); poison_src = nullptr; stamp(0); wait_all_but_h(); }
                            else if (newest) {
                                // the ring: six full instructions (waves 0..5 one each), same source plane
                                if (pw < 6 && NIP > 0) {
                                    const int ckr = st_chunk(ick);
                                    const int c0 = ckr * 32;
                                    const bool seg1 = c0 >= a.cin0;
                                    const char* base = seg1 ? a.in1 + (size_t)((c0 - a.cin0) >> 5) * a.in1_chunk_b : a.in0 + (size_t)(c0 >> 5) * a.in0_chunk_b;
                                    const unsigned stride_b = seg1 ? a.in1_stride_b : a.in0_stride_b;
                                    if (stride_b != voff_stride) {
#pragma unroll
                                        for (int i = 0; i < NIP; ++i) voff[i] = __umul24(pix[i], stride_b) + ((cst[i] >> 16) & 0xfff0u);
                                        voff_stride = stride_b;
                                    }
                                    const unsigned dst = lds_base + hb * BUF + pw * 1024;
                                    if (inb[0] == val[0]) dma_s(base, voff[0], dst, val[0]);
                                    else dma_v(pix[0] != ~0u ? base + voff[0] : zero, dst, val[0]);
                                }
                                poison_src = nullptr; stamp(0);
                                if (pw < 6) __builtin_amdgcn_s_waitcnt((1 & 15) | 0x0F70); else __builtin_amdgcn_s_waitcnt(0x0F70);
                            } else { poison_src = nullptr; stamp(0); __builtin_amdgcn_s_waitcnt(0x0F70); }
                        }
#else
#ifndef RESR_TIMING_NO_H
                        issue_h(ick, hb);
#endif
                        poison_src = nullptr;
                        stamp(0);
#ifndef RESR_TIMING_NO_H
                        wait_all_but_h();
#else
                        __builtin_amdgcn_s_waitcnt(0x0F70);
#endif
#endif
                    } else {
                        __builtin_amdgcn_s_waitcnt(0x0F70);
                    }
                    stamp(0);
                    __syncthreads();       // barrier s+1
                    ck_next = ick; job_next = job;
                    have_next = have_next2;
                }
                stamp_dump(0);
                return;
            }
            if (first >= ntiles) { __syncthreads(); return; }
            // X2: stage (chunk, part 1) multiplies the SAME x_hi halo as (chunk, part 0) with the second weight block: no new
            // halo is requested for it and the consumers stay on the buffer (one LDS-DMA halo less per three stages)
            auto needs_h = [&](int ck) { return !X2 || st_part(ck) != 1; };
            const bool wres = nchunks <= C::NWB;
            int it = first, ick = 0;   // the stage whose halo was requested last
            int hb = 0;                // ... and its buffer
            stamp(0);
            tile_pix(it);
            stamp(0);
            issue_w(0, 0);
            stamp(0);
            issue_h(0, 0);
            stamp(0);
            __syncthreads();           // the set-up barrier
            int ck_next = 0;           // chunk of stage s+1 (valid when have_next)
            bool have_next;            // stage s+1 exists (its halo is in flight)
            if (++ick == nchunks) { ick = 0; it += G; }
            have_next = it < ntiles;
            if (have_next) {
                if (ick == 0) tile_pix(it);
                ck_next = ick;
                if (needs_h(ick)) {
                    hb = 1;
                    issue_h(ick, hb);
                    wait_all_but_h();
                } else {
                    __builtin_amdgcn_s_waitcnt(0x0F70);
                }
            } else {
                __builtin_amdgcn_s_waitcnt(0x0F70);
            }
            stamp(0);
            __syncthreads();           // barrier 0
            for (int s = 0; have_next; ++s) {   // stage s is being multiplied; H(s+1) is in flight
                stamp(0);
                // ring: its buffer was read by stage s-1, free past barrier s; resident: loaded during the first tile only
                if (!wres) issue_w(ck_next, (s + 1) & 1);
                else if (s + 1 < nchunks) issue_w(ck_next, ck_next);
                if (++ick == nchunks) { ick = 0; it += G; }
                const bool have_next2 = it < ntiles;
                if (have_next2 && needs_h(ick)) {
                    if (ick == 0) tile_pix(it);
                    hb = hb == 2 ? 0 : hb + 1;
                    issue_h(ick, hb);                    // buffer of stage s-1 as well
                    stamp(0);
                    wait_all_but_h();
                } else {
                    stamp(0);
                    __builtin_amdgcn_s_waitcnt(0x0F70);
                }
                stamp(0);
                __syncthreads();       // barrier s+1
                ck_next = ick;
                have_next = have_next2;
            }
            return;
        }
        const int ups = (a.flags & RESR_CONV_UPSAMPLE_IN) ? 1 : 0;
        // LDS slot s = i*64 + lane (lane-linear destination) holds piece (s % SPP) ^ swz(hx) of halo pixel s / SPP: the
        // XOR swizzle of the consumers' conflict-free reads is applied on the source side.  Tile-independent part,
        // packed hy<<8 | hx | (piece*16)<<16; ~0u = slot beyond the tile (never read).
        // Each slot of the very first stage is requested as soon as its index math is done (the fill phase is paid by
        // every launch: the memory latency runs under the rest of the math instead of after it).
        unsigned cst[NIP], pix[NIP], pixn[NIP];
        {
            const int fsp = first % ntiles_sp;
            const int tx = fsp % a.tiles_x;
            const int t2 = fsp / a.tiles_x;
            const int ty = t2 % a.tiles_y;
            const int n = t2 / a.tiles_y;
            const int x0 = tx * TW - 1, y0 = ty * TH - 1;
            const unsigned nbase = (unsigned)n * a.hs * a.ws;
            const unsigned dst0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
#pragma unroll
            for (int i = 0; i < NIP; ++i) {
                const unsigned s = (i * NP + pw) * 64 + lane;
                const unsigned hp = s / SPP, cp = s % SPP;
                const unsigned hy = (hp * 61681u) >> 21;  // hp / 34, exact below 100000
                const unsigned hx = hp - hy * HW;
                const unsigned c = (i * NP + pw < NI && s < (unsigned)C::NSLOT) ? (hy << 8 | hx | ((cp ^ swz<SPP>((int)hx)) << 20)) : ~0u;
                cst[i] = c;
                const int iy = y0 + (int)hy, ix = x0 + (int)hx;
                const bool ok = c != ~0u && first < ntiles && (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w_;
                pix[i] = ok ? nbase + (unsigned)(iy >> ups) * a.ws + (unsigned)(ix >> ups) : ~0u;
                if (first < ntiles) {
                    if (pix[i] != ~0u) conv_glds16_s(a.in0, __umul24(pix[i], (unsigned)a.in0_stride_b) + (c >> 16), dst0 + (i * NP + pw) * 1024);
                    else if (c != ~0u) conv_glds16_s(a.zero, 0u, dst0 + (i * NP + pw) * 1024);
                }
            }
        }
        const unsigned wdst0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(smem + C::WOFF);
        auto stage_weights = [&](int tile, int ck, int par) {   // stage ck's packed weights of the tile's output group, lane-linear = fragment order
            const char* wbase = (MX && st_part(ck) == 3) ? a.w_mx + (size_t)st_chunk(ck) * C::WBUF   // (MX launches have one output group: host-checked)
                                                         : a.w + (size_t)(tile / ntiles_sp) * a.w_group_b + (size_t)st_wblock(ck) * C::WBUF;
#pragma unroll
            for (int i = 0; i < C::NWIP; ++i) {
                const int idx = i * NP + pw;
                if (idx < C::NWI) conv_glds16_s(wbase, (unsigned)(idx * 1024) + ((unsigned)lane << 4), wdst0 + par * C::WBUF + idx * 1024);
            }
        };
        if (first < ntiles) stage_weights(first, 0, 0);
        __syncthreads();   // the set-up barrier
        // source pixel index per slot (< 2^24, host-checked); ~0u = zero (padding / outside the image)
        auto tile_pix = [&](int tile, unsigned (&pix)[NIP]) {
            const int tsp = tile % ntiles_sp;
            const int tx = tsp % a.tiles_x;
            const int t2 = tsp / a.tiles_x;
            const int ty = t2 % a.tiles_y;
            const int n = t2 / a.tiles_y;
            const int x0 = tx * TW - 1, y0 = ty * TH - 1;
            const unsigned nbase = (unsigned)n * a.hs * a.ws;
#pragma unroll
            for (int i = 0; i < NIP; ++i) {
                const unsigned c = cst[i];
                const int iy = y0 + (int)((c >> 8) & 0xff), ix = x0 + (int)(c & 0xff);
                const bool ok = c != ~0u && (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w_;
                pix[i] = ok ? nbase + (unsigned)(iy >> ups) * a.ws + (unsigned)(ix >> ups) : ~0u;
            }
        };
        int par = 0, hpar = 0;   // weight-ring / halo-buffer parity (they differ in X2, where a halo serves two stages)
        int nstage = 0;   // index of the stage this iteration requests
        const bool wres2 = nchunks <= C::NWB && ntiles == ntiles_sp;   // resident weights need one output group per launch
        for (int tile = first; tile < ntiles; tile += G) {
            stamp(0);
            for (int ck = 0; ck < nchunks; ++ck) {
                const int ckr = st_chunk(ck), cpart = st_part(ck);
                const int c0 = ckr * 32;
                const bool seg1 = c0 >= a.cin0;
                const char* base = seg1 ? a.in1 + (size_t)((c0 - a.cin0) >> 5) * a.in1_chunk_b : a.in0 + (size_t)(c0 >> 5) * a.in0_chunk_b;
                if (X2 && cpart == 2) base += seg1 ? a.in1_lo_b : a.in0_lo_b;
                if (MX && cpart == 3) base += seg1 ? a.in1_q_b : a.in0_q_b;
                const unsigned stride_b = seg1 ? a.in1_stride_b : a.in0_stride_b;
                const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(smem + hpar * BUF);
                const bool new_halo = !X2 || cpart != 1;   // X2 part 1: the x_hi halo of part 0 again, second weight block
                if (tile != first || ck != 0) {  // the first stage was requested above
#pragma unroll
                    for (int i = 0; i < NIP; ++i) {
                        if (!new_halo) break;
                        // uniform base + 32-bit lane offset (tensor < 4 GB, host-checked); padding lanes copy the zero page
                        const unsigned ldst = dst + (i * NP + pw) * 1024;
                        if (pix[i] != ~0u) {
                            conv_glds16_s(base, __umul24(pix[i], stride_b) + (cst[i] >> 16), ldst);
                        } else if (cst[i] != ~0u) {
                            conv_glds16_s(a.zero, 0u, ldst);
                        }
                    }
                    if (!wres2) stage_weights(tile, ck, par);                    // ring of two by stage parity
                    else if (nstage < nchunks) stage_weights(tile, ck, ck);    // resident: buffer = chunk, first tile only
                }
                par ^= 1;
                if (!st_keeps_halo(ck)) hpar ^= 1;   // X2 part 0 of a multi-stage chunk: the next stage stays on this buffer
                ++nstage;
                stamp(0);
                // the next tile's index math runs while this tile's last chunk is in flight
                if (ck == nchunks - 1 && tile + G < ntiles) tile_pix(tile + G, pixn);
                __builtin_amdgcn_s_waitcnt(0x0F70);
                stamp(0);
                __syncthreads();  // (vmcnt already 0: the DMA is asm) the stage barrier
                stamp(0);
            }
#pragma unroll
            for (int i = 0; i < NIP; ++i) pix[i] = pixn[i];
        }
        return;
    }

    // =============================== consumers ===============================
    __syncthreads();   // the set-up barrier
    const int lx = lane & 31, kh = lane >> 5;
    const int row0 = wave * NT;
    const unsigned lane16 = (unsigned)lane << 4;
    // per-lane LDS byte offsets of the B fragment for (dx, k-step); rows add compile-time immediates
    int boff[3][KS];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            boff[dx][ks] = row0 * (HW * PB) + (lx + dx) * PB + (((ks * 2 + kh) ^ swz<SPP>(lx + dx)) << 4);

    // Accumulators start from the bias: register g*4+r of tile m holds cout m*32 + g*8 + kh*4 + r.
    float16v acc[MT][NT];
    const float* bias_cur = bias_lds;   // CH: the current job's 32 values
    auto init_acc = [&]() {
        int kh_l = lane >> 5;
        asm volatile("" : "+v"(kh_l));  // opaque: keeps the 16*MT bias values out of registers across the tile loop
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4v b = *reinterpret_cast<const float4v*>(bias_cur + m * 32 + g * 8 + kh_l * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[m][t][g * 4 + r] = b[r];
            }
    };
    if constexpr (CH) bias_cur = reinterpret_cast<const float*>(smem + C::CHAIN_OFF) + (CH == 3 ? pin_job * 32 : 0);
    // output groups: the accumulators of a tile start from ITS group's bias
    auto group_bias = [&](int tile) {
        if constexpr (MT == 2 && CH == 0) {
            if (a.ngroups > 1 && !(a.flags & RESR_CONV_NO_BIAS) && a.bias != nullptr && tile < ntiles)   // (host-checked: <= kMaxBiasGroups groups)
                bias_cur = reinterpret_cast<const float*>(smem + C::GB_OFF) + (tile / ntiles_sp) * 64;
        }
    };
    group_bias(first);
    init_acc();

    // Weight fragments: ring of (k-step, dx, dy) units -- one tap's MT fragments each -- in consumption order, fetched
    // RING-1 units ahead.  NU is a multiple of RING, so the slot of a unit is static across chunks and tiles.
    constexpr int NU = 9 * KS;
    constexpr int RING = MT == 1 ? 6 : 3;
    static_assert(NU % RING == 0, "ring slots must be static");
    uint4 wr[RING][MT];
    const char* wlds = smem + C::WOFF + lane16;   // this lane's piece of every fragment of weight buffer 0
    auto wload = [&](int slot, int par, int u) {
        const int ks = u / 9, dx = (u / 3) % 3, dy = u % 3;
#pragma unroll
        for (int m = 0; m < MT; ++m)
            wr[slot][m] = *reinterpret_cast<const uint4*>(wlds + par * C::WBUF + (((dy * 3 + dx) * KS + ks) * MT + m) * 1024);
    };

    unsigned timing_valu[4] = {0x3c003c00u, 0x3c003c00u, 0x38003800u, 0x34003400u};   // RESR_TIMING_VALU only
    (void)timing_valu;
    int par = 0, hbc = 0;   // weight-buffer parity / halo buffer of the current stage
    // CH: a finished tile is published (flags[tile] = pend_tag) at the end of the NEXT stage's multiply: by then its stores
    // have long been acknowledged, and the consumers never wait for their producers between epilogue and publication
    bool pend = false;
    int pend_tile = 0;
    unsigned pend_tag = 0;
    const int njobs_c = [&] { if constexpr (CH) return cj.njobs; else return 1; }();
    auto publish = [&]() {
        if constexpr (CH) {
            if (pend) {
                // this wave's stores of the finished tile are acknowledged (in L2).  As asm with a memory clobber: the builtin is
                // "no memory" to the compiler, which may then move the stores below it or the arrival above it
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) {
                    unsigned* cnt = reinterpret_cast<unsigned*>(smem + C::CHAIN_OFF + kMaxChain * 128);
                    const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if ((old % NWC) == NWC - 1)   // the last of the NWC consumer waves: everyone's stores are in
                        __hip_atomic_store(cj.state + kChainHdr + pend_tile, pend_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                pend = false;
            }
        }
    };
    // Lean epilogue: the tile's coordinates (column tile, row tile, image; output group) are carried from tile to tile --
    // derived by division they are ~65 dependent scalar instructions per tile (twice with the mask prefetch), paid
    // between two tiles where nothing overlaps them.  One division chain per job (the first tile), additions with carry after.
    int c_tx = 0, c_ty = 0, c_n = 0, c_grp = 0;
    const int s_tx = tile_step % a.tiles_x, s_t2 = tile_step / a.tiles_x;
    const int s_ty = s_t2 % a.tiles_y, s_n = s_t2 / a.tiles_y;   // s_n counts images of ALL groups (group = image / a.n)
    // CH 3: only the pinned job
    for (int job = (CH == 3 ? pin_job : 0); job < (CH == 3 ? pin_job + 1 : njobs_c); ++job) {
    if constexpr (CH) nchunks = stages_of(cj.job[job].cin);
    for (int tile = first; tile < tile_end; tile += tile_step) {
        if constexpr (FAST) {
            if (tile == first) {
                const int t2 = first / a.tiles_x;
                c_tx = first % a.tiles_x;
                c_ty = t2 % a.tiles_y;
                c_n = t2 / a.tiles_y;
            } else {
                c_tx += s_tx;
                if (c_tx >= a.tiles_x) { c_tx -= a.tiles_x; ++c_ty; }
                c_ty += s_ty;
                if (c_ty >= a.tiles_y) { c_ty -= a.tiles_y; ++c_n; }
                c_n += s_n;
            }
            // output groups (cout 64 shape): tile index = group * spatial tiles + spatial tile, image-major inside a group
            c_grp = 0;
            if (MT == 2 && a.ngroups > 1) c_grp = c_n / a.n;
        }
        // lean epilogue, sign-bit mask (EPI 33): the mask words of this wave's rows are requested HERE and land under the
        // tile's MFMAs (requested in the epilogue they cost one exposed memory round trip per row)
        unsigned mword[EPI == 33 ? NT : 1][MT];
        bool own_top = true;   // CH 2: not for the closing convolution's halves (kind 3: no mask)
        if constexpr (CH == 2) own_top = cj.job[job].kind != 3;
        if constexpr (EPI == 33) if (own_top) {
            int x, y0, n;
            if constexpr (FAST) {
                x = c_tx * TW + (lane & 31);
                y0 = c_ty * TH + row0;
                n = c_n - c_grp * a.n;
            } else {   // exact16 (general epilogue): no output groups with sign-word masks
                const int tsp = tile % ntiles_sp;
                x = (tsp % a.tiles_x) * TW + (lane & 31);
                y0 = ((tsp / a.tiles_x) % a.tiles_y) * TH + row0;
                n = tsp / (a.tiles_x * a.tiles_y);
            }
            const unsigned xc = (unsigned)(x < a.w_ ? x : a.w_ - 1);
            const unsigned wpp = (unsigned)((a.cout + 31) >> 5);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int y = y0 + t;
                const unsigned p = ((unsigned)n * a.h + (unsigned)(y < a.h ? y : a.h - 1)) * a.w_ + xc;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const unsigned* mk = reinterpret_cast<const unsigned*>(a.mask);
                    if constexpr (CH) mk = reinterpret_cast<const unsigned*>(cj.job[job].aux);
                    mword[t][m] = mk[(size_t)p * wpp + (m < (int)wpp ? m : 0)];
                }
            }
        }
        // One stage.  An MX kernel runs its pair chunks as (f16 stage, MX stage) PAIRS of straight-line code (is_mx_c = the second of the
        // two): with the two stage bodies as alternatives of one branch inside the stage loop the register allocator splits the
        // accumulators' live ranges around the branch and spills them (217 registers of the cout-64 shape).
        auto run_stage = [&](int ck, auto is_mx_c) {
            constexpr bool IS_MX = decltype(is_mx_c)::value;
            // consumers only read LDS: a bare barrier (no vmcnt drain of the weight ring) is enough
            if (wave == 0) stamp(1);
            asm volatile("s_barrier" ::: "memory");
            if (wave == 0) stamp(1);
            const char* lbuf = smem + hbc * BUF;
            const int wsel = (!CH && nchunks <= C::NWB && ntiles == ntiles_sp) ? ck : par;   // resident weights: buffer = chunk; else the ring of two
            // halo rows of group gi+1 are read from LDS while the MFMAs of group gi run (ping-pong registers; NG is even)
            // when the register budget allows (PP); otherwise each group reads its own rows first
            constexpr int PP = (NWC == 4 || MT * NT * 16 + (NT + 2) * 8 <= 112) ? 1 : 0;   // 4-consumer shapes have a 256-register budget
            uint4 rowf[PP + 1][NT + 2];
            auto rload = [&](int slot, int gi) {
                const char* bp = lbuf + boff[gi % 3][gi / 3];
#pragma unroll
                for (int r = 0; r < NT + 2; ++r) rowf[slot][r] = *reinterpret_cast<const uint4*>(bp + r * (HW * PB));
            };
            if constexpr (IS_MX) {
                {
                    // ---- MX stage: both corrections of the chunk, K = 64 per tap -- nine scaled 8-bit MFMAs per output row ----
                    // B fragment of (row r, dx) = pieces h and 2 + h of the pixel's q record = the two k-step reads of an f16 stage;
                    // A fragment of tap (dy, dx) = the two 1 KB fragments (tap * 2 + {0, 1}) * MT + m of the chunk's MX block.
                    static_assert(KS == 2, "MX stage: 64-byte pixel records");
                    typedef int v8i __attribute__((ext_vector_type(8)));
                    typedef int v4i __attribute__((ext_vector_type(4)));
                    const char* wl = wlds + wsel * C::WBUF;
                    const int one = 0x7f7f7f7f;   // e8m0 1.0 for every block: bf8 operands carry their own exponents
                    auto ld8 = [&](const char* p0, const char* p1) -> v8i {
                        const v4i lo4 = *reinterpret_cast<const v4i*>(p0), hi4 = *reinterpret_cast<const v4i*>(p1);
                        return __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
                    };
                    // Units u = dx * 3 + dy in consumption order.  The weight fragments of unit u + 1 are requested before unit u's MFMAs
                    // (a ring of two); the halo rows of the next dx replace this dx's rows as the dy taps release them -- row 0 before the
                    // dy = 1 MFMAs, row 1 before the dy = 2 MFMAs, the rest behind them (one dx group's rows live at a time: the 8-wave
                    // shapes have 128 registers).
                    v8i wa[2][MT], qa[NT + 2];
                    auto wld = [&](int slot, int u) {
                        const int tap = (u % 3) * 3 + u / 3;
#pragma unroll
                        for (int m = 0; m < MT; ++m)
                            wa[slot][m] = ld8(wl + ((tap * 2 + 0) * MT + m) * 1024, wl + ((tap * 2 + 1) * MT + m) * 1024);
                    };
                    auto qld = [&](int r, int dx) {
                        qa[r] = ld8(lbuf + boff[dx][0] + r * (HW * PB), lbuf + boff[dx][1] + r * (HW * PB));
                    };
                    wld(0, 0);
#pragma unroll
                    for (int r = 0; r < NT + 2; ++r) qld(r, 0);
#pragma unroll
                    for (int u = 0; u < 9; ++u) {
                        const int dx = u / 3, dy = u % 3;
                        if (u + 1 < 9) wld((u + 1) & 1, u + 1);
                        if (dx < 2 && dy >= 1) qld(dy - 1, dx + 1);     // rows 0 / 1 were last read by the dy = 0 / dy = 1 taps of this dx
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int t = 0; t < NT; ++t)
#pragma unroll
                            for (int m = 0; m < MT; ++m)
                                acc[m][t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wa[u & 1][m], qa[t + dy], acc[m][t], 1, 1, 0, one, 0, one);   // cbsz / blgp 1: bf8
                        __builtin_amdgcn_sched_barrier(0);
                        if (dx < 2 && dy == 2) {
#pragma unroll
                            for (int r = 2; r < NT + 2; ++r) qld(r, dx + 1);
                        }
                    }
                    par ^= 1;
                    hbc = hbc + 1 == C::NHB ? 0 : hbc + 1;
                    if (wave == 0) stamp(1);
                    if constexpr (CH) publish();
                    return;
                }
            }
            if constexpr (SP != 0) {
                // ---- sparse taps: valid dy in {dy0, dy0+1}, valid dx in {dx0, dx0+1}; rows dy0 .. dy0+NT; 2*KS groups ----
                const int sub = SP == 1 ? (st_chunk(ck) * 32) / a.s2d_c : ((tile / ntiles_sp) * 64) / a.tap_c;   // wave-uniform
                auto sparse_stage = [&](auto si_c, auto sj_c) {
                    constexpr int SI = decltype(si_c)::value, SJ = decltype(sj_c)::value;
                    // forward: ky = 2*ty + i - 1 in [0,4)  ->  i = 0: ty in {1,2}, i = 1: ty in {0,1}; backward-data uses the
                    // flipped taps (pack.hip: t = 8 - tap)  ->  i = 0: {0,1}, i = 1: {1,2}
                    constexpr int dy0 = SP == 1 ? (SI == 0 ? 1 : 0) : (SI == 0 ? 0 : 1);
                    constexpr int dx0 = SP == 1 ? (SJ == 0 ? 1 : 0) : (SJ == 0 ? 0 : 1);
                    constexpr int NGS = 2 * KS;
                    uint4 rows[2][NT + 1], wq[2][MT];
                    auto rl = [&](int slot, int g) {          // g = ks * 2 + dxi
                        const char* bp = lbuf + boff[dx0 + (g & 1)][g >> 1] + dy0 * (HW * PB);
#pragma unroll
                        for (int r = 0; r < NT + 1; ++r) rows[slot][r] = *reinterpret_cast<const uint4*>(bp + r * (HW * PB));
                    };
                    auto wl = [&](int slot, int g, int dyi) {
                        const int ks = g >> 1, dx = dx0 + (g & 1), dy = dy0 + dyi;
#pragma unroll
                        for (int m = 0; m < MT; ++m)
                            wq[slot][m] = *reinterpret_cast<const uint4*>(wlds + wsel * C::WBUF + (((dy * 3 + dx) * KS + ks) * MT + m) * 1024);
                    };
                    rl(0, 0);
                    wl(0, 0, 0);
#pragma unroll
                    for (int g = 0; g < NGS; ++g) {
                        if (g + 1 < NGS) rl((g + 1) & 1, g + 1);
#pragma unroll
                        for (int dyi = 0; dyi < 2; ++dyi) {
                            const int u = g * 2 + dyi;
                            if (u + 1 < NGS * 2) wl((u + 1) & 1, (u + 1) >> 1, (u + 1) & 1);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int t = 0; t < NT; ++t)
#pragma unroll
                                for (int m = 0; m < MT; ++m)
                                    acc[m][t] = Frag<T>::mma(wq[u & 1][m], rows[g & 1][t + dyi], acc[m][t]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                };
                using I0 = std::integral_constant<int, 0>;
                using I1 = std::integral_constant<int, 1>;
                switch (sub) {
                    case 0: sparse_stage(I0{}, I0{}); break;
                    case 1: sparse_stage(I0{}, I1{}); break;
                    case 2: sparse_stage(I1{}, I0{}); break;
                    default: sparse_stage(I1{}, I1{}); break;
                }
                par ^= 1;
                if (!st_keeps_halo(ck)) hbc = hbc + 1 == C::NHB ? 0 : hbc + 1;   // X2 part 0: part 1 multiplies the same halo
                if (wave == 0) stamp(1);
                return;
            }
            // the stage's weights are in LDS once the barrier is passed: no prefetch across stages
#pragma unroll
            for (int u = 0; u < RING - 1; ++u) wload(u, wsel, u);
            if (PP) rload(0, 0);
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                if (!PP) rload(0, gi);
                else if (gi + 1 < NG) rload((gi + 1) & 1, gi + 1);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int u = gi * 3 + dy;
                    if (u + RING - 1 < NU) wload((u + RING - 1) % RING, wsel, u + RING - 1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (RESR_TIMING_TAPS >= 9 || (gi % 3) * 3 + dy < RESR_TIMING_TAPS) {
#pragma unroll
                        for (int t = 0; t < NT; ++t)
#pragma unroll
                            for (int m = 0; m < MT; ++m) {
                                acc[m][t] = Frag<T>::mma(wr[u % RING][m], rowf[PP ? (gi & 1) : 0][t + dy], acc[m][t]);
#pragma unroll
                                for (int v = 0; v < RESR_TIMING_VALU; ++v)   // (timing experiment: stand-in for the transform's packed adds)
                                    asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(timing_valu[v & 3]) : "v"(timing_valu[(v + 1) & 3]));
                            }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            par ^= 1;
            if (!st_keeps_halo(ck)) hbc = hbc + 1 == C::NHB ? 0 : hbc + 1;   // X2 part 0: part 1 multiplies the same halo
            if (wave == 0) stamp(1);
            if constexpr (CH) publish();
        };
        {
            int ck = 0;
            if constexpr (MX) {
                const int npair = nchunks < x2_p3 ? nchunks : x2_p3;   // the pair chunks' stages: (part 0, part 3) per chunk
                for (; ck < npair; ck += 2) {
                    run_stage(ck, std::false_type{});
                    run_stage(ck + 1, std::true_type{});
                }
            }
            for (; ck < nchunks; ++ck) run_stage(ck, std::false_type{});
        }

        if constexpr (FAST) {
            // ---- lean epilogue (same results as the general one below; about a third of its vector instructions) ----
            // MFMA leaves lane (lx, kh) with couts g*8 + kh*4 + 0..3 of each quad g; one permlane32_swap block per quad pair
            // regroups them IN PLACE into 8 consecutive couts ((2j + kh)*8 + 0..7) in 8 consecutive registers, so every
            // later step is a packed instruction on register pairs: LeakyReLU = max(v, slope*v) (v_pk_mul_f32 + v_max_f32),
            // residuals = v_fma_mix_f32 (reads the f16 directly) + v_pk_fma_f32, v_cvt_pk_f16_f32, the sign byte from the
            // converted halves (v_pk_sub_i16 clamp / v_perm_b32 / v_dot4_u32_u8), the sign-bit mask through a 16-entry
            // float4 table in LDS (one ds_read_b128 + two v_pk_mul_f32 per four values).
            const size_t goff = MT == 2 ? (size_t)c_grp * 64 : 0;
            const int x0 = c_tx * TW;
            const int y0 = c_ty * TH;
            const int n = c_n - c_grp * a.n;
            typedef const ConvArgs __attribute__((address_space(4))) * KernargPtr;
            KernargPtr ep = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(ep));
            struct {
                const char *res0, *res1;
                char* out;
                uint8_t* aux;
                int h, w_, cout, out_stride, res0_stride, res1_stride, flags;
                int out_chunk, res0_chunk, res1_chunk;
                float s0, t0, s1, t1, slope;
                long out_lo, res0_lo, res1_lo;   // X2: element offsets hi -> lo tensor
                long out_q;                      // MX: element offset hi -> q tensor of the output (0: none)
            } e;
            e.out_q = 0;
            if constexpr (X2) e.out_q = ep->out_q;
            if constexpr (CH && X2) e.out_q = cj.job[job].out_q;
            e.out = ep->out; e.h = ep->h; e.w_ = ep->w_; e.cout = ep->cout; e.out_stride = ep->out_stride;
            e.out_chunk = ep->out_chunk; e.flags = ep->flags; e.slope = ep->slope;
            if constexpr (X2) e.out_lo = ep->out_lo;
            if constexpr (CH) e.out = cj.job[job].out;
            if constexpr (CH && X2) e.out_lo = cj.job[job].out_lo;
            constexpr bool R0c = (EPI & 2) != 0, R1c = (EPI & 4) != 0, ESB = EPI == 16, EMB = EPI == 33;
            // What the instantiation fixes at compile time a chained launch may replace per job at run time: a job of kind 3 (one
            // half of the block's closing convolution) has residuals instead of LeakyReLU / sign words / mask.  Outside chained
            // launches r0 / r1 / own are constants and the code below is what it was.
            bool r0 = R0c, r1 = R1c, own = true;
            if constexpr (R0c) { e.res0 = ep->res0; e.res0_stride = ep->res0_stride; e.res0_chunk = ep->res0_chunk; e.s0 = ep->s0; e.t0 = ep->t0; if constexpr (X2) e.res0_lo = ep->res0_lo; }
            if constexpr (R1c) { e.res1 = ep->res1; e.res1_stride = ep->res1_stride; e.res1_chunk = ep->res1_chunk; e.s1 = ep->s1; e.t1 = ep->t1; if constexpr (X2) e.res1_lo = ep->res1_lo; }
            if constexpr (ESB) e.aux = ep->aux;
            if constexpr (ESB && CH) e.aux = reinterpret_cast<uint8_t*>(cj.job[job].aux);
            if constexpr (CH == 2) {
                if (cj.job[job].kind == 3) {
                    own = false;
                    r0 = true;
                    e.res0 = cj.job[job].res0; e.res0_stride = 32; e.res0_chunk = 0; e.s0 = cj.job[job].s0; e.t0 = cj.job[job].t0;
                    r1 = cj.job[job].res1 != nullptr;
                    e.res1 = cj.job[job].res1; e.res1_stride = 32; e.res1_chunk = 0; e.s1 = cj.job[job].s1; e.t1 = cj.job[job].t1;
                    if constexpr (X2) { e.res0_lo = cj.job[job].res0_lo; e.res1_lo = cj.job[job].res1_lo; }
                }
            }
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            const int lx_e = lane_e & 31, kh_e = lane_e >> 5;
            const int x = x0 + lx_e;
            const unsigned xc = (unsigned)(x < e.w_ ? x : e.w_ - 1);
            const bool f_lrelu = own && (e.flags & RESR_CONV_LRELU);
            const float2v sl2 = {e.slope, e.slope};
            const unsigned kh16 = (unsigned)kh_e * 16u;   // byte offset of this lane half's 8 channels inside a 16-channel piece pair
            // row double buffer of the residual pieces (cout 64): row t+1 is requested before row t is processed
            constexpr bool RESc = R0c || R1c;
            constexpr int NB = (RESc && MT == 2 && !(X2 && R1c)) ? 2 : 1;   // (exact16 with two residuals: four 16-B pieces per value pair -- no room for two rows)
            uint4v rr0[NB][MT][2], rr1[NB][MT][2];
            uint4v rr0l[X2 ? NB : 1][MT][2], rr1l[X2 ? NB : 1][MT][2];   // X2: the residuals' lo tensors
            auto row_pix = [&](int t) {
                const int y = y0 + row0 + t;
                return ((unsigned)n * e.h + (unsigned)(y < e.h ? y : e.h - 1)) * e.w_ + xc;
            };
            // pieces beyond cout (cout = 8, 16, 24 ... of a 32-channel tile) are neither read nor stored
            auto piece_ok = [&](int m, int j) { return m * 32 + (2 * j + kh_e) * 8 < e.cout; };
            // Residual loads are unconditional (a load under a divergent branch makes the compiler wait with vmcnt(0), which
            // also waits for the NEXT row's prefetch): a piece pair beyond cout reads the pixel's first pair instead
            // (wave-uniform select; cout is a multiple of 16 for these instantiations, host-checked) and is dropped at the store.
            auto request = [&](int b, int t) {
                const unsigned p = row_pix(t);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const bool pair_ok = m * 32 + j * 16 < e.cout;   // wave-uniform
                        if (r0) {
                            const char* q0 = e.res0 + ((size_t)p * e.res0_stride + goff + (pair_ok ? (size_t)m * e.res0_chunk + j * 16 : (size_t)0)) * 2 + kh16;
                            rr0[b][m][j] = *reinterpret_cast<const uint4v*>(q0);
                            if constexpr (X2) rr0l[b][m][j] = *reinterpret_cast<const uint4v*>(q0 + e.res0_lo * 2);
                        }
                        if (r1) {
                            const char* q1 = e.res1 + ((size_t)p * e.res1_stride + goff + (pair_ok ? (size_t)m * e.res1_chunk + j * 16 : (size_t)0)) * 2 + kh16;
                            rr1[b][m][j] = *reinterpret_cast<const uint4v*>(q1);
                            if constexpr (X2) rr1l[b][m][j] = *reinterpret_cast<const uint4v*>(q1 + e.res1_lo * 2);
                        }
                    }
            };
            // MFMA results -> first non-MFMA reader: the swaps are asm, so the compiler cannot count this hazard
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3");
            if (r0 || r1) request(0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int y = y0 + row0 + t;
                const bool in_img = y < e.h && x < e.w_;
                const unsigned p = row_pix(t);
                const int b = NB == 2 ? (t & 1) : 0;
                if constexpr (NB == 2) {
                    if (t + 1 < NT) request((t + 1) & 1, t + 1);
                } else {
                    if ((r0 || r1) && t > 0) request(0, t);
                }
                char* const orow = e.out + ((size_t)p * e.out_stride + goff) * 2 + kh16;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    unsigned sacc[2] = {0u, 0u};   // ESB: 128 x the sign byte of piece j
                    unsigned wk = 0;
                    if constexpr (EMB) {
                        if (own) wk = mword[t][m] >> (kh_e * 8);   // this lane half's bytes: j = 0 at bit 0, j = 1 at bit 16
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        float16v& A = acc[m][t];
                        float v0 = A[8 * j + 0], v1 = A[8 * j + 1], v2 = A[8 * j + 2], v3 = A[8 * j + 3];
                        float v4 = A[8 * j + 4], v5 = A[8 * j + 5], v6 = A[8 * j + 6], v7 = A[8 * j + 7];
                        permlane32_swap4(v0, v1, v2, v3, v4, v5, v6, v7);
                        float2v q[4] = {{v0, v1}, {v2, v3}, {v4, v5}, {v6, v7}};
                        if (EMB && own) {
                            const char* lut = smem + C::LUT_OFF;
                            const float4v ma = *reinterpret_cast<const float4v*>(lut + ((wk >> (16 * j)) & 15u) * 16);
                            const float4v mb = *reinterpret_cast<const float4v*>(lut + ((wk >> (16 * j + 4)) & 15u) * 16);
                            q[0] *= float2v{ma[0], ma[1]};
                            q[1] *= float2v{ma[2], ma[3]};
                            q[2] *= float2v{mb[0], mb[1]};
                            q[3] *= float2v{mb[2], mb[3]};
                        }
                        if constexpr (X2) {   // the packed weights (and the bias) carry 2^12; with the sign-word mask the table does
                            if (!(EMB && own)) {
                                const float2v k2 = {kLoInv, kLoInv};
#pragma unroll
                                for (int k = 0; k < 4; ++k) q[k] *= k2;
                            }
                        }
                        if (f_lrelu) {   // 0 <= slope <= 1 (host-checked): LeakyReLU(v) = max(v, slope * v)
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const float2v sq = q[k] * sl2;
                                q[k][0] = vmax_f32(q[k][0], sq[0]);
                                q[k][1] = vmax_f32(q[k][1], sq[1]);
                            }
                        }
                        if (r0) {
                            const float2v s02 = {e.s0, e.s0};
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                float2v tr = {mixmul_lo(rr0[b][m][j][k], e.t0), mixmul_hi(rr0[b][m][j][k], e.t0)};
                                if constexpr (X2) tr = float2v{mixfma_lo(rr0l[b][m][j][k], e.t0 * kLoInv, tr[0]), mixfma_hi(rr0l[b][m][j][k], e.t0 * kLoInv, tr[1])};
                                q[k] = __builtin_elementwise_fma(q[k], s02, tr);
                            }
                        }
                        if (r1) {
                            const float2v s12 = {e.s1, e.s1};
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                float2v tr = {mixmul_lo(rr1[b][m][j][k], e.t1), mixmul_hi(rr1[b][m][j][k], e.t1)};
                                if constexpr (X2) tr = float2v{mixfma_lo(rr1l[b][m][j][k], e.t1 * kLoInv, tr[0]), mixfma_hi(rr1l[b][m][j][k], e.t1 * kLoInv, tr[1])};
                                q[k] = __builtin_elementwise_fma(q[k], s12, tr);
                            }
                        }
                        uint4v d;
#pragma unroll
                        for (int k = 0; k < 4; ++k) d[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(q[k], half2v));
                        if (in_img && piece_ok(m, j))
                            *reinterpret_cast<uint4v*>(orow + ((size_t)m * e.out_chunk + j * 16) * 2) = d;
                        if (X2 && e.out_lo != 0) {   // lo = f16((v - hi) * 2^12), behind the hi tensor (out_lo = 0: a single f16 output)
                            uint4v dl;
                            const float2v k4 = {kLoScale, kLoScale};
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const float2v q4 = q[k] * k4;
                                dl[k] = lo_pair(d[k], q4[0], q4[1], -kLoScale);
                            }
                            if (in_img && piece_ok(m, j))
                                *reinterpret_cast<uint4v*>(orow + ((size_t)m * e.out_chunk + j * 16 + e.out_lo) * 2) = dl;
                            if constexpr (X2) {   // (any exact16 pass with a lean epilogue may emit it: the pass BEHIND it decides whether it reads q records)
                                if (e.out_q != 0) {
                                    // the q record of the pixel's chunk m: byte c = bf8(hi[c]), byte 32 + c = bf8(lo[c]) -- e5m2 of the f16 values
                                    // just stored, round to nearest even, unit scale; this lane's 8 channels start at (2 j + kh) * 8
                                    // (scalar copies first: __builtin_bit_cast on a vector-element lvalue reads element 0 in this toolchain)
                                    const unsigned dh0 = d[0], dh1 = d[1], dh2 = d[2], dh3 = d[3], dl0 = dl[0], dl1 = dl[1], dl2 = dl[2], dl3 = dl[3];
                                    short2v h01 = {0, 0}, h23 = {0, 0}, l01 = {0, 0}, l23 = {0, 0};
                                    h01 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(h01, __builtin_bit_cast(half2v, dh0), 1.f, false);
                                    h01 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(h01, __builtin_bit_cast(half2v, dh1), 1.f, true);
                                    h23 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(h23, __builtin_bit_cast(half2v, dh2), 1.f, false);
                                    h23 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(h23, __builtin_bit_cast(half2v, dh3), 1.f, true);
                                    l01 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(l01, __builtin_bit_cast(half2v, dl0), 1.f, false);
                                    l01 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(l01, __builtin_bit_cast(half2v, dl1), 1.f, true);
                                    l23 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(l23, __builtin_bit_cast(half2v, dl2), 1.f, false);
                                    l23 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(l23, __builtin_bit_cast(half2v, dl3), 1.f, true);
                                    char* const qrec = e.out + ((size_t)p * e.out_stride + (size_t)m * e.out_chunk + e.out_q) * 2 + (2 * j + kh_e) * 8;
                                    if (in_img && piece_ok(m, j)) {
                                        *reinterpret_cast<uint2v*>(qrec) = uint2v{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
                                        *reinterpret_cast<uint2v*>(qrec + 32) = uint2v{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
                                    }
                                }
                            }
                        }
                        if (ESB && own) {
                            if constexpr (X2) {
                                // the sign of the fp32 value (hi alone rounds |v| < 2^-25 to zero): its upper half as a 16-bit integer
                                // is positive exactly when v > 0 (down to the fp32 denormals)
                                unsigned u[4];
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    const float q0 = q[k][0], q1 = q[k][1];   // (__builtin_bit_cast on a vector-element lvalue reads element 0 in this toolchain)
                                    u[k] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, q1), __builtin_bit_cast(unsigned, q0), 0x07060302u);
                                }
                                sacc[j] = signs4_x128(u[0], u[1], 0x08040201u, 0u);
                                sacc[j] = signs4_x128(u[2], u[3], 0x80402010u, sacc[j]);
                            } else {
                                sacc[j] = signs4_x128(d[0], d[1], 0x08040201u, 0u);
                                sacc[j] = signs4_x128(d[2], d[3], 0x80402010u, sacc[j]);
                            }
                        }
                    }
                    if (ESB && own) {
                        // bytes 2j + kh of the pixel's word: (byte_0 | byte_1 << 16) << 8 kh, all of it times 128 so far
                        const unsigned both = sacc[0] | (sacc[1] << 16);
                        const unsigned mine = __builtin_amdgcn_alignbit(both, both, kh_e ? 31u : 7u);   // rotate: >> 7 or << 1
                        float sa = __builtin_bit_cast(float, mine), sb = sa;
                        permlane32_swap(sa, sb);
                        const unsigned word = __builtin_bit_cast(unsigned, sa) | __builtin_bit_cast(unsigned, sb);
                        if (in_img && kh_e == 0 && m * 32 < e.cout)
                            reinterpret_cast<unsigned*>(e.aux)[(size_t)p * (size_t)((e.cout + 31) >> 5) + m] = word;
                    }
                }
            }
            if constexpr (CH) {
                if (job + 1 < njobs_c) {   // the last job's tiles have no reader inside the launch
                    pend = true;
                    pend_tile = tile;
                    pend_tag = chain_epoch + (unsigned)job + 1u;
                    if (CH != 3 && tile + tile_step >= tile_end)   // the next tile belongs to the next job
                        bias_cur = reinterpret_cast<const float*>(smem + C::CHAIN_OFF) + (job + 1) * 32;
                }
            }
            group_bias(tile + tile_step);
            init_acc();
            // A workgroup with ONE tile per job (the 64^2 training crops) publishes right here, behind the acknowledgement of the
            // stores it has just issued: its neighbours' producers poll for this plane during the very next stage, and a flag that
            // only appears at that stage's end costs them that stage plus the poll's round trips (measured: 4.2 us in the poll at the
            // first job boundary of such a launch, 0.6 us with this).  The wait overlaps the barrier wait for the next job's first stage.
            if constexpr (CH == 1 || CH == 2) {
                if (first + tile_step >= tile_end) publish();
            }
            if (wave == 0) stamp(1);  // tile done
            continue;
        }
        // ---- epilogue: lane owns pixel (row0+t, lx) and 4 consecutive couts per accumulator quad ----
        const int tsp = tile % ntiles_sp;
        const size_t goff = (size_t)(tile / ntiles_sp) * 64;   // element offset of the tile's output group inside a pixel
        const int x0 = (tsp % a.tiles_x) * TW;
        const int y0 = ((tsp / a.tiles_x) % a.tiles_y) * TH;
        const int n = tsp / (a.tiles_x * a.tiles_y);
        // The epilogue's arguments and per-lane indices are re-read / re-derived per tile through opaque copies:
        // hoisted out of the tile loop they would sit in registers across the MFMA loop.
        typedef const ConvArgs __attribute__((address_space(4))) * KernargPtr;
        KernargPtr ep = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ep));
        // one batch of scalar loads (one wait) instead of a reload at every use
        struct {
            const char *res0, *res1, *mask;
            char* out;
            uint8_t* aux;
            int h, w_, cout, out_stride, res0_stride, res1_stride, mask_stride, flags;
            int out_chunk, res0_chunk, res1_chunk, mask_chunk;
            float s0, t0, s1, t1, slope;
            long out_lo, res0_lo, res1_lo, mask_lo;
        } e;
        e.res0 = ep->res0; e.res1 = ep->res1; e.mask = ep->mask; e.out = ep->out; e.aux = ep->aux;
        e.h = ep->h; e.w_ = ep->w_; e.cout = ep->cout; e.out_stride = ep->out_stride;
        e.res0_stride = ep->res0_stride; e.res1_stride = ep->res1_stride; e.mask_stride = ep->mask_stride;
        e.out_chunk = ep->out_chunk; e.res0_chunk = ep->res0_chunk; e.res1_chunk = ep->res1_chunk; e.mask_chunk = ep->mask_chunk;
        if constexpr (X2) { e.out_lo = ep->out_lo; e.res0_lo = ep->res0_lo; e.res1_lo = ep->res1_lo; e.mask_lo = ep->mask_lo; }
        e.flags = ep->flags; e.s0 = ep->s0; e.t0 = ep->t0; e.s1 = ep->s1; e.t1 = ep->t1; e.slope = ep->slope;
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int lx_e = lane_e & 31, kh_e = lane_e >> 5;
        constexpr bool EX = (EPI & 8) != 0, EM = (EPI & 1) != 0, ER = (EPI & 6) != 0;
        constexpr bool ESB = (EPI & 16) != 0, EMB = (EPI & 32) != 0;   // write sign bits / mask given as sign bits
        const bool f_lrelu = e.flags & RESR_CONV_LRELU, f_clamp = EX && (e.flags & RESR_CONV_CLAMP01);
        const bool f_nchw = EX && (e.flags & RESR_CONV_OUT_NCHW_F32);
        const bool f_mask = EX ? (e.flags & RESR_CONV_MASK) != 0 : EM;
        // EPI bit 6: the aux tensor is stored (compile-time forms of the discriminator's two uses: before the residual of
        // an up block -- EPI 66 --, before the mask of a backward-data pass -- EPI 65)
        constexpr bool EA = (EPI & 64) != 0;
        const bool f_aux_mask = EX ? ((e.flags & RESR_CONV_AUX_BEFORE_MASK) && e.aux && !f_nchw) : (EA && EM && !ER);
        const bool f_aux_res = EX ? ((e.flags & RESR_CONV_AUX_BEFORE_RES) && e.aux && !f_nchw) : (EA && ER);
        const bool f_res0 = EX ? e.res0 != nullptr : (EPI & 2) != 0, f_res1 = EX ? e.res1 != nullptr : (EPI & 4) != 0;
        const int x = x0 + lx_e;
        // MFMA results -> first non-MFMA reader: the swaps below are asm, so the compiler cannot count this hazard
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3");
        // MFMA leaves lane (lx, kh) with couts g*8 + kh*4 + 0..3 of each quad g.  v_permlane32_swap between lanes lx and
        // lx+32 regroups a pair of quads (2j, 2j+1) into 8 consecutive couts per lane -- (2j + kh)*8 + 0..7 -- so masks,
        // residuals and the result move as 16-byte pieces (half the L2 write transactions of 8-byte pieces).
        // Per (row, cout tile): every epilogue input is requested before the first store, so the batch pays one
        // memory latency instead of one per piece (loads behind stores wait for the stores).
        static_assert(sizeof(T) == 2, "the 8-channel epilogue assumes f16 storage");
        // Every epilogue input of a batch is requested before the batch's first store (loads behind stores wait for the
        // stores' acks on the in-order counter, one memory round trip per batch).  cout 64: the batch is a whole row -- four
        // pieces with up to two residuals -- and row t+1 is requested before row t is processed (64 registers that are free
        // once the MFMA loop is over; 16 exposed round trips per tile were 3.3 us); cout <= 32 (tighter register budget): both
        // pieces when there is only a mask, one piece with residuals.
        constexpr bool RB = MT == 2 && !EX && !X2;
        constexpr int JB = (RB || !(ER || EX)) ? 2 : 1;
        constexpr int MA = RB ? MT : 1;   // without row batching one m tile's inputs are live at a time
        constexpr int NB = RB ? 2 : 1;    // row double buffer
        half8 rmask[NB][MA][2], rres0[NB][MA][2], rres1[NB][MA][2];
        half8 rres0l[NB][MA][2], rres1l[NB][MA][2];   // X2: the residuals' lo tensors
        // pieces outside the image / beyond cout read a clamped (valid) address and are dropped at the store
        auto row_p = [&](int t) {
            const int y = y0 + row0 + t;
            return ((size_t)n * e.h + (y < e.h ? y : e.h - 1)) * e.w_ + (x < e.w_ ? x : e.w_ - 1);
        };
        // element offset of piece (m, j) inside a pixel of an operand with chunk stride cs (clamped when beyond cout)
        auto poff = [&](int m, int j, int cs) {
            return goff + (m * 32 + (2 * j + kh_e) * 8 < e.cout ? (size_t)m * cs + (2 * j + kh_e) * 8 : (size_t)0);
        };
        auto request = [&](int b, size_t p, int m, int j0) {
            if (EMB) {
                // (the sign words of the tile's rows were requested at the top of the tile: mword)
            } else if (f_mask) {
#pragma unroll
                for (int j = j0; j < j0 + JB; ++j)
                    rmask[b][m % MA][j] = *reinterpret_cast<const half8*>(e.mask + (p * e.mask_stride + poff(m, j, e.mask_chunk)) * 2);
            }
            if (f_res0) {
#pragma unroll
                for (int j = j0; j < j0 + JB; ++j)
                    rres0[b][m % MA][j] = *reinterpret_cast<const half8*>(e.res0 + (p * e.res0_stride + poff(m, j, e.res0_chunk)) * 2);
                if constexpr (X2) {
#pragma unroll
                    for (int j = j0; j < j0 + JB; ++j)
                        rres0l[b][m % MA][j] = *reinterpret_cast<const half8*>(e.res0 + (p * e.res0_stride + poff(m, j, e.res0_chunk) + e.res0_lo) * 2);
                }
            }
            if (f_res1) {
#pragma unroll
                for (int j = j0; j < j0 + JB; ++j)
                    rres1[b][m % MA][j] = *reinterpret_cast<const half8*>(e.res1 + (p * e.res1_stride + poff(m, j, e.res1_chunk)) * 2);
                if constexpr (X2) {
#pragma unroll
                    for (int j = j0; j < j0 + JB; ++j)
                        rres1l[b][m % MA][j] = *reinterpret_cast<const half8*>(e.res1 + (p * e.res1_stride + poff(m, j, e.res1_chunk) + e.res1_lo) * 2);
                }
            }
        };
        if (RB) {
#pragma unroll
            for (int m = 0; m < MT; ++m) request(0, row_p(0), m, 0);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int y = y0 + row0 + t;
            const bool in_img = y < e.h && x < e.w_;
            const size_t p = row_p(t);
            constexpr int bsel = 0;
            const int b = RB ? (t & 1) : bsel;
            if (RB && t + 1 < NT) {
#pragma unroll
                for (int m = 0; m < MT; ++m) request((t + 1) & 1, row_p(t + 1), m, 0);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                int co[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) co[j] = m * 32 + (2 * j + kh_e) * 8;
                unsigned sbits = 0;   // ESB: this lane's bytes of the sign word
#pragma unroll
                for (int j0 = 0; j0 < 2; j0 += JB) {
                if (!RB) request(0, p, m, j0);
#pragma unroll
                for (int j = j0; j < j0 + JB; ++j) {
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = acc[m][t][(2 * j) * 4 + r];
                        v[4 + r] = acc[m][t][(2 * j + 1) * 4 + r];
                        permlane32_swap(v[r], v[4 + r]);
                    }
                    if constexpr (X2) {   // the packed weights (and the bias) carry 2^12
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] *= kLoInv;
                    }
                    const bool ok = in_img && co[j] < e.cout;
                    auto store8 = [&](char* base, size_t idx) {
                        half8 h;
#pragma unroll
                        for (int r = 0; r < 8; ++r) h[r] = (half_t)v[r];
                        *reinterpret_cast<half8*>(base + idx * 2) = h;
                        if (X2 && e.out_lo != 0) {   // (out_lo = 0: a single f16 output)
                            half8 l;
#pragma unroll
                            for (int r = 0; r < 8; ++r) l[r] = (half_t)((v[r] - (float)h[r]) * kLoScale);
                            *reinterpret_cast<half8*>(base + (idx + e.out_lo) * 2) = l;
                        }
                    };
                    if (f_aux_mask && ok) store8(reinterpret_cast<char*>(e.aux), p * e.out_stride + poff(m, j, e.out_chunk));
                    if (EMB) {
                        const unsigned byte = mword[EPI == 33 ? t : 0][m] >> (8 * (2 * j + kh_e));
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] *= ((byte >> r) & 1u) ? 1.f : e.slope;
                    } else if (f_mask) {
                        half8 mk = rmask[b][m % MA][j];
                        if constexpr (X2) {
                            // the mask is a saved activation (a pair): a hi value that rounded to zero (|v| < 2^-25) defers to the lo tensor,
                            // which is read only then (common.h pair_positive)
                            bool anyz = false;
#pragma unroll
                            for (int r = 0; r < 8; ++r) anyz = anyz || (float)mk[r] == 0.f;
                            if (anyz && e.mask_lo != 0) {
                                const half8 ml = *reinterpret_cast<const half8*>(e.mask + (p * e.mask_stride + poff(m, j, e.mask_chunk) + e.mask_lo) * 2);
#pragma unroll
                                for (int r = 0; r < 8; ++r) mk[r] = (float)mk[r] == 0.f ? ml[r] : mk[r];
                            }
                        }
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] *= ((float)mk[r] > 0.f ? 1.f : e.slope);
                    }
                    if (f_lrelu) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] = v[r] > 0.f ? v[r] : v[r] * e.slope;
                    }
                    if (f_aux_res && ok) store8(reinterpret_cast<char*>(e.aux), p * e.out_stride + poff(m, j, e.out_chunk));
                    if (f_res0) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            float rv = (float)rres0[b][m % MA][j][r];
                            if constexpr (X2) rv = __builtin_fmaf((float)rres0l[b][m % MA][j][r], kLoInv, rv);
                            v[r] = __builtin_fmaf(v[r], e.s0, e.t0 * rv);   // explicit: one rounding, the same in every instantiation
                        }
                    }
                    if (f_res1) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            float rv = (float)rres1[b][m % MA][j][r];
                            if constexpr (X2) rv = __builtin_fmaf((float)rres1l[b][m % MA][j][r], kLoInv, rv);
                            v[r] = __builtin_fmaf(v[r], e.s1, e.t1 * rv);
                        }
                    }
                    if (f_nchw) {
                        float* o = reinterpret_cast<float*>(e.out);
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            if (!in_img || co[j] + r >= e.cout) continue;
                            const size_t q = (((size_t)n * e.cout + co[j] + r) * e.h + y) * e.w_ + x;
                            float u = v[r];
                            if (f_clamp) {
                                if (e.aux) e.aux[q] = (u >= 0.f && u <= 1.f) ? 1 : 0;
                                u = u < 0.f ? 0.f : (u > 1.f ? 1.f : u);   // torch.clamp_ semantics: a NaN stays a NaN (fminf / fmaxf would drop it)
                            }
                            o[q] = u;
                        }
                    } else {
                        if (f_clamp) {
#pragma unroll
                            for (int r = 0; r < 8; ++r) v[r] = v[r] < 0.f ? 0.f : (v[r] > 1.f ? 1.f : v[r]);
                        }
                        if (ok) store8(e.out, p * e.out_stride + poff(m, j, e.out_chunk));
                        if (ESB) {
#pragma unroll
                            for (int r = 0; r < 8; ++r) sbits |= ((X2 ? v[r] : (float)(half_t)v[r]) > 0.f ? 1u : 0u) << (8 * (2 * j + kh_e) + r);
                        }
                    }
                }
                }
                if (ESB) {   // the two lanes of a pixel hold complementary bytes: merge, lane kh = 0 stores the word
                    float sa = __builtin_bit_cast(float, sbits), sb = sa;
                    permlane32_swap(sa, sb);
                    const unsigned word = __builtin_bit_cast(unsigned, sa) | __builtin_bit_cast(unsigned, sb);
                    if (in_img && kh_e == 0 && m * 32 < e.cout)
                        reinterpret_cast<unsigned*>(e.aux)[p * (size_t)((e.cout + 31) >> 5) + m] = word;
                }
            }
        }
        group_bias(tile + tile_step);
        init_acc();
        if (wave == 0) stamp(1);  // tile done
    }
    }   // jobs
    if (wave == 0) stamp_dump(1);
    if constexpr (CH == 3) publish();   // a pinned workgroup's last tile has no next stage to publish it from
    if constexpr (CH != 0) {
        // The last workgroup of the launch to get here prepares the state for the next launch on it (which starts after this
        // kernel has ended, so these plain stores are visible to it): tickets and the finish counter back to 0, the epoch
        // 8 further -- or, far from wrapping the signed flag comparison, every flag back to 0 and the epoch with them.
        if (wave == 0) {
            unsigned last = 0;
            if (lane == 0) last = (atomicAdd(cj.state + 1, 1u) == (unsigned)G - 1u) ? 1u : 0u;
            if (__builtin_amdgcn_readfirstlane(last)) {
                unsigned e = chain_epoch + 8u;
                if (e > 0x70000000u) {
                    for (unsigned i = lane; i < cj.cap; i += 64) cj.state[kChainHdr + i] = 0u;
                    e = 0u;
                }
                if (lane < 8) cj.state[8 + lane] = 0u;
                if (lane == 0) { cj.state[1] = 0u; cj.state[0] = e; }
            }
        }
    }
}

template <typename T, int MT, int NT, int NWC, int EPI, int X2 = 0, int SP = 0>
static int launch_ws_epi(const ConvArgs& a, hipStream_t stream) {
    using C = WsCfg<T, MT, NT, NWC>;
    ConvArgs args = a;
    args.tiles_x = (a.w_ + 31) / 32;
    args.tiles_y = (a.h + C::TH - 1) / C::TH;
    // halo buffers, bias, weight buffers (+ the mask-multiplier table; + the biases of the output groups of a grouped launch)
    const bool gbias = MT == 2 && a.ngroups > 1 && !(a.flags & RESR_CONV_NO_BIAS) && a.bias != nullptr;
    const size_t lds = C::LDS_BYTES + ((EPI == 33 || gbias) ? C::LUT_BYTES : 0) + (gbias ? C::GB_BYTES : 0);
    // per device (the boundary is callable with any current device): workgroups the device holds at once and the
    // address of this translation unit's zero page there; idempotent, so a race between two first calls is benign
    static int resident_dev[kMaxDevices] = {0};
    static const char* zero_dev[kMaxDevices] = {nullptr};
    int cur_dev = 0;
    if (hipGetDevice(&cur_dev) != hipSuccess || cur_dev < 0 || cur_dev >= kMaxDevices) return fail(RESR_ERR_LAUNCH, "conv3x3: hipGetDevice");
    int& resident = resident_dev[cur_dev];
    const char*& zero = zero_dev[cur_dev];
    if (!resident) {
        // (the attribute is an upper bound set once per device: the largest request this instantiation can make)
        const size_t lds_max = C::LDS_BYTES + C::LUT_BYTES + C::GB_BYTES;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_ws_kernel<T, MT, NT, NWC, EPI, X2, SP>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
        int per_cu = 0;
        hipDeviceProp_t prop;
        void* zp = nullptr;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv3x3_ws_kernel<T, MT, NT, NWC, EPI, X2, SP>, C::NTHR, lds_max) != hipSuccess ||
            hipGetDeviceProperties(&prop, cur_dev) != hipSuccess || per_cu <= 0 ||
            hipGetSymbolAddress(&zp, HIP_SYMBOL(g_conv_zero16)) != hipSuccess || !zp)
            return fail(RESR_ERR_LAUNCH, "conv3x3: occupancy / zero-page query failed");
        zero = (const char*)zp;
        resident = per_cu * prop.multiProcessorCount;
        if (getenv("RESR_DEBUG_OCC")) fprintf(stderr, "conv3x3_ws<%d,%d,%d,%d,%d>: %d workgroups/CU, lds %zu\n", (int)sizeof(T), MT, NT, NWC, EPI, per_cu, lds);
    }
    args.zero = zero;
    static const char* trace_chain_only = getenv("RESR_TRACE_CHAIN_ONLY");   // debug: keep the timeline buffer for the chained launches
    args.trace = trace_chain_only ? nullptr : g_conv_trace;
    const int ntiles = args.tiles_x * args.tiles_y * a.n * (MT == 2 && a.ngroups > 1 ? a.ngroups : 1);
    const unsigned grid = (unsigned)(ntiles < resident ? ntiles : resident);
    prof_before(stream);
    hipLaunchKernelGGL((conv3x3_ws_kernel<T, MT, NT, NWC, EPI, X2, SP>), dim3(grid), dim3(C::NTHR), lds, stream, args, ChainNone{});
    // sparse-tap launches count the 16 real tap-products of the 4x4 kernel (4 of 9 taps per chunk)
    prof_after(stream, (X2 ? 25000 : 20000) + (SP ? 2000 : 0) + MT * 100 + NT * 10 + NWC,
               2.0 * (SP ? 4 : 9) * a.cin * a.cout * (a.ngroups > 1 ? a.ngroups : 1) * (double)a.n * a.h * a.w_,
               conv_algorithmic_bytes(a, sizeof(T) * (X2 ? 2 : 1)));
    RESR_CHECK_LAUNCH("conv3x3_ws_kernel");
    return RESR_OK;
}

template <typename T, int MT, int NT, int NWC, int X2 = 0>
static int launch_ws(const ConvArgs& a, hipStream_t stream) {
    if constexpr (X2 == 2) {   // MX stages (RESR_CONV_MX_PAIRS): the lean epilogues -- bias / LeakyReLU / residuals, or the sign-word mask of a backward-data pass
        if ((a.flags & RESR_CONV_MASK_BITS) && !a.aux && !a.res0 && !a.res1 && !(a.flags & ~(RESR_CONV_MASK | RESR_CONV_MASK_BITS | RESR_CONV_NO_BIAS)))
            return launch_ws_epi<T, MT, NT, NWC, 33, X2>(a, stream);
        const bool lean_ok = !a.aux && !a.mask && !(a.flags & ~(RESR_CONV_LRELU | RESR_CONV_NO_BIAS | RESR_CONV_UPSAMPLE_IN)) &&
                             !((a.flags & RESR_CONV_LRELU) && !(a.slope >= 0.f && a.slope <= 1.f)) && !((a.res0 || a.res1) && (a.cout & 15));
        if (!lean_ok) return fail(RESR_ERR_ARG, "conv3x3: RESR_CONV_MX_PAIRS supports bias / LeakyReLU (0 <= slope <= 1) / residual / sign-word-mask epilogues only");
        if (a.res0 && a.res1) return launch_ws_epi<T, MT, NT, NWC, 6, X2>(a, stream);
        if (a.res0) return launch_ws_epi<T, MT, NT, NWC, 2, X2>(a, stream);
        if (a.res1) return fail(RESR_ERR_ARG, "conv3x3: res1 without res0");
        return launch_ws_epi<T, MT, NT, NWC, 0, X2>(a, stream);
    } else {
    const int combo = ((a.flags & RESR_CONV_MASK) ? 1 : 0) | (a.res0 ? 2 : 0) | (a.res1 ? 4 : 0);
    if (a.flags & RESR_CONV_WRITE_SIGNBITS) {   // forward conv + LeakyReLU that also emits its 1-bit mask (checked by the caller)
        if (!X2 && (a.flags & RESR_CONV_LRELU) && !(a.slope >= 0.f && a.slope <= 1.f)) return fail(RESR_ERR_ARG, "conv3x3: sign-bit output needs 0 <= slope <= 1");
        return launch_ws_epi<T, MT, NT, NWC, 16, X2>(a, stream);
    }
    if (a.flags & RESR_CONV_MASK_BITS) return launch_ws_epi<T, MT, NT, NWC, 33, X2>(a, stream);
    if constexpr (MT == 2 && !X2) {   // the discriminator's aux-storing passes (cout 64 groups), specialised like the plain ones
        if (a.aux && !(a.flags & (RESR_CONV_OUT_NCHW_F32 | RESR_CONV_CLAMP01))) {
            if (combo == 2 && (a.flags & RESR_CONV_AUX_BEFORE_RES) && !(a.flags & RESR_CONV_AUX_BEFORE_MASK))
                return launch_ws_epi<T, MT, NT, NWC, 66, X2>(a, stream);
            if (combo == 1 && (a.flags & RESR_CONV_AUX_BEFORE_MASK) && !(a.flags & RESR_CONV_AUX_BEFORE_RES))
                return launch_ws_epi<T, MT, NT, NWC, 65, X2>(a, stream);
        }
    }
    // the lean epilogue computes LeakyReLU as max(v, slope * v): slopes outside [0, 1] take the general one
    const bool odd_slope = (a.flags & RESR_CONV_LRELU) && !(a.slope >= 0.f && a.slope <= 1.f);
    const bool odd_cout = (a.res0 || a.res1) && (a.cout & 15);   // its residual loads select whole 16-channel piece pairs
    const bool extras = odd_slope || odd_cout || a.aux || (a.flags & (RESR_CONV_OUT_NCHW_F32 | RESR_CONV_CLAMP01));
    if (a.out_q != 0 && (extras || !(combo == 0 || combo == 2 || combo == 6)))   // (the sign-word forms above are lean as well)
        return fail(RESR_ERR_ARG, "conv3x3: out_q_offset needs a lean epilogue (bias / LeakyReLU / residuals / sign words)");
    if (!extras) switch (combo) {
        case 0: return launch_ws_epi<T, MT, NT, NWC, 0, X2>(a, stream);  // forward convs 1-4, upsampling, D forward
        case 1: return launch_ws_epi<T, MT, NT, NWC, 1, X2>(a, stream);  // backward-data through a LeakyReLU
        case 2: return launch_ws_epi<T, MT, NT, NWC, 2, X2>(a, stream);  // conv5 of a dense block
        case 6: return launch_ws_epi<T, MT, NT, NWC, 6, X2>(a, stream);  // conv5 closing an RRDB / its backward
        case 3: return launch_ws_epi<T, MT, NT, NWC, 3, X2>(a, stream);  // masked backward with gradient accumulation
        default: break;
    }
    return launch_ws_epi<T, MT, NT, NWC, 15, X2>(a, stream);
    }
}


}  // namespace resr
