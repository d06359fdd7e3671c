// conv3x3_ws.hip -- entry of the fast-mode (f16) producer/consumer convolution (kernel: conv3x3_ws.h; instantiations:
// conv3x3_ws_mt1.hip for cout <= 32, conv3x3_ws_mt2.hip for cout 64 -- two translation units so they build in parallel).
#include <stdlib.h>

#include <mutex>
#include <vector>

#include "conv3x3.h"

namespace resr {

unsigned long long* g_conv_trace = nullptr;
void conv_trace_set(void* p) { g_conv_trace = (unsigned long long*)p; }

int conv3x3_ws_mt1(const ConvArgs& a, int tile_rows, hipStream_t stream);
int conv3x3_ws_mt2(const ConvArgs& a, int tile_rows, hipStream_t stream);
int conv3x3_ws_x2_mt1(const ConvArgs& a, int tile_rows, hipStream_t stream);   // RESR_F16X2 instantiations
int conv3x3_ws_x2_mt2(const ConvArgs& a, int tile_rows, hipStream_t stream);

// Preconditions of the producer's 24 x 24-bit offsets and of the 8-channel epilogue; otherwise the caller uses the
// one-role kernel.
// The chained launches split the image range over 8 XCDs of 32 CUs (conv3x3_ws.h, CH): only on the full device (an
// MI355X in SPX mode reports 256 CUs; a partitioned device has one XCD per agent).
bool conv3x3_chain_device_ok() {
    static int ok_dev[kMaxDevices] = {0};   // 0 unknown, 1 yes, -1 no
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return false;
    if (!ok_dev[dev]) {
        hipDeviceProp_t prop;
        ok_dev[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount == 256) ? 1 : -1;
    }
    return ok_dev[dev] > 0;
}

bool conv3x3_ws_supported(const ConvArgs& a) {
    const size_t px = (size_t)a.n * a.hs * a.ws;
    const size_t stride = (size_t)(a.in0_stride_b > a.in1_stride_b ? a.in0_stride_b : a.in1_stride_b);
    if (px > (1u << 24) || stride >= (1u << 24) || px * stride + 64 >= (1ull << 32)) return false;
    if (!(a.flags & RESR_CONV_OUT_NCHW_F32) && (a.cout & 7)) return false;
    return true;
}

// Tile height: the tallest tile (least halo, most weight reuse) that still gives every CU a workgroup; small images
// (GAN crops, the discriminator's coarse levels) fall back to shorter tiles instead of leaving CUs idle.
static int pick_rows(const ConvArgs& a, const int* rows, int nrows) {
    static const char* force = getenv("RESR_CONV_TILE_ROWS");   // experiment knob: force the tile height (16 or 8)
    if (force) return atoi(force);
    static int cus_dev[kMaxDevices] = {0};   // per device; idempotent
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
    int& cus = cus_dev[dev];
    if (!cus) {
        hipDeviceProp_t prop;
        cus = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
    }
    for (int i = 0; i < nrows; ++i) {
        const long tiles = (long)((a.w_ + 31) / 32) * ((a.h + rows[i] - 1) / rows[i]) * a.n * (a.ngroups > 1 ? a.ngroups : 1);
        if (tiles >= cus) return rows[i];
    }
    return rows[nrows - 1];
}

int conv3x3_ws_sparse(const ConvArgs& a, int tile_rows, int sp, hipStream_t stream);   // conv3x3_ws_sp.hip
int conv3x3_ws_chain_launch(const ConvArgs& a, const ChainArgs& cj, int tile_rows, int kind, unsigned* ticket_base, double flop, double bytes, hipStream_t stream);   // conv3x3_ws_chain.hip

// ---- chained dense-block passes (conv3x3_ws.h, CH): per-(device, stream) progress flags ----
// flags[tile] only ever grows: a launch's jobs publish epoch + 1 ... epoch + njobs, and the next launch on the stream
// starts 8 higher, so nothing is reset between launches.  One buffer per stream: two streams would read each other's epochs.
namespace {
struct ChainState {
    int dev;
    hipStream_t stream;
    unsigned* buf;      // [cap] flags + [2] error counters + [8] per-XCD workgroup tickets
    size_t cap;
    unsigned epoch;
    unsigned ticket_base;   // value of the ticket counters before the next launch
};
std::mutex g_chain_mu;
std::vector<ChainState> g_chain;
}  // namespace

// sum over all streams of (poll time-outs, misplaced workgroups); synchronises the device (debug / test entry)
long long conv3x3_chain_errors() {
    std::lock_guard<std::mutex> lk(g_chain_mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    long long polls = 0, xcd = 0;
    for (const ChainState& c : g_chain) {
        if (c.dev != dev) continue;
        unsigned e[16] = {0};
        if (hipMemcpy(e, c.buf + c.cap, sizeof(e), hipMemcpyDeviceToHost) != hipSuccess) return -1;
        polls += e[0];
        xcd += e[1];
        if (getenv("RESR_DEBUG_CHAIN"))
            fprintf(stderr, "chain state: stream %p cap %zu epoch %u ticket_base %u | time-outs %u beyond-share %u | tickets %u %u %u %u %u %u %u %u\n",
                    (void*)c.stream, c.cap, c.epoch, c.ticket_base, e[0], e[1], e[8], e[9], e[10], e[11], e[12], e[13], e[14], e[15]);
    }
    return polls + (xcd << 32);
}

int conv3x3_ws_chain_f16(const ConvArgs& a, const ChainJob* jobs, int njobs, const double* flop, const double* bytes, hipStream_t stream) {
    static const int rows1[] = {16, 8};
    int rows = pick_rows(a, rows1, 2);
    // Experiment (RESR_CHAIN_PIPE="5,7,9,11"): a pinned pipeline instead of a walk -- 8-row tiles, the 32 workgroups of an XCD split
    // over the four jobs as given, each workgroup running ONE job over the XCD's bands top to bottom behind its predecessor's
    // flags, so that the planes a block's passes exchange stay inside the XCD's L2
    int split[kMaxChain + 2] = {0};
    const char* pipe_env = getenv("RESR_CHAIN_PIPE");   // read per call (a test flips it)
    if (pipe_env && njobs == 4) {
        int c[4] = {0, 0, 0, 0};
        if (sscanf(pipe_env, "%d,%d,%d,%d", &c[0], &c[1], &c[2], &c[3]) == 4 && c[0] > 0 && c[1] > 0 && c[2] > 0 && c[3] > 0 &&
            c[0] + c[1] + c[2] + c[3] == 32 && (size_t)((a.w_ + 31) / 32) * ((a.h + 7) / 8) * a.n >= 256 * 8) {
            rows = 8;
            for (int j = 0; j < 4; ++j) split[j + 1] = split[j] + c[j];
            for (int j = 5; j < kMaxChain + 2; ++j) split[j] = 32;
        }
    }
    const size_t ntiles = (size_t)((a.w_ + 31) / 32) * ((a.h + rows - 1) / rows) * a.n;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(RESR_ERR_LAUNCH, "conv3x3_chain: hipGetDevice");
    ChainArgs cj;
    memset(&cj, 0, sizeof(cj));
    {
        std::lock_guard<std::mutex> lk(g_chain_mu);
        // A chained launch waits for its own workgroups from inside: two of them running side by side on different streams
        // could each hold CUs the other's missing workgroups need.  One stream per device owns chaining at a time; a
        // launch from another stream first drains the device (once per change of owner, not per launch).
        static hipStream_t owner[kMaxDevices] = {nullptr};
        static bool owned[kMaxDevices] = {false};
        if (dev >= 0 && dev < kMaxDevices) {
            if (owned[dev] && owner[dev] != stream && hipDeviceSynchronize() != hipSuccess)   // (the old owner may be gone: drain the device)
                return fail(RESR_ERR_LAUNCH, "conv3x3_chain: hipDeviceSynchronize");
            owner[dev] = stream;
            owned[dev] = true;
        }
        ChainState* st = nullptr;
        for (ChainState& c : g_chain)
            if (c.dev == dev && c.stream == stream) st = &c;
        if (!st) {
            g_chain.push_back(ChainState{dev, stream, nullptr, 0, 8u, 0u});
            st = &g_chain.back();
        }
        if (st->cap < ntiles) {   // first use / larger geometry: a fresh zeroed buffer (the old one may still be read by queued launches: kept)
            const size_t cap = ntiles < 4096 ? 4096 : ntiles * 2;
            unsigned* nb = nullptr;
            // zeroed ON THE LAUNCHING STREAM: hipMemset runs on the null stream, which a non-blocking stream does not wait for
            if (hipMalloc((void**)&nb, (cap + 16) * sizeof(unsigned)) != hipSuccess || hipMemsetAsync(nb, 0, (cap + 16) * sizeof(unsigned), stream) != hipSuccess)
                return fail(RESR_ERR_LAUNCH, "conv3x3_chain: flag buffer");
            st->buf = nb; st->cap = cap; st->epoch = 8u; st->ticket_base = 0u;
        }
        if (st->epoch > 0x70000000u) {   // far from wrapping the signed comparison: start over behind everything queued
            if (hipMemsetAsync(st->buf, 0, st->cap * sizeof(unsigned), stream) != hipSuccess) return fail(RESR_ERR_LAUNCH, "conv3x3_chain: flag reset");
            st->epoch = 8u;
        }
        cj.flags = st->buf;
        cj.errors = st->buf + st->cap;
        cj.tickets = st->buf + st->cap + 8;
        cj.epoch = st->epoch;
        st->epoch += 8u;
        cj.njobs = njobs;
        memcpy(cj.split, split, sizeof(split));
        double f = 0, b = 0;
        for (int j = 0; j < njobs; ++j) { cj.job[j] = jobs[j]; f += flop[j]; b += bytes[j]; }
        for (int j = njobs; j < kMaxChain; ++j) cj.job[j] = jobs[njobs - 1];
        const int kind = (a.flags & RESR_CONV_MASK_BITS) ? 2 : (a.flags & RESR_CONV_WRITE_SIGNBITS) ? 1 : 0;
        return conv3x3_ws_chain_launch(a, cj, rows, kind, &st->ticket_base, f, b, stream);   // under the lock: the ticket base follows launch order
    }
}

int conv3x3_ws_f16(const ConvArgs& a, int mt, bool x2, hipStream_t stream) {
    static const int rows1[] = {16, 8}, rows2[] = {16, 8};
    // 4x4 / stride-2 convolutions as sparse-tap 3x3 convolutions over the space-to-depth image: plain epilogue, 64-channel groups
    if (!x2 && mt == 2 && (a.s2d_c > 0 || a.tap_c > 0) && !a.res0 && !a.res1 && !a.mask && !a.aux &&
        !(a.flags & ~(RESR_CONV_LRELU | RESR_CONV_NO_BIAS)) && a.cin0 == a.cin)
        return conv3x3_ws_sparse(a, 8, a.s2d_c > 0 ? 1 : 2, stream);   // 8-row tiles: the 16-row shape of this variant spills
    if (x2) {
        if (mt == 1) return conv3x3_ws_x2_mt1(a, pick_rows(a, rows1, 2), stream);
        return conv3x3_ws_x2_mt2(a, pick_rows(a, rows2, 2), stream);
    }
    if (mt == 1) return conv3x3_ws_mt1(a, pick_rows(a, rows1, 2), stream);
    return conv3x3_ws_mt2(a, pick_rows(a, rows2, 2), stream);
}

}  // namespace resr
