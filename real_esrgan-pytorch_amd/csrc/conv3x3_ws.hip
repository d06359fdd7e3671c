// conv3x3_ws.hip -- entry of the fast-mode (f16) producer/consumer convolution (kernel: conv3x3_ws.h; instantiations:
// conv3x3_ws_mt1.hip for cout <= 32, conv3x3_ws_mt2.hip for cout 64 -- two translation units so they build in parallel).
#include <stdlib.h>

#include <mutex>

#include "conv3x3.h"

namespace resr {

unsigned long long* g_conv_trace = nullptr;
void conv_trace_set(void* p) { g_conv_trace = (unsigned long long*)p; }

int conv3x3_ws_mt1(const ConvArgs& a, int tile_rows, hipStream_t stream);
int conv3x3_ws_mt2(const ConvArgs& a, int tile_rows, hipStream_t stream);
int conv3x3_ws_x2_mt1(const ConvArgs& a, int tile_rows, hipStream_t stream);   // RESR_F16X2 instantiations
int conv3x3_ws_x2_mt2(const ConvArgs& a, int tile_rows, hipStream_t stream);
int conv3x3_ws_mx_mt1(const ConvArgs& a, int tile_rows, hipStream_t stream);   // RESR_CONV_MX_PAIRS instantiations (conv3x3_ws_mx.hip)
int conv3x3_ws_mx_mt2(const ConvArgs& a, int tile_rows, hipStream_t stream);

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

int conv3x3_ws_sparse(const ConvArgs& a, int tile_rows, int sp, bool x2, hipStream_t stream);   // conv3x3_ws_sp.hip
int conv3x3_ws_chain_launch(const ConvArgs& a, const ChainArgs& cj, int tile_rows, int kind, bool x2, double flop, double bytes, hipStream_t stream);   // conv3x3_ws_chain.hip

// ---- chained dense-block passes (conv3x3_ws.h, CH) ----
// All device-side state (epoch, tickets, flags, error counters) lives in memory the CALLER hands in (conv3x3.h, ChainArgs):
// nothing is allocated here and nothing is synchronised.  What the host keeps per device is (a) which stream launched the
// last chain and an event behind it -- a chained launch waits for its own workgroups from inside, so two of them side by
// side on different streams could each hold CUs the other's missing workgroups need: a launch from another stream first
// WAITS (stream-side, hipStreamWaitEvent) for the previous owner's last chain -- and (b) two host-mapped error words the
// kernels add to when a flag poll gives up or an XCD receives more than its share, so the host can look without a sync.
namespace {
struct ChainHost {
    bool init = false, owned = false;
    hipStream_t owner = nullptr;
    hipEvent_t ev = nullptr;
    unsigned* err_host = nullptr;   // [2], host-mapped
    unsigned* err_dev = nullptr;    // device view of the same words
    const char* nan16 = nullptr;
};
std::mutex g_chain_mu;
ChainHost g_chain[kMaxDevices];
__device__ uint4 g_conv_nan16 = {0x7e007e00u, 0x7e007e00u, 0x7e007e00u, 0x7e007e00u};   // eight f16 quiet NaNs
}  // namespace

// (time-outs | beyond-share << 32) seen on the current device so far; reads host memory only -- no synchronisation, so it
// reflects the launches that have finished by now
long long conv3x3_chain_errors() {
    std::lock_guard<std::mutex> lk(g_chain_mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return -1;
    const ChainHost& c = g_chain[dev];
    if (!c.init) return 0;
    const volatile unsigned* e = c.err_host;
    return (long long)e[0] + ((long long)e[1] << 32);
}

size_t conv3x3_chain_state_bytes(int n, int h, int w) {
    if (n <= 0 || h <= 0 || w <= 0) return 0;
    const size_t tiles = (size_t)((w + 31) / 32) * ((h + 7) / 8) * n;   // the smallest tile shape
    return align_up((kChainHdr + tiles) * sizeof(unsigned), 256);
}

// rc: RESR_OK launched; 1 = cannot chain here (state too small / misaligned): the caller runs one launch per job
int conv3x3_ws_chain_f16(const ConvArgs& a, const ChainJob* jobs, int njobs, const double* flop, const double* bytes, bool x2,
                         void* state, size_t state_bytes, hipStream_t stream) {
    static const int rows1[] = {16, 8};
    int rows = pick_rows(a, rows1, 2);
    // Experiment (RESR_CHAIN_PIPE="5,7,9,11"): a pinned pipeline instead of a walk -- 8-row tiles, the 32 workgroups of an XCD split
    // over the four jobs as given, each workgroup running ONE job over the XCD's bands top to bottom behind its predecessor's
    // flags, so that the planes a block's passes exchange stay inside the XCD's L2
    int split[kMaxChain + 2] = {0};
    const char* pipe_env = getenv("RESR_CHAIN_PIPE");   // read per call (a test flips it)
    if (pipe_env && njobs == 4 && !x2) {
        int c[4] = {0, 0, 0, 0};
        if (sscanf(pipe_env, "%d,%d,%d,%d", &c[0], &c[1], &c[2], &c[3]) == 4 && c[0] > 0 && c[1] > 0 && c[2] > 0 && c[3] > 0 &&
            c[0] + c[1] + c[2] + c[3] == 32 && (size_t)((a.w_ + 31) / 32) * ((a.h + 7) / 8) * a.n >= 256 * 8) {
            rows = 8;
            for (int j = 0; j < 4; ++j) split[j + 1] = split[j] + c[j];
            for (int j = 5; j < kMaxChain + 2; ++j) split[j] = 32;
        }
    }
    if (x2)   // exact16 with the closing convolution's halves runs on 8-row tiles only (conv3x3_ws_chain.h)
        for (int j = 0; j < njobs; ++j)
            if (jobs[j].kind == 3) rows = 8;
    const size_t ntiles = (size_t)((a.w_ + 31) / 32) * ((a.h + rows - 1) / rows) * a.n;
    if (!state || ((size_t)state & 15) || state_bytes < (kChainHdr + ntiles) * sizeof(unsigned)) return 1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return fail(RESR_ERR_LAUNCH, "conv3x3_chain: hipGetDevice");
    ChainArgs cj;
    memset(&cj, 0, sizeof(cj));
    {
        std::lock_guard<std::mutex> lk(g_chain_mu);
        ChainHost& c = g_chain[dev];
        if (!c.init) {   // once per device: an event, two host-mapped words, the address of the NaN page
            void* hp = nullptr; void* dp = nullptr; void* np = nullptr;
            if (hipEventCreateWithFlags(&c.ev, hipEventDisableTiming) != hipSuccess ||
                hipHostMalloc(&hp, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess ||
                hipGetSymbolAddress(&np, HIP_SYMBOL(g_conv_nan16)) != hipSuccess || !np)
                return fail(RESR_ERR_LAUNCH, "conv3x3_chain: per-device set-up failed");
            memset(hp, 0, 64);
            c.err_host = (unsigned*)hp; c.err_dev = (unsigned*)dp; c.nan16 = (const char*)np;
            c.init = true;
        }
        // Under stream capture nothing runs now: the launch becomes a node of a graph that is replayed later, possibly many times,
        // and neither a wait on the owner's event nor a record of it would mean anything at replay time (a captured record turns
        // the event into a graph-internal dependency that is never "really" recorded; an eager chain waiting on it afterwards is
        // an error or a no-op depending on the runtime).  So a capturing stream neither waits, nor records, nor takes ownership:
        // inside ONE graph the chain nodes are ordered by the capture itself; what the caller must guarantee (include/resr.h) is
        // that a graph holding chained launches is not replayed while ANOTHER stream runs chained launches on the same device.
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cap) != hipSuccess) cap = hipStreamCaptureStatusNone;
        const bool capturing = cap == hipStreamCaptureStatusActive;
        if (!capturing) {
            if (c.owned && c.owner != stream && hipStreamWaitEvent(stream, c.ev, 0) != hipSuccess)
                return fail(RESR_ERR_LAUNCH, "conv3x3_chain: hipStreamWaitEvent");
            c.owner = stream;
            c.owned = true;
        }
        cj.state = (unsigned*)state;
        cj.cap = (unsigned)(state_bytes / sizeof(unsigned) - kChainHdr);
        cj.host_errors = c.err_dev;
        cj.nan16 = c.nan16;
        cj.njobs = njobs;
        memcpy(cj.split, split, sizeof(split));
        double f = 0, b = 0;
        for (int j = 0; j < njobs; ++j) { cj.job[j] = jobs[j]; f += flop[j]; b += bytes[j]; }
        for (int j = njobs; j < kMaxChain; ++j) cj.job[j] = jobs[njobs - 1];
        const int kind = (a.flags & RESR_CONV_MASK_BITS) ? 2 : (a.flags & RESR_CONV_WRITE_SIGNBITS) ? 1 : 0;
        const int rc = conv3x3_ws_chain_launch(a, cj, rows, kind, x2, f, b, stream);   // under the lock: owner and event follow launch order
        if (rc == RESR_OK && !capturing && hipEventRecord(c.ev, stream) != hipSuccess) return fail(RESR_ERR_LAUNCH, "conv3x3_chain: hipEventRecord");
        return rc;
    }
}

int conv3x3_ws_f16(const ConvArgs& a, int mt, bool x2, hipStream_t stream) {
    static const int rows1[] = {16, 8}, rows2[] = {16, 8};
    // 4x4 / stride-2 convolutions as sparse-tap 3x3 convolutions over the space-to-depth image: plain epilogue, 64-channel groups
    if (mt == 2 && (a.s2d_c > 0 || a.tap_c > 0) && !a.res0 && !a.res1 && !a.mask && !a.aux &&
        !(a.flags & ~(RESR_CONV_LRELU | RESR_CONV_NO_BIAS)) && a.cin0 == a.cin && (!(a.flags & RESR_CONV_LRELU) || (a.slope >= 0.f && a.slope <= 1.f)))
        return conv3x3_ws_sparse(a, 8, a.s2d_c > 0 ? 1 : 2, x2, stream);   // 8-row tiles: the 16-row shape of this variant spills
    if (x2 && a.mx) {
        if (mt == 1) return conv3x3_ws_mx_mt1(a, pick_rows(a, rows1, 2), stream);
        return conv3x3_ws_mx_mt2(a, pick_rows(a, rows2, 2), stream);
    }
    if (x2) {
        if (mt == 1) return conv3x3_ws_x2_mt1(a, pick_rows(a, rows1, 2), stream);
        return conv3x3_ws_x2_mt2(a, pick_rows(a, rows2, 2), stream);
    }
    if (mt == 1) return conv3x3_ws_mt1(a, pick_rows(a, rows1, 2), stream);
    return conv3x3_ws_mt2(a, pick_rows(a, rows2, 2), stream);
}

}  // namespace resr
