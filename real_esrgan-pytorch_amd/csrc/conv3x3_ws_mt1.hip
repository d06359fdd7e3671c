// cout <= 32 instantiations of the producer/consumer convolution: 8 consumer waves x {2, 1} rows (tiles of 16 / 8 rows).
// (Measured on the dense-block passes at B = 16, 256^2: 16 rows with three halo buffers 852 TFLOP/s, 24 rows x two buffers
// 795, 32 rows x two buffers with per-wave weight streams 775, 4 waves x 4 rows 765, 8 rows with all weights resident 783.
// Re-measured with LDS weights and the lean epilogue, same box, unchained: 4 waves x 4 rows -- 0.75 instead of 1.17 LDS
// fragment reads per MFMA -- 797-808 TFLOP/s against 823-854 for 8 x 2: one consumer wave per SIMD does not cover its own
// LDS latency.)
#include "conv3x3_ws.h"

namespace resr {

int conv3x3_ws_mt1(const ConvArgs& a, int tile_rows, hipStream_t stream) {
    if (tile_rows >= 16) return launch_ws<half_t, 1, 2, 8>(a, stream);
    return launch_ws<half_t, 1, 1, 8>(a, stream);
}

}  // namespace resr
