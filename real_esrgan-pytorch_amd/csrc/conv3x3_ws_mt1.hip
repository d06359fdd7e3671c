// cout <= 32 instantiations of the producer/consumer convolution: 8 consumer waves x {4, 2, 1} rows (tiles of 32/16/8 rows).
#include "conv3x3_ws.h"

namespace resr {

int conv3x3_ws_mt1(const ConvArgs& a, int tile_rows, hipStream_t stream) {
    if (tile_rows >= 32) return launch_ws<half_t, 1, 4, 8>(a, stream);
    if (tile_rows >= 16) return launch_ws<half_t, 1, 2, 8>(a, stream);
    return launch_ws<half_t, 1, 1, 8>(a, stream);
}

}  // namespace resr
