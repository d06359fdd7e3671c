// RESR_F16X2 ("exact16") instantiations of the producer/consumer convolution, cout 64 (see conv3x3_ws_mt2.hip).
#include "conv3x3_ws.h"

namespace resr {

int conv3x3_ws_x2_mt2(const ConvArgs& a, int tile_rows, hipStream_t stream) {
    if (tile_rows >= 16) return launch_ws<half_t, 2, 4, 4, 1>(a, stream);
    return launch_ws<half_t, 2, 2, 4, 1>(a, stream);
}

}  // namespace resr
