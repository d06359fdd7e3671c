// RESR_F16X2 ("exact16") instantiations of the producer/consumer convolution, cout <= 32: the same tile shapes as the
// plain f16 ones (conv3x3_ws_mt1.hip); three stages per real chunk and the hi/lo epilogue (see conv3x3_ws.h).
#include "conv3x3_ws.h"

namespace resr {

int conv3x3_ws_x2_mt1(const ConvArgs& a, int tile_rows, hipStream_t stream) {
    if (tile_rows >= 16) return launch_ws<half_t, 1, 2, 8, 1>(a, stream);
    return launch_ws<half_t, 1, 1, 8, 1>(a, stream);
}

}  // namespace resr
