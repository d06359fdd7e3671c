// RESR_CONV_MX_PAIRS instantiations of the producer/consumer convolution (exact16 with one f16 stage + one MX-fp8 stage per pair
// chunk; kernel: conv3x3_ws.h, X2 = 2): the inference epilogues of both output widths, same tile shapes as the plain exact16 ones.
#include "conv3x3_ws.h"

namespace resr {

int conv3x3_ws_mx_mt1(const ConvArgs& a, int tile_rows, hipStream_t stream) {
    if (tile_rows >= 16) return launch_ws<half_t, 1, 2, 8, 2>(a, stream);
    return launch_ws<half_t, 1, 1, 8, 2>(a, stream);
}

int conv3x3_ws_mx_mt2(const ConvArgs& a, int tile_rows, hipStream_t stream) {
    if (tile_rows >= 16) return launch_ws<half_t, 2, 4, 4, 2>(a, stream);
    return launch_ws<half_t, 2, 2, 4, 2>(a, stream);
}

}  // namespace resr
