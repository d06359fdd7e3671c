// Chained dense-block passes of an exact16 INFERENCE forward with MX stages on the pair chunks (RESR_CONV_MX_PAIRS; kernel:
// conv3x3_ws.h, X2 = 2; launchers: conv3x3_ws_chain.h).
#include "conv3x3_ws_chain.h"

namespace resr {

int conv3x3_ws_chain_launch_mx(const ConvArgs& a, const ChainArgs& cj, int tile_rows, int kind, double flop, double bytes, hipStream_t stream) {
    return chain_launch_t<2>(a, cj, tile_rows, kind, flop, bytes, stream);
}

}  // namespace resr
