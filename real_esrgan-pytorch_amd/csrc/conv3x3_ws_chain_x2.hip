// Chained dense-block passes, exact16 (RESR_F16X2, hi/lo pairs) instantiations (launchers: conv3x3_ws_chain.h).
#include "conv3x3_ws_chain.h"

namespace resr {

int conv3x3_ws_chain_launch_x2(const ConvArgs& a, const ChainArgs& cj, int tile_rows, int kind, double flop, double bytes, hipStream_t stream) {
    return chain_launch_t<1>(a, cj, tile_rows, kind, flop, bytes, stream);
}

}  // namespace resr
