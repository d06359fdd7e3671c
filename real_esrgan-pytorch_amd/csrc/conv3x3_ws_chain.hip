// Chained dense-block passes, fast mode (f16) instantiations (launchers: conv3x3_ws_chain.h; kernel: conv3x3_ws.h, CH).
#include "conv3x3_ws_chain.h"

namespace resr {

int conv3x3_ws_chain_launch_x2(const ConvArgs& a, const ChainArgs& cj, int tile_rows, int kind, double flop, double bytes, hipStream_t stream);   // conv3x3_ws_chain_x2.hip
int conv3x3_ws_chain_launch_mx(const ConvArgs& a, const ChainArgs& cj, int tile_rows, int kind, double flop, double bytes, hipStream_t stream);   // conv3x3_ws_chain_mx.hip

int conv3x3_ws_chain_launch(const ConvArgs& a, const ChainArgs& cj, int tile_rows, int kind, bool x2, double flop, double bytes, hipStream_t stream) {
    if (x2 && a.mx) return conv3x3_ws_chain_launch_mx(a, cj, tile_rows, kind, flop, bytes, stream);
    if (x2) return conv3x3_ws_chain_launch_x2(a, cj, tile_rows, kind, flop, bytes, stream);
    return chain_launch_t<0>(a, cj, tile_rows, kind, flop, bytes, stream);
}

}  // namespace resr
