// cout 64 instantiations of the producer/consumer convolution: 4 consumer waves x {4, 2} rows (tiles of 16/8 rows).
// (8 waves x 2 rows ties at 16 rows; 3 rows per wave -- 24-row tiles -- measured slower: 151 vs 123 us for 192->64.)
#include "conv3x3_ws.h"

namespace resr {

int conv3x3_ws_mt2(const ConvArgs& a, int tile_rows, hipStream_t stream) {
    if (tile_rows >= 16) return launch_ws<half_t, 2, 4, 4>(a, stream);
    return launch_ws<half_t, 2, 2, 4>(a, stream);
}

}  // namespace resr
