// Sparse-tap instantiations of the producer/consumer convolution (conv3x3_ws.h, SP): the discriminator's 4x4 / stride-2
// convolutions (model.py:140-152) and their backward-data passes as 3x3 convolutions over the space-to-depth image that
// skip the virtual kernel's zero taps.  cout groups of 64 (two M tiles), plain bias-less epilogue (+ LeakyReLU flag); fast
// mode (f16) and exact16 (hi/lo pairs).
#include "conv3x3_ws.h"

namespace resr {

int conv3x3_ws_sparse(const ConvArgs& a, int tile_rows, int sp, bool x2, hipStream_t stream) {
    (void)tile_rows;   // 8-row tiles only: the 16-row shape of this variant exceeds the register budget (it spilled)
    if (x2) {
        if (sp == 1) return launch_ws_epi<half_t, 2, 2, 4, 0, true, 1>(a, stream);
        return launch_ws_epi<half_t, 2, 2, 4, 0, true, 2>(a, stream);
    }
    if (sp == 1) return launch_ws_epi<half_t, 2, 2, 4, 0, false, 1>(a, stream);
    return launch_ws_epi<half_t, 2, 2, 4, 0, false, 2>(a, stream);
}

}  // namespace resr
