// Instantiations of the Winograd F(2x2,3x3) convolution.  variant 0 = the shipped one; ids >= 32 are
// in-process A/B variants for tools/conv_bench.py (ring depth).
#include "conv_wino.h"
namespace cdx {
int conv_dispatch_wino(int variant, const ConvParams& p, hipStream_t stream) {
    switch (variant) {
        case 0: return conv_wino_launch<WinoCfg<2, 0>>(p, stream);
        case 32: return conv_wino_launch<WinoCfg<1, 0>>(p, stream);
        case 33: return conv_wino_launch<WinoCfg<4, 0>>(p, stream);
        case 35: return conv_wino_persist_launch<WinoCfg<2, 0>>(p, stream);
        case 37: return conv_wino_persist_launch<WinoCfg<2, 128>>(p, stream);
        case 36: return conv_wino_persist_launch<WinoCfg<4, 0>>(p, stream);
        case 34: return conv_wino_launch<WinoCfg<2, 64>>(p, stream);
        default: return CDX_ENOTSUP;
    }
}
}  // namespace cdx
