// Instantiations of the Winograd F(2x2,3x3) convolution.  variant 0 = the shipped one; ids >= 32 are
// in-process A/B variants for tools/conv_bench.py (ring depth).
#include "conv_wino.h"
namespace cdx {
int conv_dispatch_wino(int variant, const ConvParams& p, hipStream_t stream) {
    switch (variant) {
        case 0: return conv_wino_launch<WinoCfg<2, 0>>(p, stream);
        case 32: return conv_wino_launch<WinoCfg<1, 0>>(p, stream);
        case 33: return conv_wino_launch<WinoCfg<4, 0>>(p, stream);
        case 34: return conv_wino_launch<WinoCfg<2, 1>>(p, stream);
        case 35: return conv_wino_launch<WinoCfg<2, 2>>(p, stream);
        case 36: return conv_wino_launch<WinoCfg<2, 4>>(p, stream);
        case 37: return conv_wino_launch<WinoCfg<2, 7>>(p, stream);
        case 38: return conv_wino_launch<WinoCfg<2, 8>>(p, stream);
        case 39: return conv_wino_launch<WinoCfg<2, 15>>(p, stream);
        case 40: return conv_wino_launch<WinoCfg<2, 16>>(p, stream);
        case 41: return conv_wino_launch<WinoCfg<2, 31>>(p, stream);
        default: return CDX_ENOTSUP;
    }
}
}  // namespace cdx
