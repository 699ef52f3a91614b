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
        case 38: return conv_wino_launch<WinoCfg<2, 16>>(p, stream);   // ablation: no epilogue
        case 39: return conv_wino_launch<WinoCfg<2, 285>>(p, stream);  // ablation: MFMA stream only (1|4|8|16|256)
        case 40: return conv_wino_launch<WinoCfg<2, 13>>(p, stream);   // ablation: loop stripped (1|4|8), prologue + epilogue kept
        case 41: return conv_wino_launch<WinoCfg<2, 256>>(p, stream);  // ablation: no residual loads
        case 42: return conv_wino_launch<WinoCfg<2, 272>>(p, stream);  // ablation: no residual loads, no epilogue
        case 34: return conv_wino_launch<WinoCfg<2, 64>>(p, stream);
        default: return CDX_ENOTSUP;
    }
}
}  // namespace cdx
