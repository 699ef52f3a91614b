// Instantiations of the Winograd F(2x2,3x3) convolution.  variant 0 = the shipped kernel; the other ids are timing
// ablations for tools/conv_bench.py and exist only in the tuning build (make EXPERIMENTS=1 -> libcdx_tune.so).
#include "conv_wino.h"
namespace cdx {
int conv_dispatch_wino(int variant, const ConvParams& p, hipStream_t stream) {
    switch (variant) {
        case 0: return conv_wino8_launch<WinoCfg<2, 0>>(p, stream);     // shipped: 8 waves, two per SIMD
#ifdef CDX_TUNING
        case 45: return conv_wino8_launch<WinoCfg<4, 0>>(p, stream);    // deeper weight ring
        case 46: return conv_wino8_launch<WinoCfg<2, 1>>(p, stream);    // ablations: no weight refills
        case 47: return conv_wino8_launch<WinoCfg<2, 2>>(p, stream);    //   one patch half-load per chunk
        case 48: return conv_wino8_launch<WinoCfg<2, 4>>(p, stream);    //   no staging
        case 49: return conv_wino8_launch<WinoCfg<2, 8>>(p, stream);    //   no transform adds
        case 50: return conv_wino8_launch<WinoCfg<2, 15>>(p, stream);   //   all of the above
        case 52: return conv_wino8_launch<WinoCfg<2, 16>>(p, stream);   //   no epilogue
        case 53: return conv_wino8_launch<WinoCfg<2, 256>>(p, stream);  //   no residual loads
        case 54: return conv_wino8_launch<WinoCfg<2, 15 + 16 + 256>>(p, stream);   //   MFMA stream + prologue staging only
#endif
        default: return CDX_ENOTSUP;
    }
}
}  // namespace cdx
