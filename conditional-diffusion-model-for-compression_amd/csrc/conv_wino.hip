// Instantiations of the Winograd F(2x2,3x3) convolution.  variant 0 = the shipped one (8-wave form); the other ids
// are in-process A/B and timing-ablation variants for tools/conv_bench.py (reached through cdx_conv_f32_tile).
#include "conv_wino.h"
namespace cdx {
int conv_dispatch_wino(int variant, const ConvParams& p, hipStream_t stream) {
    switch (variant) {
        case 0: return conv_wino8_launch<WinoCfg<2, 0>>(p, stream);     // shipped: 8 waves, two per SIMD
        case 31: return conv_wino_launch<WinoCfg<2, 0>>(p, stream);     // 4-wave form (one 512-register wave per SIMD)
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
        case 44: return conv_wino8_launch<WinoCfg<2, 0>>(p, stream);   // 8-wave form (two waves per SIMD), experimental
        case 45: return conv_wino8_launch<WinoCfg<4, 0>>(p, stream);
        case 46: return conv_wino8_launch<WinoCfg<2, 1>>(p, stream);    // 8-wave ablations: no weight refills
        case 47: return conv_wino8_launch<WinoCfg<2, 2>>(p, stream);    //   one patch half-load per chunk
        case 48: return conv_wino8_launch<WinoCfg<2, 4>>(p, stream);    //   no staging
        case 49: return conv_wino8_launch<WinoCfg<2, 8>>(p, stream);    //   no transform adds
        case 50: return conv_wino8_launch<WinoCfg<2, 15>>(p, stream);   //   all of the above
        case 52: return conv_wino8_launch<WinoCfg<2, 16>>(p, stream);   //   no epilogue
        case 53: return conv_wino8_launch<WinoCfg<2, 256>>(p, stream);  //   no residual loads
        case 54: return conv_wino8_launch<WinoCfg<2, 15 + 16 + 256>>(p, stream);   //   MFMA stream + prologue staging only
        case 51: return conv_wino8_launch<WinoCfg<2, 32>>(p, stream);   // transform adds pinned as v_pk_add_f32 (inline asm)
        case 34: return conv_wino_launch<WinoCfg<2, 64>>(p, stream);
        default: return CDX_ENOTSUP;
    }
}
}  // namespace cdx
