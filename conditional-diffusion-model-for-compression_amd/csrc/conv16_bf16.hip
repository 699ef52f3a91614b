// bfloat16 instantiations of the 16-bit convolution kernel (conv16_kernel.h, BF = 1: v_mfma_f32_32x32x16_bf16), reached
// through cdx_conv_f16 with CDX_CONV_BF16.  A translation unit of its own (build time).
#include "conv16_kernel.h"
#include "conv_kpar_kernel.h"

namespace cdx {
int conv16_dispatch_bf16(int ks, int stride, int logtw, int mt, const Conv16Params& p, hipStream_t stream) {
    if (ks == 3 && stride == 1 && logtw == 4 && mt == 4) return conv16_ws_launch<Conv16Cfg<3, 1, 4, 4, 3, 0, 0, 1, 1, 1>>(p, stream);      // wave-specialised 8 x 16 tile (conv16.hip)
#define C16(KS, ST, LT, MT) if (ks == KS && stride == ST && logtw == LT) return conv16_launch<Conv16Cfg<KS, ST, LT, MT, 3, 0, 0, 1, 1>>(p, stream);
    C16(3, 1, 2, 4) C16(3, 1, 3, 2) C16(3, 1, 4, 2)
    C16(1, 1, 2, 4) C16(1, 1, 3, 2) C16(1, 1, 4, 2) C16(1, 1, 5, 4)
    C16(3, 2, 2, 2) C16(3, 2, 3, 2) C16(3, 2, 4, 2) C16(3, 2, 5, 2)
#undef C16
    return CDX_ENOTSUP;
}
int conv16_kpar_dispatch_bf16(int ks, const Conv16Params& p, hipStream_t stream) {
    if (ks == 3) return conv_kpar_launch<KparCfg<3, 3, 0, 1>>(p, stream);
    return conv_kpar_launch<KparCfg<1, 3, 0, 1>>(p, stream);
}
}  // namespace cdx
