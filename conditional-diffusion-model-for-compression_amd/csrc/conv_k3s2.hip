// Instantiations of the implicit-GEMM convolution: ksize=3 stride=2.
// CDX_CONV_CASE(KS, ST, log2(TW), shape id, WM, WN, MT[, WK, PF])
#include "conv_kernel.h"
namespace cdx {
#define CDX_CONV_CASES(KS, ST) \
    CDX_CONV_CASE(KS, ST, 2, 3, 1, 4, 2) \
    CDX_CONV_CASE(KS, ST, 2, 4, 2, 2, 1) \
    CDX_CONV_CASE(KS, ST, 2, 5, 1, 1, 1, 4, 3) \
    CDX_CONV_CASE(KS, ST, 3, 3, 1, 4, 2) \
    CDX_CONV_CASE(KS, ST, 3, 4, 2, 2, 1) \
    CDX_CONV_CASE(KS, ST, 3, 5, 1, 1, 1, 4, 3) \
    CDX_CONV_CASE(KS, ST, 4, 3, 1, 4, 2) \
    CDX_CONV_CASE(KS, ST, 4, 4, 2, 2, 1) \
    CDX_CONV_CASE(KS, ST, 4, 5, 1, 1, 1, 4, 3) \
    CDX_CONV_CASE(KS, ST, 5, 3, 1, 4, 2) \
    CDX_CONV_CASE(KS, ST, 5, 4, 2, 2, 1) \
    CDX_CONV_CASE(KS, ST, 5, 5, 1, 1, 1, 4, 3) \

int conv_dispatch_k3s2(int logtw, int wcfg, const ConvParams& p, hipStream_t stream) {
    CDX_CONV_DISPATCH_BODY(3, 2)
}
}  // namespace cdx
