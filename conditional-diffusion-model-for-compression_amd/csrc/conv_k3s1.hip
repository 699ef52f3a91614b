// Instantiations of the implicit-GEMM convolution: ksize=3 stride=1.
// CDX_CONV_CASE(KS, ST, log2(TW), shape id, WM, WN, MT[, WK, PF])
#include "conv_kernel.h"
namespace cdx {
#define CDX_CONV_CASES(KS, ST) \
    CDX_CONV_CASE(KS, ST, 2, 0, 1, 4, 4, 1, 4, OPT_OCC2) \
    CDX_CONV_CASE(KS, ST, 2, 1, 2, 2, 2) \
    CDX_CONV_CASE(KS, ST, 2, 2, 4, 1, 1) \
    CDX_CONV_CASE(KS, ST, 2, 5, 1, 1, 1, 4, 3) \
    CDX_CONV_CASE(KS, ST, 2, 6, 1, 1, 2, 4, 3) \
    CDX_CONV_CASE(KS, ST, 3, 0, 1, 4, 4, 1, 4, OPT_OCC2) \
    CDX_CONV_CASE(KS, ST, 3, 1, 2, 2, 2) \
    CDX_CONV_CASE(KS, ST, 3, 2, 4, 1, 1) \
    CDX_CONV_CASE(KS, ST, 3, 5, 1, 1, 1, 4, 3) \
    CDX_CONV_CASE(KS, ST, 3, 6, 1, 1, 2, 4, 3) \
    CDX_CONV_CASE(KS, ST, 4, 0, 1, 4, 4, 1, 4, OPT_OCC2) \
    CDX_CONV_CASE(KS, ST, 4, 1, 2, 2, 2) \
    CDX_CONV_CASE(KS, ST, 4, 2, 4, 1, 1) \
    CDX_CONV_CASE(KS, ST, 4, 5, 1, 1, 1, 4, 3) \
    CDX_CONV_CASE(KS, ST, 4, 6, 1, 1, 2, 4, 3) \
    CDX_CONV_CASE(KS, ST, 5, 0, 1, 4, 4, 1, 4, OPT_OCC2) \
    CDX_CONV_CASE(KS, ST, 5, 1, 2, 2, 2) \
    CDX_CONV_CASE(KS, ST, 5, 2, 4, 1, 1) \
    CDX_CONV_CASE(KS, ST, 5, 5, 1, 1, 1, 4, 3) \
    CDX_CONV_CASE(KS, ST, 5, 6, 1, 1, 2, 4, 3) \
    CDX_CONV_CASE(KS, ST, 5, 10, 1, 4, 4, 1, 3, OPT_OCC2 | OPT_CIN8) \

int conv_dispatch_k3s1(int logtw, int wcfg, const ConvParams& p, hipStream_t stream) {
    CDX_CONV_DISPATCH_BODY(3, 1)
}
}  // namespace cdx
