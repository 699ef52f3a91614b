// Experimental instantiations of the dominant shape (ksize 3, stride 1, TW = 32, 128 x 128 tile) for in-process
// A/B runs (tools/conv_bench.py --tiles 0,16,17,...).  Not used by cdx_conv_f32's own selection.
// id = 16 + index; columns: PF (prefetch ring depth), OPT bits (conv_kernel.h).
#include "conv_kernel.h"
namespace cdx {
#define EXP_CASE(ID, PF, OPT) case ID: return conv_launch<ConvCfg<3, 1, 5, 1, 4, 4, 1, PF, OPT>>(p, stream);
int conv_dispatch_exp(int logtw, int wcfg, const ConvParams& p, hipStream_t stream) {
    if (logtw != 5) return CDX_ENOTSUP;
    switch (wcfg) {
        EXP_CASE(16, 4, OPT_OCC2 | OPT_ABL_NO_STAGE)
        EXP_CASE(17, 4, OPT_OCC2 | OPT_ABL_NO_EPILOGUE)
        EXP_CASE(18, 4, OPT_OCC2 | OPT_ABL_NO_STAGE | OPT_ABL_NO_EPILOGUE)
        EXP_CASE(19, 4, OPT_OCC2 | OPT_ABL_ONE_WG)
        EXP_CASE(20, 4, OPT_OCC2 | OPT_ABL_ONE_WG | OPT_ABL_NO_STAGE | OPT_ABL_NO_EPILOGUE)
        EXP_CASE(21, 4, 0)
        default: return CDX_ENOTSUP;
    }
}
}  // namespace cdx
