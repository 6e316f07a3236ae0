// The fp16x3 combination of the implicit-GEMM conv kernel (FTC_FLAG_SPLIT16, see split16() in conv_igemm_impl.h): fp32 tensors and
// weights, every product as three fp16 MFMAs of hi / lo split operands.  Its own compute-type tag = its own instantiations.
#include "conv_igemm_impl.h"

hipError_t launch_conv_x3(const convimpl::ConvP& p, const ftc_op& o, hipStream_t s) {
    return convimpl::launch_types<convimpl::x3f32, float, float>(p, o, s);
}
