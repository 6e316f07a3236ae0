// One (compute, input, output) type combination of the implicit-GEMM conv kernel (see conv_igemm_impl.h);
// split per combination so the ~20 tile/BK/buffering instantiations of each compile in parallel.
#include "conv_igemm_impl.h"

hipError_t launch_conv_f16_fh(const convimpl::ConvP& p, const ftc_op& o, hipStream_t s) {
    return convimpl::launch_types<_Float16, float, _Float16>(p, o, s);
}
