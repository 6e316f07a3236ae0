// Dispatcher of the (fp16 compute, fp16 input, _Float16 output) combination: its four parts are separate translation units
// (conv_igemm_part.hip compiled with different -D flags, see build.py).
#include "conv_igemm_impl.h"

hipError_t launch_conv_f16_hh_p0(const convimpl::ConvP& p, const ftc_op& o, hipStream_t s);
hipError_t launch_conv_f16_hh_p1(const convimpl::ConvP& p, const ftc_op& o, hipStream_t s);
hipError_t launch_conv_f16_hh_p2(const convimpl::ConvP& p, const ftc_op& o, hipStream_t s);
hipError_t launch_conv_f16_hh_p3(const convimpl::ConvP& p, const ftc_op& o, hipStream_t s);

hipError_t launch_conv_f16_hh(const convimpl::ConvP& p, const ftc_op& o, hipStream_t s) {
    switch (convimpl::conv_part<_Float16, _Float16>(o)) {
    case convimpl::PART_HALO: return launch_conv_f16_hh_p0(p, o, s);
    case convimpl::PART_BK32: return launch_conv_f16_hh_p1(p, o, s);
    case convimpl::PART_BK64: return launch_conv_f16_hh_p2(p, o, s);
    default: return launch_conv_f16_hh_p3(p, o, s);
    }
}
