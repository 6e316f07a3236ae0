// Dispatcher of the (bf16 compute, bf16 input, __bf16 output) combination: its four parts are separate translation units
// (conv_igemm_part.hip compiled with different -D flags, see build.py).
#include "conv_igemm_impl.h"

hipError_t launch_conv_bf16_bb_p0(const convimpl::ConvP& p, const ftc_op& o, hipStream_t s);
hipError_t launch_conv_bf16_bb_p1(const convimpl::ConvP& p, const ftc_op& o, hipStream_t s);
hipError_t launch_conv_bf16_bb_p2(const convimpl::ConvP& p, const ftc_op& o, hipStream_t s);
hipError_t launch_conv_bf16_bb_p3(const convimpl::ConvP& p, const ftc_op& o, hipStream_t s);

hipError_t launch_conv_bf16_bb(const convimpl::ConvP& p, const ftc_op& o, hipStream_t s) {
    switch (convimpl::conv_part<__bf16, __bf16>(o)) {
    case convimpl::PART_HALO: return launch_conv_bf16_bb_p0(p, o, s);
    case convimpl::PART_BK32: return launch_conv_bf16_bb_p1(p, o, s);
    case convimpl::PART_BK64: return launch_conv_bf16_bb_p2(p, o, s);
    default: return launch_conv_bf16_bb_p3(p, o, s);
    }
}
