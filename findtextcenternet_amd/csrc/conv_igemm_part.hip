// One PART (halo kernels | tiles with K step 32 | 64 | 128) of one heavy type combination of the implicit-GEMM conv kernel;
// compiled several times with -DFTC_PART_FN=<symbol> -DFTC_PART_OUT=<float|__bf16> -DFTC_PART=<0..3> (build.py) so that the
// ~60 instantiations of a bf16-in combination build in parallel instead of in one three-minute translation unit.
#include "conv_igemm_impl.h"

#ifndef FTC_PART_FN            // stand-alone `hipcc -c` of this file (no build.py flags): a harmless default instance
#define FTC_PART_FN launch_conv_part_standalone
#define FTC_PART_OUT __bf16
#define FTC_PART 1
#endif
#ifndef FTC_PART_W             // 16-bit compute / input type of the combination: __bf16 (default) or _Float16
#define FTC_PART_W __bf16
#endif

hipError_t FTC_PART_FN(const convimpl::ConvP& p, const ftc_op& o, hipStream_t s) {
    return convimpl::launch_part<FTC_PART_W, FTC_PART_W, FTC_PART_OUT, FTC_PART>(p, o, s);
}
