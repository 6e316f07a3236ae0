// One (16-bit compute type, output type) combination of the pointwise-GEMM kernel (pw_gemm_impl.h); compiled four times with
// -DFTC_PW_FN=<symbol> -DFTC_PW_W=<__bf16|_Float16> -DFTC_PW_OUT=<same|float> (build.py).
#include "pw_gemm_impl.h"

#ifndef FTC_PW_FN              // stand-alone `hipcc -c` of this file: a harmless default instance
#define FTC_PW_FN launch_pw_standalone
#define FTC_PW_W __bf16
#define FTC_PW_OUT __bf16
#endif

hipError_t FTC_PW_FN(const convimpl::ConvP& p, const ftc_op& o, hipStream_t s) { return convimpl::launch_pw<FTC_PW_W, FTC_PW_OUT>(p, o, s); }
