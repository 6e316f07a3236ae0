// Host-side definitions shared by the C-ABI translation units (ftc_api.hip, model.hip).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/ftc.h"

struct ftc_plan {
    std::vector<ftc_op> ops;
    int64_t workspace_bytes;
    int64_t weights_bytes;
};

// Records the thread-local message returned by ftc_last_error() and returns `code`.
int ftc_set_error(int code, const std::string& msg);

// mbconv_slice.hip: shapes FTC_OP_MBHEAD accepts (the plan builder asks before it emits one)
bool ftc_mbhead_legal(const ftc_op& o);
int ftc_mbhead_bands(const ftc_op& o);          // workgroup bands per image (1 = the whole map)
int ftc_mbhead_band_rows(int H, int W);         // ftc_op.aux1 for an H x W map: 0 = whole map, > 0 = output rows per band, -1 = does not fit
int ftc_mbhead_slice(const ftc_op& o);          // expanded channels per workgroup (ftc_op.Cout_total, or the form's default)

// fused_mbconv.hip: shapes FTC_OP_FMBCONV accepts (the plan builder asks before it emits one)
bool ftc_fmbconv_legal(const ftc_op& o);

// conv_igemm.hip: NULL if the convolution op (incl. its tuned kernel choice ftc_op.aux0) is supported, else the reason
const char* conv_validate(const ftc_op& op);
