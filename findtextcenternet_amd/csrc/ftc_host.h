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
