// pysonic_amd/csrc/lib_common.hpp -- error plumbing shared by the translation units of
// libpysonic_amd.so.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/pysonic_amd.h"

inline std::string &last_error_string()
{
    static thread_local std::string s;
    return s;
}

inline int set_error(int code, const std::string &msg)
{
    last_error_string() = msg;
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return set_error(SONIC_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)
