// Library-internal helpers shared by the C-ABI translation units (error slot, HIP error mapping).
#pragma once
#include "../../include/streamformer_hip.h"
#include <hip/hip_runtime.h>

// formats into the thread-local message returned by sf_last_error(); returns `code`
int sf_set_err(int code, const char* fmt, ...);

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) return sf_set_err(SF_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)
