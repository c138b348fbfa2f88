// Error plumbing shared by the extern "C" translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/arcflow_hip.h"

extern thread_local char afx_g_err[512];
int afx_fail(int code, const char* fmt, ...);
#define fail afx_fail

#define HIP_TRY(expr)                                                                           \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) return fail(AFX_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)
