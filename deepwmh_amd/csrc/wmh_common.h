// wmh_common.h -- shared helpers of libdeepwmh_amd.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/deepwmh_amd.h"

int wmh_fail(int code, const char* msg);           // records msg, returns code
int wmh_check_launch(const char* what);            // hipGetLastError() -> WMH_OK / WMH_ELAUNCH

__device__ __forceinline__ float wmh_lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

// ---- diagnostics live in a second library -------------------------------------------------------------------------------------
// libdeepwmh_amd.so (the product) is compiled WITHOUT -DWMH_DIAG: no s_memtime stamp branch exists in any kernel and no environment
// variable selects a kernel variant -- the variant switches below are compile-time constants (their defaults).
// libdeepwmh_amd_diag.so (deepwmh_amd/build.py, same sources with -DWMH_DIAG; load it with WMH_LIB_PATH or _lib.load_diag()) keeps
// both: tools/*_check.py, tools/cl_diag.py, bench.py's in-kernel clock leg and the "wave-specialised == one-role" tests use it.
#include <stdlib.h>
#ifdef WMH_DIAG
#define WMH_STAMPS(a) ((a).stamps)
static inline int wmh_diag_env(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
static inline bool wmh_diag_env_set(const char* name) { return getenv(name) != nullptr; }
#else
#define WMH_STAMPS(a) (static_cast<unsigned long long*>(nullptr))
#define wmh_diag_env(name, dflt) (dflt)
#define wmh_diag_env_set(name) (false)
#endif
