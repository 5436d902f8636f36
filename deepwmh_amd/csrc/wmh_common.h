// wmh_common.h -- shared helpers of libdeepwmh_amd.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/deepwmh_amd.h"

int wmh_fail(int code, const char* msg);           // records msg, returns code
int wmh_check_launch(const char* what);            // hipGetLastError() -> WMH_OK / WMH_ELAUNCH

__device__ __forceinline__ float wmh_lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }
