// Host-side interfaces between the translation units of libkanconv (not part of the C ABI; nothing here is exported).
//   kanconv.hip     C-ABI entry points, planning, the tap-major / halo / position-major GEMM kernels, layout and norm kernels
//   kan_direct.hip  band-halo kernels for the narrow layers (few input channels, any kernel size and stride) and for wide kernels
#pragma once
#include "kanconv.h"

// sets the thread-local message behind kan_last_error() and returns -1 (defined in kanconv.hip)
int kan_fail_msg(const char* fmt, const char* a);
