// Shared host-side helpers for the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdarg.h>

namespace mt {

inline char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(-2, "%s: launch failed: %s", what, hipGetErrorString(e));
  return 0;
}

enum { MT_ERR_ARG = -1, MT_ERR_LAUNCH = -2, MT_ERR_UNSUPPORTED = -3 };

}  // namespace mt
