// Library identification and error text for the C ABI (include/gwen_hip.h).
#include "common.h"

extern "C" const char *gwen_hip_version(void) { return "gwen_hip 0.1.0 gfx950"; }

extern "C" const char *gwen_hip_error_string(int code) {
  switch (code) {
    case GWEN_OK: return "success";
    case GWEN_EINVAL: return "invalid argument (size, null pointer or alignment)";
    case GWEN_ERANGE: return "size does not fit the int32 CSR / launch grid";
    case GWEN_ENOSPACE: return "workspace too small";
    default: break;
  }
  if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
  return "unknown gwen_hip error";
}
