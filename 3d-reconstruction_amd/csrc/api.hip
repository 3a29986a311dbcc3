// Version / error-string entry points of libgsplat_mi355x.so.
#include "gs_common.hpp"

extern "C" int gs_version(void) { return 100; }  // 0.1.0 (round 1)

extern "C" const char* gs_error_string(int status) {
  switch (status) {
    case GS_OK: return "ok";
    case GS_ERR_INVALID_ARG: return "gsplat_mi355x: invalid argument";
    case GS_ERR_UNSUPPORTED: return "gsplat_mi355x: unsupported configuration";
    case GS_ERR_WORKSPACE: return "gsplat_mi355x: workspace too small";
    default: break;
  }
  if (status > 0) return hipGetErrorString((hipError_t)status);
  return "gsplat_mi355x: unknown error";
}
