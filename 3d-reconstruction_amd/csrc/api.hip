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

// Zero `bytes` bytes at `ptr` on `stream` (hipMemsetAsync).  The gradient record the rasterizer's backward accumulates
// into is cleared with this on a side stream while the forward runs; going through the raw stream handle spares the
// host the torch stream-context switch a `tensor.zero_()` on another stream needs.
// (A dedicated 16-B-store fill kernel clears the 320 MB record of configs[1] in 42 us against the runtime's 98 us, but
// takes more of the part away from the forward rasterizer it overlaps: 1.268-1.273 vs 1.262-1.264 ms per step.)
extern "C" int gs_fill_zero(void* ptr, int64_t bytes, void* stream) {
  if (bytes < 0 || (bytes > 0 && !ptr)) return GS_ERR_INVALID_ARG;
  if (bytes == 0) return GS_OK;
  const hipError_t e = hipMemsetAsync(ptr, 0, (size_t)bytes, (hipStream_t)stream);
  return e == hipSuccess ? GS_OK : (int)e;
}
