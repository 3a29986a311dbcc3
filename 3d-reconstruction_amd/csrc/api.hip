// Version / error-string entry points of libgsplat_mi355x.so.
#include <chrono>
#include <mutex>
#include "gs_common.hpp"

extern "C" int gs_version(void) { return 100; }  // 0.1.0 (round 1)

extern "C" const char* gs_error_string(int status) {
  switch (status) {
    case GS_OK: return "ok";
    case GS_ERR_INVALID_ARG: return "gsplat_mi355x: invalid argument";
    case GS_ERR_UNSUPPORTED: return "gsplat_mi355x: unsupported configuration";
    case GS_ERR_WORKSPACE: return "gsplat_mi355x: workspace too small";
    case GS_ERR_NOT_POSTED: return "gsplat_mi355x: the kernels did not post their results";
    case GS_ERR_CONTRACT: return "gsplat_mi355x: a kernel found its inputs violating the entry point's precondition "
                                 "(tile_order / size_classes do not describe these lists); outputs of that call are invalid";
    default: break;
  }
  if (status > 0) return hipGetErrorString((hipError_t)status);
  return "gsplat_mi355x: unknown error";
}

// ---- contract flag (gs_common.hpp): one int32 of pinned coherent host memory for the life of the process
static int32_t* g_contract_host = nullptr;
static int32_t* g_contract_dev = nullptr;
static std::once_flag g_contract_once;

int32_t* gs_contract_flag_device() {
  std::call_once(g_contract_once, [] {
    void* h = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
      (void)hipGetLastError();
      return;
    }
    *reinterpret_cast<int32_t*>(h) = 0;
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(h); return; }
    g_contract_host = reinterpret_cast<int32_t*>(h);
    g_contract_dev = reinterpret_cast<int32_t*>(d);
  });
  return g_contract_dev;
}

int gs_contract_take() {
  if (!g_contract_host) return 0;
  return (int)__atomic_exchange_n(g_contract_host, (int32_t)0, __ATOMIC_ACQ_REL);
}

extern "C" int gs_contract_status(int32_t clear) {
  if (!g_contract_host) return GS_OK;
  const int32_t v = clear ? __atomic_exchange_n(g_contract_host, (int32_t)0, __ATOMIC_ACQ_REL)
                          : __atomic_load_n(g_contract_host, __ATOMIC_ACQUIRE);
  return v ? GS_ERR_CONTRACT : GS_OK;
}

// Zero `bytes` bytes at `ptr` on `stream` (hipMemsetAsync).  The gradient record the rasterizer's backward accumulates
// into is cleared with this on a side stream while the forward runs; going through the raw stream handle spares the
// host the torch stream-context switch a `tensor.zero_()` on another stream needs.
// (A dedicated 16-B-store fill kernel clears the 320 MB record of configs[1] in 42 us against the runtime's 98 us, but
// takes more of the part away from the forward rasterizer it overlaps: 1.268-1.273 vs 1.262-1.264 ms per step.)
namespace gs {
// Large fills (the 320 MB gradient record, cleared on a side stream beside the forward rasterizer) go out as nontemporal
// stores from one workgroup per CU: the zeros are not allocated in the caches on their way to HBM (the neighbours' L2
// miss traffic is unchanged; what they come back for stays in the Infinity Cache).  Same-box A/B at configs[1] (profiles/r02_ab_runs.txt item 30): step 1.122 -> 1.100 ms
// against hipMemsetAsync; the same kernel with plain stores gains nothing.
constexpr int64_t FILL_NT_MIN_BYTES = (int64_t)32 << 20;
constexpr int FILL_NT_WGS = 256;
__global__ __launch_bounds__(256) void fill_zero_nt_kernel(float4* __restrict__ p, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    __builtin_nontemporal_store(0.f, &p[i].x);
    __builtin_nontemporal_store(0.f, &p[i].y);
    __builtin_nontemporal_store(0.f, &p[i].z);
    __builtin_nontemporal_store(0.f, &p[i].w);
  }
}
}  // namespace gs

extern "C" int gs_fill_zero(void* ptr, int64_t bytes, void* stream) {
  if (bytes < 0 || (bytes > 0 && !ptr)) return GS_ERR_INVALID_ARG;
  if (bytes == 0) return GS_OK;
  if (bytes > gs::FILL_NT_MIN_BYTES && bytes % 16 == 0 && ((uintptr_t)ptr & 15) == 0) {
    hipLaunchKernelGGL(gs::fill_zero_nt_kernel, dim3(gs::FILL_NT_WGS), dim3(256), 0, (hipStream_t)stream,
                       (float4*)ptr, bytes / 16);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? GS_OK : (int)e;
  }
  const hipError_t e = hipMemsetAsync(ptr, 0, (size_t)bytes, (hipStream_t)stream);
  return e == hipSuccess ? GS_OK : (int)e;
}

// ------------------------------------------------------------------------------------------ frames -> uint8
// The reference's host-side frame conversion (models/video_utils.py:694-700: rgb.clip(1e-6, 1 - 1e-6), (rgb * 255)
// .astype(np.uint8): truncation) in one pass over the rendered floats instead of three elementwise kernels.
namespace gs {
__device__ __forceinline__ unsigned frame_u8(float x) {
  const float lo = 1.0e-6f, hi = (float)(1.0 - 1.0e-6);
  const float v = fminf(fmaxf(x, lo), hi) * 255.0f;
  return (unsigned)(int)v;
}
__global__ __launch_bounds__(256) void frames_to_uint8_kernel(int64_t n, const float* __restrict__ src,
                                                              unsigned char* __restrict__ dst) {
  const int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 + 4 <= n && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
    const float4 v = *reinterpret_cast<const float4*>(src + i4);
    *reinterpret_cast<unsigned*>(dst + i4) =
        frame_u8(v.x) | (frame_u8(v.y) << 8) | (frame_u8(v.z) << 16) | (frame_u8(v.w) << 24);
  } else {
    for (int64_t i = i4; i < n && i < i4 + 4; ++i) dst[i] = (unsigned char)frame_u8(src[i]);
  }
}
}  // namespace gs

extern "C" int gs_frames_to_uint8(int64_t n, const float* src, uint8_t* dst, void* stream) {
  if (n < 0 || (n > 0 && (!src || !dst))) return GS_ERR_INVALID_ARG;
  if (n == 0) return GS_OK;
  const int64_t blocks = gs::ceil_div(gs::ceil_div(n, (int64_t)4), (int64_t)256);
  if (blocks > 0x7fffffff) return GS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(gs::frames_to_uint8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n, src, dst);
  GS_CHECK_LAUNCH();
  return GS_OK;
}

// ------------------------------------------------------------------------------------------ composite launches
// The host side of one rasterization() call is a dozen launches; on the shapes the reference trains at (one camera,
// 1e5 Gaussians) the step is bound by the host reaching them, not by the kernels.  These entry points issue the fixed
// launch sequences of the binned pipeline in one call each -- same kernels, same order, same stream -- so the Python
// layer crosses the FFI three times per forward instead of eleven.
extern "C" int gs_isect_bin_prepare(int32_t C, int32_t N, int32_t tile_w, int32_t tile_h, const float* splats,
                                    const int32_t* tiles_per_gauss, int32_t n_slices, uint32_t* table,
                                    uint64_t* vis_list, int32_t* vis_counts, uint64_t* vis_masks, int32_t tile_size,
                                    int32_t* tile_counts, int32_t* isect_offsets, int64_t* total, int32_t* tile_order,
                                    int32_t* size_classes, int32_t* redo_flags, int64_t* n_isects_host,
                                    int32_t* size_classes_host, int32_t synchronize, void* stream) {
  if (!n_isects_host || !total || synchronize < 0 || synchronize > 2) return GS_ERR_INVALID_ARG;
  const bool posted = synchronize == 2;  // the kernels write n_isects / the size classes into the host slot themselves
  int st = gs_isect_bin_count(C, N, tile_w, tile_h, splats, tiles_per_gauss, n_slices, table, vis_list, vis_counts,
                              vis_masks, tile_size, stream);
  if (st != GS_OK) return st;
  hipStream_t s = (hipStream_t)stream;
  const int32_t n_tiles = C * tile_w * tile_h;
  hipError_t e;
  if (posted) {
    // slot = sentinel until the kernel's store lands (the launches below order these host stores before the kernels)
    __atomic_store_n(n_isects_host, (int64_t)-1, __ATOMIC_RELAXED);
    if (tile_order && size_classes_host && size_classes)
      for (int i = 0; i < 3; ++i) __atomic_store_n(&size_classes_host[i], (int32_t)-1, __ATOMIC_RELAXED);
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
  }
  // (posted: the offsets kernel also posts the size classes the tile order will find -- same counts, one kernel earlier)
  const bool want_classes = tile_order && size_classes_host && size_classes;
  st = gs_isect_bin_offsets_post(C, tile_w, tile_h, n_slices, table, tile_counts, isect_offsets, total,
                                 posted ? n_isects_host : nullptr,
                                 (posted && want_classes) ? size_classes_host : nullptr, stream);
  if (st != GS_OK) return st;
  if (!posted) {
    // the one host synchronisation of the pipeline (gsplat has the same one): n_isects sizes the arrays that follow
    e = hipMemcpyAsync(n_isects_host, total, sizeof(int64_t), hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) return (int)e;
  }
  if (tile_order) {
    st = gs_raster_tile_order(n_tiles, nullptr, 0, tile_counts, tile_order, size_classes, stream);
    if (st != GS_OK) return st;
    if (!posted && want_classes) {  // lets the host skip the launches of size classes that hold no tile
      e = hipMemcpyAsync(size_classes_host, size_classes, 3 * sizeof(int32_t), hipMemcpyDeviceToHost, s);
      if (e != hipSuccess) return (int)e;
    }
  }
  if (redo_flags) {
    e = hipMemsetAsync(redo_flags, 0, ((size_t)n_tiles + 1) * sizeof(int32_t), s);
    if (e != hipSuccess) return (int)e;
  }
  if (synchronize != 1) return GS_OK;
  e = hipStreamSynchronize(s);
  return e == hipSuccess ? GS_OK : (int)e;
}

// ---- host slots the kernels post into (gs_isect_bin_prepare with synchronize == 2)
extern "C" int gs_host_post_alloc(void** slot, int64_t bytes) {
  if (!slot || bytes < 1) return GS_ERR_INVALID_ARG;
  // coherent (fine-grained) pinned host memory mapped into the device's address space: a system-scope store from a
  // kernel is visible to the polling host while the kernel is still running
  const hipError_t e = hipHostMalloc(slot, (size_t)bytes, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent);
  return e == hipSuccess ? GS_OK : (int)e;
}

extern "C" int gs_host_post_free(void* slot) {
  const hipError_t e = hipHostFree(slot);
  return e == hipSuccess ? GS_OK : (int)e;
}

extern "C" int gs_host_post_wait(const int64_t* n_isects_host, const int32_t* size_classes_host, int32_t timeout_ms,
                                 void* stream) {
  if (!n_isects_host || timeout_ms < 0) return GS_ERR_INVALID_ARG;
  auto ready = [&]() {
    if (__atomic_load_n(n_isects_host, __ATOMIC_ACQUIRE) == (int64_t)-1) return false;
    if (size_classes_host)
      for (int i = 0; i < 3; ++i)
        if (__atomic_load_n(&size_classes_host[i], __ATOMIC_ACQUIRE) == (int32_t)-1) return false;
    return true;
  };
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spin = 0; !ready(); ++spin) {
    __builtin_ia32_pause();
    if ((spin & 1023u) == 1023u &&
        std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(timeout_ms)) {
      // not posted in time: let the stream finish (a failed launch surfaces here) and look once more
      const hipError_t e = hipStreamSynchronize((hipStream_t)stream);
      if (e != hipSuccess) return (int)e;
      return ready() ? GS_OK : GS_ERR_NOT_POSTED;
    }
  }
  return GS_OK;
}

extern "C" int gs_stream_synchronize(void* stream) {
  const hipError_t e = hipStreamSynchronize((hipStream_t)stream);
  return e == hipSuccess ? GS_OK : (int)e;
}

extern "C" int gs_isect_bin_build(int32_t C, int32_t N, int32_t tile_w, int32_t tile_h, int32_t n_slices,
                                  int32_t slices_per_group, int32_t n_regions, const uint64_t* vis_list,
                                  const int32_t* vis_counts, const uint64_t* vis_masks, const uint32_t* table,
                                  const int32_t* isect_offsets, int64_t n_isects, const int32_t* tile_order,
                                  const int32_t* size_classes, int32_t k_target, uint64_t* pairs, int64_t* isect_ids,
                                  int32_t* flatten_ids, int32_t* sorted_counts, const int32_t* size_classes_host,
                                  void* stream) {
  int st = gs_isect_bin_scatter_regions(C, N, tile_w, tile_h, n_slices, slices_per_group, n_regions, vis_list,
                                        vis_counts, vis_masks, table, isect_offsets, sorted_counts, pairs, stream);
  if (st != GS_OK) return st;
  if (k_target > 0)
    return gs_isect_tile_sort_front_ex(C, tile_w, tile_h, isect_offsets, n_isects, tile_order, size_classes, k_target,
                                       pairs, isect_ids, flatten_ids, sorted_counts, size_classes_host, stream);
  return gs_isect_tile_sort_ex(C, tile_w, tile_h, isect_offsets, n_isects, tile_order, size_classes, pairs, isect_ids,
                               flatten_ids, size_classes_host, stream);
}

extern "C" int gs_rasterize_fwd_front(int32_t C, int32_t width, int32_t height, int32_t tile_size, int32_t channels,
                                      const float* splats, const int32_t* isect_offsets, int32_t* flatten_ids,
                                      int64_t n_isects, const float* backgrounds, int32_t ed_last_channel,
                                      float* render_colors, float* render_alphas, int32_t* last_ids,
                                      const int32_t* tile_order, int32_t* tile_work, int32_t* sorted_counts,
                                      int32_t* redo_flags, const float* opacity_override, int32_t n_per_camera,
                                      float* final_T, const int32_t* size_classes, uint64_t* pairs, int64_t* isect_ids,
                                      const int32_t* size_classes_host, void* stream) {
  if (!sorted_counts || !redo_flags || !tile_order || !size_classes || !pairs) return GS_ERR_INVALID_ARG;
  if (tile_size < 1) return GS_ERR_INVALID_ARG;
  int st = gs_rasterize_fwd(C, width, height, tile_size, channels, splats, isect_offsets, flatten_ids, n_isects,
                            backgrounds, ed_last_channel, render_colors, render_alphas, last_ids, tile_order, tile_work,
                            sorted_counts, redo_flags, 0, opacity_override, n_per_camera, final_T, stream);
  if (st != GS_OK) return st;
  const int32_t tw = (width + tile_size - 1) / tile_size, th = (height + tile_size - 1) / tile_size;
  st = gs_isect_tile_sort_redo_ex(C, tw, th, isect_offsets, n_isects, tile_order, size_classes, redo_flags, pairs,
                                  isect_ids, flatten_ids, sorted_counts, size_classes_host, stream);
  if (st != GS_OK) return st;
  return gs_rasterize_fwd(C, width, height, tile_size, channels, splats, isect_offsets, flatten_ids, n_isects,
                          backgrounds, ed_last_channel, render_colors, render_alphas, last_ids, tile_order, tile_work,
                          sorted_counts, redo_flags, 1, opacity_override, n_per_camera, final_T, stream);
}
