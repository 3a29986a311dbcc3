/*
 * gsplat_mi355x.h -- C ABI of libgsplat_mi355x.so, the MI355X (gfx950) native implementation of
 * the gsplat `rasterization()` hot path that the reference calls.
 *
 * Reference interface each entry point stands in for (R = /root/reference/3D-Reconstruction-main;
 * the callee is the un-vendored third-party package `gsplat`, so the citation is the reference's
 * binding/call site plus the gsplat stage of SURVEY.md section 8(a) it replaces):
 *
 *   gs_projection_fwd / _bwd        R/models/gaussians/basics.py:12 -> gsplat.rendering.rasterization,
 *                                   stage `fully_fused_projection` fwd/bwd (8a rows 2,3) fused with the
 *                                   in-rasterization SH colour evaluation (8a row 4), the opacity x
 *                                   compensation glue, the depth-channel `cat` (8a row 1 glue) and the
 *                                   per-Gaussian tile count of `isect_tiles` pass 1 (8a row 5).
 *   gs_sh_fwd / _bwd                R/models/gaussians/basics.py:15 -> gsplat.cuda._wrapper.spherical_harmonics
 *                                   (called at R/models/gaussians/vanilla.py:369, R/models/nodes/rigid.py:486,
 *                                   smpl.py:475, deformable.py:83) (8a row 4).
 *   gs_exclusive_scan_i32,
 *   gs_isect_emit                   `isect_tiles` cumsum + pass 2 (8a row 5).
 *   gs_radix_sort_pairs_u64_u32     the cub::DeviceRadixSort::SortPairs<int64,int32> inside `isect_tiles` (8a row 6).
 *   gs_isect_offset_encode          `isect_offset_encode` (8a row 7).
 *   gs_rasterize_fwd / _bwd         `rasterize_to_pixels` fwd/bwd incl. the "ED" normalisation (8a rows 8,9,10),
 *                                   call site R/models/trainers/base.py:393-408 and :772-787.
 *   gs_compact_visible              the CSR emission of `fully_fused_projection_packed`, packed=True.
 *
 * Conventions: every pointer is a DEVICE pointer unless its name ends in `_host`; all tensors are
 * dense row-major fp32 / int32 / int64 as stated; `stream` is a hipStream_t passed as void*.
 * Every function returns 0 on success, a positive hipError_t, or a negative GS_ERR_* code, and
 * never throws.  The library never allocates device memory: scratch space is passed in by the caller
 * (sizes from the *_workspace_bytes functions), and it keeps no state between calls -- with one
 * stated exception, the static per-device scratch of gs_radix_sort_pairs_u64_u32_static.
 *
 * Splat record ("splats", 16 floats = 64 B per (camera, Gaussian), one HBM request per gather):
 *   [0] mean2d.x  [1] mean2d.y  [2] conic.a  [3] conic.b  [4] conic.c  [5] opacity*compensation
 *   [6..9] up to 4 colour channels (colours, then the depth channel when requested; unused = 0)
 *   [10] depth  [11] radius (int32 bits)  [12] compensation  [13] tile min x|y<<16  [14] tile max x|y<<16
 *   [15] unused
 * Gradient record ("v_splats", same shape): [0,1] v_mean2d  [2..4] v_conic  [5] v_opacity_eff
 *   [6..9] v_colour channels  [10,11] sum over pixels of |v_mean2d| (absgrad)  [12..15] unused.
 */
#ifndef GSPLAT_MI355X_H
#define GSPLAT_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS_REC 16 /* floats per splat record */

#define GS_OK 0
#define GS_ERR_INVALID_ARG (-1)
#define GS_ERR_UNSUPPORTED (-2)
#define GS_ERR_WORKSPACE (-3)
#define GS_ERR_NOT_POSTED (-4) /* gs_host_post_wait: the stream finished without the kernels having posted */
#define GS_ERR_CONTRACT (-5)   /* a kernel found device-side inputs violating a documented precondition (below) */

/* colour source of gs_projection_fwd/_bwd */
#define GS_COLOR_NONE 0 /* no colour input (render_mode "D"/"ED")                    */
#define GS_COLOR_N 1    /* colors [N, color_dim], shared by all cameras               */
#define GS_COLOR_CN 2   /* colors [C, N, color_dim]                                   */
#define GS_COLOR_SH 3   /* SH coefficients [N, sh_K, 3], evaluated per (camera, n)    */

typedef struct gs_proj_params {
  int32_t N, C;
  int32_t width, height;
  int32_t tile_size, tile_w, tile_h;
  float eps2d, near_plane, far_plane, radius_clip;
  int32_t calc_compensations; /* rasterize_mode == "antialiased" */
  int32_t color_mode;         /* GS_COLOR_*                      */
  int32_t color_dim;          /* channels of `colors` (SH: 3)    */
  int32_t sh_degree, sh_K;    /* GS_COLOR_SH only                */
  int32_t depth_channel;      /* 1: append depth after colours   */
  int32_t sh_rows;            /* GS_COLOR_SH, forward only: when a Gaussian's coefficient row is requested.
                                 0 = chosen by camera count (up front from four cameras on), 1 = up front (most rows get
                                 used: a rig looking all around), 2 = at the first camera that sees the Gaussian (few
                                 rows get used: one or two cameras, neighbouring views of a trajectory).  Speed only. */
} gs_proj_params;

int gs_version(void);
const char* gs_error_string(int status);
/* Zeroes `bytes` at `ptr` on `stream`: clears the gradient record gs_rasterize_bwd accumulates into (a 16-B aligned
 * buffer above 32 MiB by nontemporal stores, so that the zeros are not allocated in the caches on their way to HBM;
 * hipMemsetAsync otherwise). */
int gs_fill_zero(void* ptr, int64_t bytes, void* stream);
/* Rendered floats -> video frames as the reference converts them on the host
 * (/root/reference/3D-Reconstruction-main/models/video_utils.py:694-700): dst[i] = (uint8)(clip(src[i], 1e-6, 1 - 1e-6) * 255),
 * truncated; one pass.  n = number of values. */
int gs_frames_to_uint8(int64_t n, const float* src, uint8_t* dst, void* stream);

/* ---- fully_fused_projection (+SH, +opacity*comp, +tile count) ------------------------------- */
int gs_projection_fwd(const gs_proj_params* p,
                      const float* means,      /* [N,3]                         */
                      const float* quats,      /* [N,4] wxyz, normalised inside */
                      const float* scales,     /* [N,3]                         */
                      const float* opacities,  /* [N]                           */
                      const float* viewmats,   /* [C,4,4] world->camera         */
                      const float* Ks,         /* [C,3,3]                       */
                      const float* campos,     /* [C,3] camera centres for the SH view directions; NULL: derived from viewmats in-kernel */
                      const float* colors,     /* per color_mode, may be NULL   */
                      int32_t* radii,          /* out [C,N]                     */
                      float* means2d,          /* out [C,N,2], or NULL: not written  */
                      float* splats,           /* out [C,N,16], rows with radii==0 untouched */
                      int32_t* tiles_per_gauss,/* out [C,N]                     */
                      const float* covars,     /* [N,6] world covariances {xx,xy,xz,yy,yz,zz} (gsplat `covars=`) or NULL;
                                                  when given, quats / scales are ignored and may be NULL */
                      uint64_t* vis_words,     /* out [C, ceil(N/64)] or NULL: bit n%64 of word n/64 = (c, n) visible */
                      int32_t* vis_counts,     /* out [C, ceil(N/64)] (with vis_words): set bits of each word; its exclusive
                                                  scan (gs_exclusive_scan_i32) is the `vis_prefix` of gs_projection_bwd_split
                                                  = the row of each pair in gsplat's packed (c, n) order (A.7) */
                      void* stream);

int gs_projection_bwd(const gs_proj_params* p,
                      const float* means, const float* quats, const float* scales,
                      const float* opacities, const float* viewmats, const float* Ks,
                      const float* campos, const float* colors,
                      const int32_t* radii,    /* [C,N]    from forward         */
                      const float* splats,     /* [C,N,16] from forward         */
                      const float* v_means2d,  /* [C,N,*] gradient of means2d, row stride ld_v_means2d floats */
                      int64_t ld_v_means2d,
                      const float* v_splats,   /* [C,N,16] gradient record ([0,1] ignored) */
                      float* v_means,          /* out [N,3]                     */
                      float* v_quats,          /* out [N,4]                     */
                      float* v_scales,         /* out [N,3]                     */
                      float* v_opacities,      /* out [N]                       */
                      float* v_colors,         /* out, shape of `colors`, may be NULL */
                      float* v_viewmats,       /* out [C,4,4], PRE-ZEROED, may be NULL */
                      float* v_campos,         /* out [C,3],   PRE-ZEROED, may be NULL */
                      const float* covars,     /* [N,6] or NULL, as in the forward (not with GS_COLOR_SH) */
                      float* v_covars,         /* out [N,6] when covars is given (v_quats / v_scales unused) */
                      void* stream);

/* ---- backward split for camera-sharded data parallelism (no reference counterpart: the reference is single-GPU,
 * R/models/trainers/base.py:411) ------------------------------------------------------------------
 * With SH colours evaluated inside `rasterization` (GS_COLOR_SH), the SH-coefficient gradient is 12 K bytes per
 * Gaussian but only rank 1 per camera: basis(dir) (x) v_rgb.  gs_projection_bwd_split does the geometry half of
 * gs_projection_bwd (v_means without the view-direction term, v_quats, v_scales, v_opacities, v_viewmats without the
 * camera-centre term) and writes the clamp-gated colour gradient of every visible (camera, Gaussian) pair, 12 B each,
 * packed in ascending (c, n) order, into a "colour-gradient message":
 *     [int32 n_cams, 12 B reserved | campos Cl x 3 f32 | words Cl x W u64 | prefix Cl x W i32 | vals cap x 3 f32]
 * with W = ceil(N / 64) and Cl = `layout_cams` >= this call's C: every rank of an exchange lays its message out for the
 * largest camera count of any rank, so all messages have one size, and records its own count in n_cams
 * (sections 256-B aligned; byte offsets from gs_colour_message_layout(Cl, N, cap): {campos, words, prefix, vals, total}).
 * gs_sh_bwd_packed turns n_msgs such messages, msg_stride bytes apart (this rank's own, or every rank's after ONE
 * all-gather), into v_sh [N,K,3] and the view-direction part of v_means, in a fixed order (bit-identical on every
 * rank).  cap >= number of visible pairs (rows beyond cap are dropped). */
/* Exclusive scan of vis_counts [C][n_words] (gs_projection_fwd) -> vis_prefix [C][n_words] (the first packed row of
 * every word, what gs_projection_bwd_split takes), and stat[0] = number of visible pairs, stat[1] = C (int64 device
 * array of 2: what the ranks MAX-reduce to agree on `cap` and `layout_cams`).  Workspace as for
 * gs_exclusive_scan_i32 over C * n_words elements. */
int gs_visibility_prefix(int32_t C, int32_t n_words, const int32_t* vis_counts, int32_t* vis_prefix, int64_t* stat,
                         void* workspace, int64_t workspace_bytes, void* stream);
int gs_colour_message_layout(int32_t C, int32_t N, int64_t cap, int64_t* out5_host);
int gs_projection_bwd_split(const gs_proj_params* p,
                            const float* means, const float* quats, const float* scales,
                            const float* opacities, const float* viewmats, const float* Ks,
                            const float* campos,        /* [C,3] or NULL (derived from viewmats) */
                            const int32_t* radii, const float* splats,
                            const float* v_means2d, int64_t ld_v_means2d, const float* v_splats,
                            const int32_t* vis_prefix,  /* [C, W] exclusive scan of gs_projection_fwd's vis_counts */
                            float* v_means, float* v_quats, float* v_scales, float* v_opacities,
                            float* v_viewmats,          /* out [C,4,4] PRE-ZEROED or NULL */
                            void* message,              /* out, gs_colour_message_layout(layout_cams, N, cap)[4] bytes */
                            int64_t cap, int32_t layout_cams, void* stream);
int gs_sh_bwd_packed(int32_t N, int32_t K, int32_t degree, int32_t C /* layout_cams of the messages */, int32_t n_msgs,
                     int64_t msg_stride, int64_t cap, const float* means, const float* coeffs /* [N,K,3] */,
                     const void* messages,
                     float* v_coeffs,           /* out [N,K,3], written whole */
                     float* v_means,            /* [N,3]: view-direction part of v_means, += when accumulate_means else = */
                     int32_t accumulate_means,
                     int32_t local_msg,         /* index of this rank's own message (for v_campos) */
                     float* v_campos,           /* out [C,3] PRE-ZEROED or NULL: camera-centre gradient of the local cameras */
                     void* stream);

/* ---- spherical_harmonics --------------------------------------------------------------------- */
int gs_sh_fwd(int64_t n, int32_t K, int32_t degree,
              const float* dirs,    /* [n,3]   */
              const float* coeffs,  /* [n,K,3] */
              const uint8_t* masks, /* [n] or NULL */
              float* out,           /* [n,3], masked rows = 0 */
              void* stream);
int gs_sh_bwd(int64_t n, int32_t K, int32_t degree,
              const float* dirs, const float* coeffs, const uint8_t* masks,
              const float* v_out,   /* [n,3]   */
              float* v_coeffs,      /* out [n,K,3] (all K written; unused bands = 0) */
              float* v_dirs,        /* out [n,3] or NULL */
              void* stream);

/* ---- isect_tiles / offsets -------------------------------------------------------------------- */
int64_t gs_exclusive_scan_workspace_bytes(int64_t n);
int gs_exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n,
                          int64_t* total,      /* out, device, 1 element */
                          void* workspace, int64_t workspace_bytes, void* stream);

int gs_isect_emit(int32_t C, int32_t N, int32_t tile_w, int32_t tile_h,
                  const float* splats,             /* [C,N,16] */
                  const int32_t* tiles_per_gauss,  /* [C,N]    */
                  const int32_t* cum_excl,         /* [C,N] exclusive scan of tiles_per_gauss */
                  int64_t* isect_ids,              /* out [n_isects] */
                  int32_t* flatten_ids,            /* out [n_isects] */
                  void* stream);

int64_t gs_radix_sort_workspace_bytes(int64_t n);
/* Stable LSD radix sort of (key,value) pairs on key bits [0,end_bit).  Ping-pongs between the a and b
 * buffers; *result_in_b_host (HOST int) says which pair holds the sorted output. */
int gs_radix_sort_pairs_u64_u32(int64_t n, int32_t end_bit,
                                uint64_t* keys_a, uint32_t* vals_a,
                                uint64_t* keys_b, uint32_t* vals_b,
                                void* workspace, int64_t workspace_bytes,
                                int32_t* result_in_b_host, void* stream);

/* Same, on key bits [begin_bit, end_bit) only. */
int gs_radix_sort_pairs_u64_u32_bits(int64_t n, int32_t begin_bit, int32_t end_bit,
                                     uint64_t* keys_a, uint32_t* vals_a,
                                     uint64_t* keys_b, uint32_t* vals_b,
                                     void* workspace, int64_t workspace_bytes,
                                     int32_t* result_in_b_host, void* stream);

/* Same, with the element count optionally in DEVICE memory: n = min(*n_dev, n_max) when n_dev != NULL, else n_max.
 * Buffers, workspace and grids are sized for n_max, so the host can enqueue the sort before it knows n. */
int gs_radix_sort_pairs_u64_u32_ex(int64_t n_max, const int64_t* n_dev, int32_t begin_bit, int32_t end_bit,
                                   uint64_t* keys_a, uint32_t* vals_a,
                                   uint64_t* keys_b, uint32_t* vals_b,
                                   void* workspace, int64_t workspace_bytes,
                                   int32_t* result_in_b_host, void* stream);

/* The same sort with NO workspace argument and no allocation of any kind.  It has no counterpart in the reference (nor
 * in gsplat, whose cub sort takes a caller-allocated temp buffer): it is for callers that cannot allocate -- code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc -- and so cannot size and provide the
 * workspace of the entry points above.  The digit table lives in a static device array of the library
 * (GS_RADIX_SORT_STATIC_MAX_N / 4096 * 2 KiB + 2 KiB = 8 MiB + 2 KiB per GPU the library runs on, held until the process
 * ends); n > GS_RADIX_SORT_STATIC_MAX_N returns GS_ERR_WORKSPACE.  Same kernels, same result bit for bit as
 * gs_radix_sort_pairs_u64_u32_bits.  All calls on one GPU share that GPU's table: calls that overlap in time on one GPU
 * (different streams or threads) race -- serialise them.  The stream's device must be the CURRENT device (else
 * GS_ERR_INVALID_ARG: every GPU has its own copy of the table at its own address).  The first call on a device looks that
 * address up (hipGetSymbolAddress, may load the code object): make one call outside any stream capture per device first
 * -- n = 0 is enough -- before calling it on a capturing stream. */
#define GS_RADIX_SORT_STATIC_MAX_N (1 << 24)
int gs_radix_sort_pairs_u64_u32_static(int64_t n, int32_t begin_bit, int32_t end_bit,
                                       uint64_t* keys_a, uint32_t* vals_a,
                                       uint64_t* keys_b, uint32_t* vals_b,
                                       int32_t* result_in_b_host, void* stream);

int gs_isect_offset_encode(int64_t n_isects, const int64_t* isect_ids_sorted,
                           int32_t C, int32_t tile_w, int32_t tile_h,
                           int32_t* offsets, /* out [C,tile_h,tile_w] */
                           void* stream);

/* ---- binned isect_tiles + sort + offsets ------------------------------------------------------------------
 * One pipeline that produces exactly what gs_isect_emit + the full-width stable sort + gs_isect_offset_encode produce
 * (8a rows 5, 6, 7: same sorted isect_ids / flatten_ids / isect_offsets, bit for bit) without materialising or
 * radix-sorting the unsorted key array: per-camera tile histograms over `n_slices` row slices, per-tile offsets, an
 * LDS-cursor scatter of (depth bits, flat row id) pairs into the tile segments, and a per-tile LDS sort by depth with
 * ties ordered by row id.  Requires tile_w * tile_h <= gs_isect_bin_max_tiles().
 *   table        uint32 [C][n_slices][tile_h*tile_w]  scratch (count writes it, offsets turns it into slice prefixes)
 *   tile_counts  int32  [C*tile_h*tile_w]             intersections per tile
 *   total        int64  device scalar                 n_isects
 *   pairs        uint64 [n_isects]                    scratch (low word depth bits, high word flat row id)      */
int32_t gs_isect_bin_max_tiles(void);
/* vis_list / vis_counts (both or neither): the count pass also leaves every slice's visible rows as 16-byte entries
 * {flat row id, tile box lo, tile box hi, depth bits} in vis_list [C][n_slices][gs_isect_bin_slice_rows(N, n_slices)]
 * (uint64 pairs, 16-byte aligned) and their number in vis_counts [C][n_slices], for gs_isect_bin_scatter_regions.
 *
 * Culled lists (vis_masks != NULL, needs vis_list; tile_size = the rasterizer's tile edge in pixels): a (row, tile)
 * pair is counted only if the splat can reach alpha >= 1/255 on some pixel centre of the tile -- the rasterizer's own
 * conservative whole-tile test, so every pair dropped is one that each pixel of the tile would skip (Appendix A.5,
 * `alpha < 1/255 -> continue`) and NO rendered value or gradient changes.  The tile lists are then a subset of gsplat's
 * (tile_counts / isect_offsets / n_isects describe the culled lists; tiles_per_gauss stays the box count), each still
 * in gsplat's (depth, row id) order.  The decision is taken once, by this pass: vis_masks [C][n_slices][slice_rows]
 * (uint64) holds, for every visible-list entry whose box has at most 64 tiles, bit t = tile t of the box (row-major)
 * takes the pair; larger boxes are not culled.  gs_isect_bin_scatter_regions must be given the same vis_masks.
 * With vis_masks == NULL the pipeline is gsplat's, bit for bit (tile_size is ignored). */
int32_t gs_isect_bin_slice_rows(int32_t N, int32_t n_slices);
int gs_isect_bin_count(int32_t C, int32_t N, int32_t tile_w, int32_t tile_h, const float* splats,
                       const int32_t* tiles_per_gauss, int32_t n_slices, uint32_t* table, uint64_t* vis_list,
                       int32_t* vis_counts, uint64_t* vis_masks, int32_t tile_size, void* stream);
int gs_isect_bin_offsets(int32_t C, int32_t tile_w, int32_t tile_h, int32_t n_slices, uint32_t* table,
                         int32_t* tile_counts, int32_t* isect_offsets, int64_t* total, void* stream);
/* ... the same, and the kernel also stores the total into *total_post and the sort's size classes (#lists longer than
 * 2048, 1024, 512 entries: what gs_raster_tile_order by list length will report) into size_classes_post[0..2] (either
 * NULL: not), host slots from gs_host_post_alloc that the caller polls with gs_host_post_wait ("posted results"
 * below). */
int gs_isect_bin_offsets_post(int32_t C, int32_t tile_w, int32_t tile_h, int32_t n_slices, uint32_t* table,
                              int32_t* tile_counts, int32_t* isect_offsets, int64_t* total, int64_t* total_post,
                              int32_t* size_classes_post, void* stream);
int gs_isect_bin_scatter(int32_t C, int32_t N, int32_t tile_w, int32_t tile_h, const float* splats,
                         const int32_t* tiles_per_gauss, int32_t n_slices, const uint32_t* table,
                         const int32_t* isect_offsets, uint64_t* pairs, void* stream);
/* Same result as gs_isect_bin_scatter up to the order inside a tile segment (which the sort fixes), from the visible
 * lists of the count pass: a workgroup owns a band of tile rows (n_regions bands) and slices_per_group consecutive
 * slices, so the runs it writes into a tile segment are slices_per_group * n_regions times as long.
 * tile_counts (gs_isect_bin_offsets' output, or NULL): with the list lengths at hand the workgroup sorts its pairs by
 * tile in LDS first and writes every tile's run in one piece (consecutive lanes, consecutive entries) instead of one
 * 8-byte store per pair; NULL, or a band whose cursors leave no room for a staging area, takes the direct stores. */
int gs_isect_bin_scatter_regions(int32_t C, int32_t N, int32_t tile_w, int32_t tile_h, int32_t n_slices,
                                 int32_t slices_per_group, int32_t n_regions, const uint64_t* vis_list,
                                 const int32_t* vis_counts, const uint64_t* vis_masks /* or NULL: see above */,
                                 const uint32_t* table, const int32_t* isect_offsets,
                                 const int32_t* tile_counts /* or NULL */, uint64_t* pairs, void* stream);
int gs_isect_tile_sort(int32_t C, int32_t tile_w, int32_t tile_h, const int32_t* isect_offsets, int64_t n_isects,
                       const int32_t* tile_order /* launch order from gs_raster_tile_order, or NULL */,
                       const int32_t* size_classes /* its size_classes output (device int32[3]), or NULL */,
                       uint64_t* pairs, int64_t* isect_ids /* out [n_isects] */, int32_t* flatten_ids /* out */,
                       void* stream);

/* PRECONDITION of every gs_isect_tile_sort* entry point given a tile_order / size_classes pair: the pair must be the
 * output of gs_raster_tile_order (or gs_isect_bin_prepare) BY LIST LENGTH for exactly these isect_offsets -- the sort
 * launches one kernel per contiguous size class (lists > 2048 | 1025..2048 | 513..1024 | <= 512 entries) and each
 * class sorts in an LDS buffer of its class's capacity.  An order made from other lengths (the backward's order by
 * walked length, another call's lists) breaks that.  The kernels check it: a list longer than its class's capacity is
 * NOT sorted -- its flatten_ids segment receives the list's ids in arrival order (valid ids, so the rasterizer reads
 * wrong-ordered but in-bounds records) -- and the process-wide contract flag is raised.  The next call of one of these
 * entry points, of gs_rasterize_fwd[_front], or of gs_contract_status returns GS_ERR_CONTRACT once and clears the flag
 * (asynchronous, like a sticky stream error: the call that returns it is not necessarily the one that caused it;
 * synchronise the stream before gs_contract_status to attribute it).  No reference counterpart (gsplat's radix sort has
 * no such precondition). */
int gs_contract_status(int32_t clear /* 1: reset the flag */);

/* Front-only variant of gs_isect_tile_sort (no reference counterpart: gsplat sorts every key).  Segments longer than
 * 1024 entries get only their leading >= k_target entries (1 <= k_target <= 1024) sorted and written to
 * isect_ids / flatten_ids; sorted_counts[tile] (pre-filled by the caller with the list lengths) receives that count.
 * Shorter segments, and long ones whose front is degenerate, are sorted whole.  The written prefix of every segment is
 * bit-identical to the same positions of gs_isect_tile_sort's output; the rest of isect_ids / flatten_ids is
 * unspecified until gs_isect_tile_sort_redo (tiles with redo_flags != 0, as set by gs_rasterize_fwd) or
 * gs_isect_tile_sort (every tile; `pairs` stays valid for both) completes it.  tile_order and size_classes are required. */
int gs_isect_tile_sort_front(int32_t C, int32_t tile_w, int32_t tile_h, const int32_t* isect_offsets, int64_t n_isects,
                             const int32_t* tile_order, const int32_t* size_classes, int32_t k_target, uint64_t* pairs,
                             int64_t* isect_ids, int32_t* flatten_ids, int32_t* sorted_counts /* in/out */, void* stream);
int gs_isect_tile_sort_redo(int32_t C, int32_t tile_w, int32_t tile_h, const int32_t* isect_offsets, int64_t n_isects,
                            const int32_t* tile_order, const int32_t* size_classes,
                            const int32_t* redo_flags /* [C*tile_h*tile_w + 1], last element = number of flagged tiles */,
                            uint64_t* pairs, int64_t* isect_ids, int32_t* flatten_ids, int32_t* sorted_counts,
                            void* stream);

/* ---- rasterize_to_pixels ---------------------------------------------------------------------- */
int gs_rasterize_fwd(int32_t C, int32_t width, int32_t height, int32_t tile_size,
                     int32_t channels,            /* 1..4 colour channels of the splat record */
                     const float* splats,         /* [*,16] indexed by flatten_ids */
                     const int32_t* isect_offsets,/* [C,tile_h,tile_w] */
                     const int32_t* flatten_ids,  /* [n_isects] sorted */
                     int64_t n_isects,
                     const float* backgrounds,    /* [C,channels] or NULL */
                     int32_t ed_last_channel,     /* 1: last channel /= max(alpha,1e-10) */
                     float* render_colors,        /* out [C,H,W,channels] */
                     float* render_alphas,        /* out [C,H,W]          */
                     int32_t* last_ids,           /* out [C,H,W]          */
                     const int32_t* tile_order,   /* [C*tile_h*tile_w] launch order of the tiles (gs_raster_tile_order) or NULL */
                     int32_t* tile_work,          /* out [C*tile_h*tile_w] list entries the backward will walk, or NULL */
                     const int32_t* sorted_counts,/* [C*tile_h*tile_w] from gs_isect_tile_sort_front: only that many leading
                                                     entries of a tile's list are sorted yet; NULL = lists fully sorted */
                     int32_t* redo_flags,         /* in/out [C*tile_h*tile_w + 1], zeroed by the caller; needed with
                                                     sorted_counts.  Pass 0 sets 1 where pixels were still live at the end
                                                     of the sorted front and counts those tiles in the last element;
                                                     pass 1 rasterizes exactly those tiles again */
                     int32_t redo_pass,           /* 0: normal launch; 1: after gs_isect_tile_sort_redo */
                     const float* opacity_override,/* [n_per_camera] or NULL: composite with opacity_override[row %
                                                     n_per_camera] * compensation instead of the record's opacity (the
                                                     reference's per-class re-renders, R/models/trainers/scene_graph.py:
                                                     282-301, reuse one projection + sort and change only this) */
                     int32_t n_per_camera,        /* N: rows per camera of `splats` (used with opacity_override) */
                     float* final_T,              /* out [C,H,W] or NULL: each pixel's final transmittance, exactly
                                                     (render_alphas = 1 - final_T rounds its low bits away) */
                     void* stream);

int gs_rasterize_bwd(int32_t C, int32_t width, int32_t height, int32_t tile_size,
                     int32_t channels,
                     const float* splats, const int32_t* isect_offsets,
                     const int32_t* flatten_ids, int64_t n_isects,
                     const float* backgrounds, int32_t ed_last_channel,
                     const float* render_colors,  /* forward outputs */
                     const float* render_alphas, const int32_t* last_ids,
                     const float* v_render_colors,/* [C,H,W,channels] */
                     const float* v_render_alphas,/* [C,H,W]          */
                     int32_t absgrad,
                     float* v_splats,             /* in/out [*,16], PRE-ZEROED, atomically accumulated */
                     const int32_t* tile_order,   /* launch order of the tiles or NULL */
                     const float* final_T,        /* [C,H,W] from the forward, or NULL: use 1 - render_alphas as gsplat does
                                                     (loses up to 3e-4 relative on saturated pixels' gradients) */
                     void* stream);

/* Longest-first launch order of the tiles (a permutation of 0..n_tiles-1; speed only, never results): by
 * `tile_work` when given (written by gs_rasterize_fwd), else by list length from isect_offsets. No reference
 * counterpart: gsplat launches tiles in grid order. */
int gs_raster_tile_order(int32_t n_tiles, const int32_t* isect_offsets, int64_t n_isects,
                         const int32_t* tile_work, int32_t* tile_order,
                         int32_t* size_classes /* out, device int32[3] or NULL: #tiles with cost > 2048, > 1024, > 512;
                                                  the order is class-major, so these delimit contiguous ranges */,
                         void* stream);
/* ... the same, and the kernel also stores the three size classes into size_classes_post[0..2] (NULL: not), part of a
 * host slot from gs_host_post_alloc ("posted results" below). */
int gs_raster_tile_order_post(int32_t n_tiles, const int32_t* isect_offsets, int64_t n_isects,
                              const int32_t* tile_work, int32_t* tile_order, int32_t* size_classes,
                              int32_t* size_classes_post, void* stream);

/* ---- posted results (no reference counterpart: gsplat reads n_isects with a blocking device-to-host copy) ----------
 * The pipeline's one host read -- n_isects, which sizes the arrays that follow -- costs more than the copy: a copy
 * kernel behind the producer, the runtime's wake-up of the waiting thread, and a GPU that idles from then until the
 * host has issued the next launches.  Here the producing kernels store their results straight into a host slot and the
 * host polls it: the value is on the host a microsecond after the kernel computed it, while the launches queued behind
 * the producer (tile order, flag clears) are still running.
 *   gs_host_post_alloc   pinned, coherent (fine-grained) host memory mapped into the device's address space; keep it
 *                        for the life of the process and give every host thread its own slot (>= 24 bytes: int64
 *                        n_isects, then int32[3] size classes at byte offset 8)
 *   gs_host_post_wait    spins until the slot(s) hold a posted value (sentinel -1, set by gs_isect_bin_prepare before it
 *                        launches); after timeout_ms without one it synchronises `stream` and looks once more:
 *                        GS_ERR_NOT_POSTED if still nothing, else GS_OK
 * Only slots from gs_host_post_alloc may be given to the *_post entry points and to gs_isect_bin_prepare(synchronize = 2):
 * other host memory may not be visible to the host before the stream is synchronised. */
int gs_host_post_alloc(void** slot, int64_t bytes);
int gs_host_post_free(void* slot);
int gs_host_post_wait(const int64_t* n_isects_host, const int32_t* size_classes_host /* or NULL */,
                      int32_t timeout_ms, void* stream);

/* ---- composite launches (no reference counterpart: gsplat's Python layer issues its kernels one by one) ----------
 * Fixed launch sequences of the binned pipeline in one call each, for hosts whose per-call cost matters (the Python
 * layer on launch-bound shapes).  Same kernels, same order, same stream as the individual entry points above; every
 * argument means what it means there.
 *   gs_isect_bin_prepare   = gs_isect_bin_count + gs_isect_bin_offsets + gs_raster_tile_order (by tile_counts; skipped
 *                            if tile_order == NULL) + zeroing of redo_flags [C*tile_h*tile_w + 1] (skipped if NULL), then
 *                            copies `total` to *n_isects_host and, if `synchronize` != 0, SYNCHRONISES the stream: on
 *                            return *n_isects_host is valid (the pipeline's one host synchronisation; pinned memory
 *                            avoids a staging copy).  With synchronize == 0 the caller does its remaining host work
 *                            first and then calls gs_stream_synchronize(stream) before reading *n_isects_host.
 *                            synchronize == 2 ("posted results" above): no copies are queued, the kernels post into
 *                            n_isects_host / size_classes_host (slots from gs_host_post_alloc) and the caller waits
 *                            with gs_host_post_wait.
 *   gs_isect_bin_build     = gs_isect_bin_scatter_regions (tile_counts = sorted_counts as it arrives: the list
 *                            lengths, or NULL) + gs_isect_tile_sort_front (k_target > 0; sorted_counts must then hold
 *                            the list lengths) or gs_isect_tile_sort (k_target == 0).
 *   gs_rasterize_fwd_front = gs_rasterize_fwd pass 0 + gs_isect_tile_sort_redo + gs_rasterize_fwd pass 1.             */
int gs_isect_bin_prepare(int32_t C, int32_t N, int32_t tile_w, int32_t tile_h, const float* splats,
                         const int32_t* tiles_per_gauss, int32_t n_slices, uint32_t* table, uint64_t* vis_list,
                         int32_t* vis_counts, uint64_t* vis_masks, int32_t tile_size, int32_t* tile_counts,
                         int32_t* isect_offsets, int64_t* total, int32_t* tile_order, int32_t* size_classes,
                         int32_t* redo_flags, int64_t* n_isects_host,
                         int32_t* size_classes_host /* out int32[3] or NULL: host copy of size_classes */,
                         int32_t synchronize, void* stream);
int gs_stream_synchronize(void* stream); /* hipStreamSynchronize */
int gs_isect_bin_build(int32_t C, int32_t N, int32_t tile_w, int32_t tile_h, int32_t n_slices,
                       int32_t slices_per_group, int32_t n_regions, const uint64_t* vis_list, const int32_t* vis_counts,
                       const uint64_t* vis_masks, const uint32_t* table, const int32_t* isect_offsets, int64_t n_isects,
                       const int32_t* tile_order, const int32_t* size_classes, int32_t k_target, uint64_t* pairs,
                       int64_t* isect_ids, int32_t* flatten_ids, int32_t* sorted_counts,
                       const int32_t* size_classes_host /* or NULL */, void* stream);
int gs_rasterize_fwd_front(int32_t C, int32_t width, int32_t height, int32_t tile_size, int32_t channels,
                           const float* splats, const int32_t* isect_offsets, int32_t* flatten_ids, int64_t n_isects,
                           const float* backgrounds, int32_t ed_last_channel, float* render_colors,
                           float* render_alphas, int32_t* last_ids, const int32_t* tile_order, int32_t* tile_work,
                           int32_t* sorted_counts, int32_t* redo_flags, const float* opacity_override,
                           int32_t n_per_camera, float* final_T, const int32_t* size_classes, uint64_t* pairs,
                           int64_t* isect_ids, const int32_t* size_classes_host /* or NULL */, void* stream);
/* ---- device-side n_isects (no reference counterpart: gsplat reads the count on the host, as the entry points above do)
 * The count of intersections sizes the pair / id arrays, so the host of the entry points above waits for it in the middle
 * of the forward -- on launch-bound shapes (the reference's own step: one camera, base.py:393-411) the GPU then idles
 * until the host has issued the rest.  These two variants take the count from DEVICE memory (`n_isects_dev` = the
 * `total` output of gs_isect_bin_prepare) and a host-chosen `capacity` (>= 1) that the arrays were allocated with --
 * normally a little above the last step's count:
 *   gs_isect_bin_build_dev  = gs_isect_bin_scatter_regions + gs_isect_tile_sort (every list whole, all four size
 *                             classes launched; isect_ids may be NULL, and then a list above 6144 entries raises the
 *                             contract flag: gs_contract_status) with pairs / flatten_ids (/ isect_ids) of `capacity`
 *   gs_rasterize_fwd_dev    = gs_rasterize_fwd (no front-sorted lists) on such arrays
 * If *n_isects_dev > capacity the kernels write NOTHING (no list, no pixel): the caller reads the posted count once
 * the launches are issued (gs_host_post_wait: no stall, the producer finished long before) and, on overflow, repeats
 * the two calls with arrays of the right size.  With *n_isects_dev <= capacity the results are bit-identical to the
 * exact-size entry points; the elements of the arrays beyond the count are not written. */
int gs_isect_bin_build_dev(int32_t C, int32_t N, int32_t tile_w, int32_t tile_h, int32_t n_slices,
                           int32_t slices_per_group, int32_t n_regions, const uint64_t* vis_list,
                           const int32_t* vis_counts, const uint64_t* vis_masks, const uint32_t* table,
                           const int32_t* isect_offsets, int64_t capacity, const int64_t* n_isects_dev,
                           const int32_t* tile_order, const int32_t* size_classes, uint64_t* pairs,
                           int64_t* isect_ids /* [capacity] or NULL */, int32_t* flatten_ids /* [capacity] */,
                           const int32_t* tile_counts /* or NULL: direct 8-B stores */, void* stream);
int gs_rasterize_fwd_dev(int32_t C, int32_t width, int32_t height, int32_t tile_size, int32_t channels,
                         const float* splats, const int32_t* isect_offsets, const int32_t* flatten_ids,
                         int64_t capacity, const int64_t* n_isects_dev, const float* backgrounds,
                         int32_t ed_last_channel, float* render_colors, float* render_alphas, int32_t* last_ids,
                         const int32_t* tile_order, int32_t* tile_work, const float* opacity_override,
                         int32_t n_per_camera, float* final_T, void* stream);

/* The three sort entry points with a HOST copy of size_classes (from gs_isect_bin_prepare; NULL = as the plain entry
 * points): launches for size classes that hold no tile are skipped -- on launch-bound shapes only the shortest class
 * exists, and a launch that finds nothing to do still costs microseconds.  gs_isect_tile_sort_front_ex with no tile
 * above 1024 entries launches neither the front kernel nor the long classes.
 * With size_classes_host given and no list above 2048 entries (size_classes_host[0] == 0) isect_ids may be NULL: the
 * keys are then not written (a third of the sort's output bytes) -- the rasterizer reads flatten_ids and the offsets
 * only.  Longer lists are sorted on global scratch that borrows the isect_ids segment, so there it is required. */
int gs_isect_tile_sort_ex(int32_t C, int32_t tile_w, int32_t tile_h, const int32_t* isect_offsets, int64_t n_isects,
                          const int32_t* tile_order, const int32_t* size_classes, uint64_t* pairs, int64_t* isect_ids,
                          int32_t* flatten_ids, const int32_t* size_classes_host, void* stream);
int gs_isect_tile_sort_front_ex(int32_t C, int32_t tile_w, int32_t tile_h, const int32_t* isect_offsets, int64_t n_isects,
                                const int32_t* tile_order, const int32_t* size_classes, int32_t k_target,
                                uint64_t* pairs, int64_t* isect_ids, int32_t* flatten_ids, int32_t* sorted_counts,
                                const int32_t* size_classes_host, void* stream);
int gs_isect_tile_sort_redo_ex(int32_t C, int32_t tile_w, int32_t tile_h, const int32_t* isect_offsets, int64_t n_isects,
                               const int32_t* tile_order, const int32_t* size_classes, const int32_t* redo_flags,
                               uint64_t* pairs, int64_t* isect_ids, int32_t* flatten_ids, int32_t* sorted_counts,
                               const int32_t* size_classes_host, void* stream);

/* ---- packed mode ------------------------------------------------------------------------------ */
/* Emits, in ascending (camera, gaussian) order, the rows with radii > 0. cum_excl = exclusive scan of
 * (radii > 0) computed with gs_exclusive_scan_i32 over `visible` (int32 0/1) written by gs_mark_visible. */
int gs_mark_visible(int64_t n, const int32_t* radii, int32_t* visible, void* stream);
int gs_compact_visible(int32_t C, int32_t N, const int32_t* radii, const int32_t* cum_excl,
                       const float* splats, const float* means2d,
                       int64_t* camera_ids, int64_t* gaussian_ids,   /* out [nnz] */
                       int32_t* radii_packed,                         /* out [nnz] */
                       float* means2d_packed,                         /* out [nnz,2] */
                       float* depths_packed,                          /* out [nnz]   */
                       float* conics_packed,                          /* out [nnz,3] */
                       void* stream);

/* Instance-pose transform of the rigid nodes' points (R/models/nodes/rigid.py:403-470), per point with p = point_ids[i]:
 *   out_means = quat_to_rotmat(a_p / |a_p|) means + t_p,   out_quats = (a_p / |a_p|) (x) (quats / |quats|)   (Hamilton, wxyz)
 * with a = inst_quats [n_inst,4], t = inst_trans [n_inst,3] (the frame's pose rows).  The backward accumulates the pose
 * gradients into g_inst_quats / g_inst_trans (zeroed by the caller); g_* and v_* may be NULL. */
int gs_rigid_transform_fwd(int64_t n, int32_t n_inst, const float* means, const float* quats, const int64_t* point_ids,
                           const float* inst_quats, const float* inst_trans, float* out_means, float* out_quats,
                           void* stream);
int gs_rigid_transform_bwd(int64_t n, int32_t n_inst, const float* means, const float* quats, const int64_t* point_ids,
                           const float* inst_quats, const float* v_world_means, const float* v_world_quats,
                           float* g_means, float* g_quats, float* g_inst_quats, float* g_inst_trans, void* stream);

/* ---- densification statistics (next to the path: SURVEY.md 8(f) rank 2) -------------------------- */
/* One fused pass of BasicTrainer.postprocess_per_train_step's gradient scaling (R/models/trainers/base.py:279-297)
 * + VanillaGaussians.after_train (R/models/gaussians/vanilla.py:154-182).  grad2d points at the 2-float screen-space
 * gradient of row (c, n) with a row stride of `ld` floats (means2d.absgrad / .grad are views of the gradient record:
 * ld = 16); scale_x = width / 2 * batch_size, scale_y = height / 2 * batch_size, inv_last_size = 1 / max(W, H).
 * first != 0: initialise (grad_norm_sum = |grad| for every row, vis_counts = 1, max_2dsize from zero). */
int gs_densify_accumulate(int32_t C, int32_t N, const int32_t* radii /* [C,N] */, const float* grad2d, int64_t ld,
                          float scale_x, float scale_y, float inv_last_size, int32_t first,
                          float* grad_norm_sum /* [N] in/out */, float* vis_counts /* [N] in/out */,
                          float* max_2dsize /* [N] in/out */, void* stream);

/* ---- caller pre-pass, fused (next to the path: SURVEY.md 8(f) rank 1, training half) --------------- */
/* One kernel per Gaussian class for what VanillaGaussians.get_gaussians (R/models/gaussians/vanilla.py:359-393) and
 * BasicTrainer.collect_gaussians' torch.cat (R/models/trainers/base.py:365-366) do with ~25 torch ops:
 *   rgbs = clamp(SH_degree(normalize(means - cam_center), cat(features_dc, features_rest)) + 0.5, 0, 1)
 *          (degree = -1: rgbs = sigmoid(features_dc), vanilla.py:371-372)
 *   opacities = sigmoid(raw_opacities) [* opacity_mask, R/models/nodes/rigid.py:493]
 *   scales = exp(raw_scales)  (scale_mode 0: [n,3]; 1: [n,1] repeated 3x; 2: [n,3], third axis 0; vanilla.py:112-128)
 *   quats = raw_quats / |raw_quats|,  means copied
 * written into this class's row slice of the scene-level arrays (the out_* pointers are already offset).
 * clamp_mask [n] u8 keeps which colour channels passed the clamp (for the backward); finite_flag is one u32 that
 * gets bit 2*key (NaN) / 2*key+1 (Inf) OR-ed in for key 0 _means, 1 _opacities, 2 _rgbs, 3 _scales, 4 _quats
 * (the reference's isnan/isinf checks, vanilla.py:388-393, without their ten host synchronisations).
 * K_rest = rows of features_rest per Gaussian (K - 1); (degree + 1)^2 - 1 <= K_rest. */
int gs_activate_fwd(int64_t n, int32_t K_rest, int32_t degree, const float* means /* [n,3] */,
                    const float* cam_center /* [3] device */, const float* features_dc /* [n,3] */,
                    const float* features_rest /* [n,K_rest,3] */, const float* raw_opacities /* [n] */,
                    const float* opacity_mask /* [n] or NULL */, const float* raw_scales, int32_t scale_mode,
                    const float* raw_quats /* [n,4] */, float* out_means, float* out_opacities, float* out_rgbs,
                    float* out_scales, float* out_quats, uint8_t* clamp_mask, uint32_t* finite_flag, void* stream);
/* vjp of gs_activate_fwd: gradients of the raw Parameters from the gradients of the class's slices.  The means pass
 * through (view directions are detached, vanilla.py:366), so v_means needs no kernel.  g_features_rest is written
 * whole (zeros above the degree in use). */
int gs_activate_bwd(int64_t n, int32_t K_rest, int32_t degree, const float* means, const float* cam_center,
                    const float* features_dc, const float* raw_opacities, const float* opacity_mask,
                    const float* raw_scales, int32_t scale_mode, const float* raw_quats, const uint8_t* clamp_mask,
                    const float* v_opacities, const float* v_rgbs, const float* v_scales, const float* v_quats,
                    float* g_features_dc, float* g_features_rest, float* g_raw_opacities, float* g_raw_scales,
                    float* g_raw_quats, void* stream);

/* ---- densify / cull of one Gaussian class (next to the path: SURVEY.md 8(f) rank 2, second half) ---- */
/* VanillaGaussians.refinement_after (R/models/gaussians/vanilla.py:197-283: split_gaussians :314-342, dup_gaussians
 * :344-357, cull_gaussians :285-312) and the Adam-moment surgery (R/models/gaussians/basics.py:162-206), composed:
 * decide -> gs_exclusive_scan_i32 over counts [4n+1] -> plan -> one gather per tensor.
 * gs_refine_decide: flags [n] u8 (1 split, 2 dup, 4 row survives, 8 its split children survive, 16 its duplicate
 * survives); counts [4n+1] i32 = the columns {split, survives, children survive, duplicate survives} then a 0.
 * densify_size_thresh / cull_scale_thresh are already multiplied by scene_scale; split_screen_size, cull_scale_thresh,
 * cull_screen_size < 0 switch that test off (the reference's step conditions :217, :292, :299). */
int gs_refine_decide(int64_t n, int32_t scale_mode, int32_t do_densify, int32_t do_cull, float densify_grad_thresh,
                     float densify_size_thresh, float split_screen_size, float cull_alpha_thresh,
                     float cull_scale_thresh, float cull_screen_size, const float* raw_scales,
                     const float* raw_opacities, const float* grad_norm_sum, const float* vis_counts,
                     const float* max_2dsize, uint8_t* flags, int32_t* counts, void* stream);
/* scan = exclusive scan of counts.  Output rows in the reference's order [survivors | split children, sample-major |
 * duplicates]: n_out = scan[4n] - scan[n] + (n_split_samples - 1) * (scan[3n] - scan[2n]).  kinds: 0 old row,
 * 1 duplicate, 2 + s split child of sample s; sample_idx = s * n_splits + rank among ALL split rows (the row of the
 * reference's `randn((samps * n_splits, 3))` draw, :321), -1 otherwise. */
int gs_refine_plan(int64_t n, int32_t n_split_samples, const uint8_t* flags, const int32_t* scan, int32_t* src_rows,
                   uint8_t* kinds, int32_t* sample_idx, void* stream);
/* out [n_out,W] = in [src_rows,:]; mode 0 copy (features, opacities, quats), 1 zeros for new rows (Adam exp_avg /
 * exp_avg_sq), 2 log(exp(x) / 1.6) on rows whose source was split (scales, :338-339).  W <= 256. */
int gs_refine_gather_rows(int64_t n_out, int32_t W, int32_t mode, const int32_t* src_rows, const uint8_t* kinds,
                          const uint8_t* flags, const float* in, float* out, void* stream);
/* new means: split children = mean + R(q/|q|) (exp(scales) * samples[sample_idx]) (:321-329), other rows copied. */
int gs_refine_split_means(int64_t n_out, int32_t scale_mode, const int32_t* src_rows, const uint8_t* kinds,
                          const int32_t* sample_idx, const float* means, const float* raw_scales,
                          const float* raw_quats, const float* samples /* [n_split_samples * n_splits, 3] */,
                          float* out_means, void* stream);

/* ---- RGB loss terms on the render (next to the path: SURVEY.md 8(f) rank 4) ------------------------- */
/* BasicTrainer.compute_losses' rgb terms (R/models/trainers/base.py:533-541): with pred, gt [H,W,3] and an optional
 * validity mask [H,W] multiplied into both,
 *   out[0] = |gt - pred|.mean()                                           (:540)
 *   out[1] = pytorch_msssim.SSIM(data_range=1, size_average=True, channel=3)(gt, pred)   (:114, :541; the caller takes 1 - )
 * (11-tap sigma-1.5 Gaussian window, VALID separable filter, K = (0.01, 0.03); pytorch_msssim 1.0.0,
 * R/requirements.txt:7).  H, W > 10.  maps: NULL, or [(H-10)*(W-10)*9] floats kept for the backward.
 * workspace: gs_l1_ssim_workspace_floats(H, W) floats (per-workgroup partial sums; the reduction order is fixed). */
int64_t gs_l1_ssim_workspace_floats(int32_t H, int32_t W);
int gs_l1_ssim_fwd(int32_t H, int32_t W, const float* pred, const float* gt, const float* mask, float* maps,
                   float* workspace, int64_t workspace_floats, float* out /* [2] */, void* stream);
/* v_pred [H,W,3] = g_out[0] * d out[0] / d pred + g_out[1] * d out[1] / d pred; g_out: 2 floats on the device. */
int gs_l1_ssim_bwd(int32_t H, int32_t W, const float* pred, const float* gt, const float* mask, const float* maps,
                   const float* g_out, float* v_pred, void* stream);

/* The per-pixel scalar terms of compute_losses in the shipped configuration (R/configs/omnire.yaml:24-31), one pass:
 *   out[0] = binary_cross_entropy(opacity * valid, (1 - sky_mask) * valid).mean()   (R/models/trainers/base.py:536-550)
 *   out[1] = DepthLoss("l1", normalize=False, inverse=False, "mean_on_hit")(depth, lidar_depth, (lidar_depth > 0) * valid)
 *            (R/models/losses.py:96-199, base.py:553-557), out[2] = number of pixels it averages over
 *   out[3] = (-p log p).mean(), p = clamp(opacity, 1e-6, 1 - 1e-6)                  (base.py:568-573)
 * over n = H*W pixels; valid_mask, sky_mask, lidar_depth may be NULL. */
int64_t gs_pixel_losses_workspace_floats(void);
int gs_pixel_losses_fwd(int64_t n, const float* opacity, const float* depth, const float* sky_mask,
                        const float* valid_mask, const float* lidar_depth, float max_depth, float* workspace,
                        int64_t workspace_floats, float* out /* [4] */, void* stream);
int gs_pixel_losses_bwd(int64_t n, const float* opacity, const float* depth, const float* sky_mask,
                        const float* valid_mask, const float* lidar_depth, float max_depth, const float* fwd_out,
                        const float* g_out /* [4] device */, float* v_opacity, float* v_depth /* or NULL */,
                        void* stream);

/* The same terms in every configuration the reference can be given (R/models/trainers/base.py:231-250,
 * R/models/losses.py): mask loss "bce" | "safe_bce" (SafeBCE, losses.py:33-83, limit 0.1 at base.py:238), DepthLoss
 * loss_type l1 | l2 | smooth_l1, normalize, use_inverse_depth, reduction mean_on_hit | mean_on_hw | sum
 * (losses.py:91-199).  `select` [n] u8 or NULL restricts the depth term to a subset of the selected pixels
 * (depth_error_percentile, losses.py:163-171: the host ranks `err_map` and passes the keep-mask back); `err_map` [n]
 * or NULL receives each pixel's depth error (-1 where the pixel is not selected). */
typedef struct gs_pixel_loss_opts {
  int32_t bce_mode;        /* 0 binary_cross_entropy, 1 safe_binary_cross_entropy */
  float bce_limit;         /* safe_bce: limit (0.1) */
  int32_t depth_type;      /* 0 l1, 1 l2, 2 smooth_l1 */
  int32_t depth_normalize; /* safe_normalize_depth: clamp(x / max_depth, 1e-6, 1) */
  int32_t depth_inverse;   /* 1 / x behind the normalisation */
  int32_t depth_reduction; /* 0 mean_on_hit, 1 mean_on_hw, 2 sum */
  float max_depth;         /* DepthLoss.upper_bound */
} gs_pixel_loss_opts;
int gs_pixel_losses_ex_fwd(int64_t n, const gs_pixel_loss_opts* opts, const float* opacity, const float* depth,
                           const float* sky_mask, const float* valid_mask, const float* lidar_depth,
                           const uint8_t* select, float* err_map, float* workspace, int64_t workspace_floats,
                           float* out /* [4] */, void* stream);
int gs_pixel_losses_ex_bwd(int64_t n, const gs_pixel_loss_opts* opts, const float* opacity, const float* depth,
                           const float* sky_mask, const float* valid_mask, const float* lidar_depth,
                           const uint8_t* select, const float* fwd_out, const float* g_out, float* v_opacity,
                           float* v_depth /* or NULL */, void* stream);

/* kornia.losses.inverse_depth_smoothness_loss(1 / (depth + 1e-5), image) (R/models/trainers/base.py:575-585; kornia is not
 * available here: parity unpinned, the published formula restated): depth [H,W], image [H,W,3], out [1];
 * workspace: gs_pixel_losses_workspace_floats() floats. */
int gs_inv_depth_smooth_fwd(int32_t H, int32_t W, const float* depth, const float* image, float* workspace,
                            int64_t workspace_floats, float* out, void* stream);
int gs_inv_depth_smooth_bwd(int32_t H, int32_t W, const float* depth, const float* image, const float* g_out /* [1] */,
                            float* v_depth, void* stream);

/* Dynamic-region RGB loss (R/models/trainers/base.py:598-613): mean over the pixels with region > thresh (and
 * valid_mask != 0) and the 3 channels of |gt * valid - pred * valid|.  out [2] = {loss (0 when no pixel qualifies: the
 * reference then adds no term), number of pixels}. */
int gs_region_l1_fwd(int64_t n, const float* pred /* [n,3] */, const float* gt, const float* valid_mask,
                     const float* region, float thresh, float* workspace, int64_t workspace_floats, float* out,
                     void* stream);
int gs_region_l1_bwd(int64_t n, const float* pred, const float* gt, const float* valid_mask, const float* region,
                     float thresh, const float* fwd_out, const float* g_out /* [1] */, float* v_pred, void* stream);

/* Affine regulariser (R/models/trainers/base.py:587-596): |A[:3,:3] - I|.mean() + |A[:3,3]|.mean() over m matrices of
 * `rows` x 4 (rows >= 3).  out [1] and / or v_A (same shape as A, rows >= 3 left untouched) from g_out [1]. */
int gs_affine_reg(int32_t m, int32_t rows, const float* A, const float* g_out, float* out, float* v_A, void* stream);

/* Opacity reset of VanillaGaussians.refinement_after (R/models/gaussians/vanilla.py:268-280), in place:
 * raw = logit(min(sigmoid(raw), reset_alpha_value)); the opacity parameter's Adam moments (may be NULL) = 0. */
int gs_opacity_reset(int64_t n, float reset_alpha_value, float* raw_opacities, float* exp_avg, float* exp_avg_sq,
                     void* stream);

/* ---- sky environment light (next to the path: SURVEY.md 8(f) rank 4) --------------------------------- */
/* EnvLight.forward (R/models/modules.py:191-203): nvdiffrast.torch.texture(base[None], l, filter_mode='linear',
 * boundary_mode='cube') -- bilinear, seamless cube-map lookup (OpenGL face order +x,-x,+y,-y,+z,-z and orientation).
 * tex [6,R,R,C], dirs [n,3] (any length), rot: NULL or 9 floats on the device, d' = rot d (the `to_opengl` matrix of
 * modules.py:186,194), out [n,C].  nvdiffrast is CUDA-only and not available here: parity unpinned. */
int gs_cube_texture_fwd(int64_t n, int32_t R, int32_t C, const float* tex, const float* dirs, const float* rot,
                        float* out, void* stream);
/* v_tex [6,R,R,C] (zeroed by the caller) += the transpose of the lookup applied to v_out [n,C]. */
int gs_cube_texture_bwd(int64_t n, int32_t R, int32_t C, const float* dirs, const float* rot, const float* v_out,
                        float* v_tex, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GSPLAT_MI355X_H */
