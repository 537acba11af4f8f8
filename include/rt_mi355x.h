/*
 * rt_mi355x.h -- C ABI of librt_mi355x.so, the MI355X-native drop-in for the RayTracer/
 * sub-project of ipilter/RayTracerTest.
 *
 * The reference's boundary is a C++ class in a static library (RayTracer/RayTracer.h:14-41,
 * callback type RayTracer/RaytracerCallback.h:9); its only caller is
 * OpenGLView/MainFrame.cpp:45,219-220,233,249,254,293,311,438.  This header is the flat
 * extern "C" form of exactly that class -- one entry point per public method, same
 * argument order, units and error behaviour -- so that any FFI (the header-only C++
 * class in include/RayTracer.h, ctypes in raytracertest_amd/api.py, cgo, JNI ...) binds
 * the same symbols.  Plain pointers and sizes only.
 *
 * Declared semantic change (BASELINE.json north_star): the OpenGL PBO interop is cut.  The
 * pointer handed to the callbacks is a HOST-readable BGRA8 image (pinned memory owned by
 * the tracer, valid until the next rt_tracer_resize / rt_tracer_destroy), not a device
 * pointer as in RayTracerImpl.cu:272,304.
 *
 * Error behaviour follows the reference: nothing throws across the API.  Functions that
 * the reference declares void either return void here or an int status that callers may
 * ignore; the text of the last failure is kept (rt_tracer_last_error / rt_last_error).
 */
#ifndef RT_MI355X_H
#define RT_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK              0
#define RT_ERR_INVALID     1   /* bad argument (e.g. UploadScene size not a multiple of 3) */
#define RT_ERR_NO_DEVICE   2   /* no usable HIP device: the HIP path is mandatory, there is no CPU fallback */
#define RT_ERR_HIP         3   /* a HIP runtime call or kernel launch failed */
#define RT_ERR_STATE       4

#define RT_MATH_FMA        0u  /* default: the documented a*b+c shapes are fused (nvcc -fmad=true analogue) */
#define RT_MATH_STRICT     1u  /* every source-level operation rounds separately */

#define RT_FLAG_NO_FILTER  1u  /* disable the conservative wave-uniform rejections (debug / parity tests) */
#define RT_FLAG_NO_BINNING 2u  /* disable the per-tile triangle classification: every ray scans the whole
                                  list from block-staged LDS chunks (debug / parity tests / A-B) */

#define RT_FLAG_NEAREST_HIT 4u  /* hit selection extension: keep the nearest hit with t > 0 instead of the
                                  reference's farthest hit incl. negative t (Kernels.cuh:73,84).  Not the default. */

#define RT_FLAG_NO_MACRO_BINS 16u /* disable the macro-tile level of the classification (large scenes): every block
                                  pre-culls the whole triangle list (debug / parity tests / A-B) */
#define RT_FLAG_NO_SUPER_BINS 64u /* dense scenes: the macro tiles scan the whole triangle list instead of their super tile's
                                  (debug / parity tests / A-B; the result is the same) */
#define RT_FLAG_NO_SURE_HIT 32u /* small scenes: run the intersection tests also on tiles whose candidate list is one triangle that
                                  every ray of the tile certainly hits (debug / parity tests / A-B; the result is the same) */
#define RT_FLAG_SMOOTH_NORMALS 8u /* shading extension for scenes uploaded with rt_tracer_upload_scene_edges: the
                                  colour is |normalize(w*n0 + u*n1 + v*n2)|, the rows' packed vertex normals
                                  interpolated at the hit (w = (1-u)-v), instead of |face normal| (Kernels.cuh:97-99).
                                  No effect on scenes uploaded as absolute vertices.  Not the default. */

#define RT_BUF_RENDER      0   /* rows*W*4 float  RGBA accumulators   (mRenderBuffer)      */
#define RT_BUF_COUNTS      1   /* rows*W   uint32 sample counts       (mSampleCountBuffer) */
#define RT_BUF_IMAGE       2   /* rows*W   uint32 BGRA8               (mImageBuffer)       */
#define RT_BUF_RNG         3   /* 6 planes of rows*W uint32: d, v0..v4 (mRandomStates)     */
#define RT_BUF_FRAME       4   /* H*W uint32 BGRA8: the gathered frame of a sharded image, on its root only */
#define RT_GROUP_ID_BYTES  128 /* rt_group_unique_id (an ncclUniqueId)                      */

typedef struct rt_tracer rt_tracer;                      /* opaque: rt::RayTracer + rt::RayTracerImpl */
typedef struct rt_float4 { float x, y, z, w; } rt_float4;   /* CUDA's float4, RayTracer.h:34 */

/* rt::CallBackFunction, RaytracerCallback.h:9: (ColorPtr imageBuffer, size_t size in bytes) + user data.
 * Runs on the tracer's render thread (RayTracerImpl.cu:256-272,287-305). */
typedef void (*rt_callback_fn)(uint32_t* imageBuffer, size_t size, void* user);

/* Extensions the reference has no way to express (all optional; zero-initialise + struct_size). */
typedef struct rt_options {
  uint32_t struct_size;        /* sizeof(rt_options) */
  int32_t  device;             /* HIP device ordinal; the reference pins device 0 (GLCanvas.cpp:259-260) */
  uint32_t full_height;        /* 0: the tracer owns the whole image.  Otherwise the image is
                                  imageSize[0] x full_height and this tracer owns the row band
                                  [row_begin, row_begin + imageSize[1])  (multi-GPU sharding) */
  uint32_t row_begin;
  uint32_t use_time_seed;      /* 1: seed = (uint32)time(NULL) like Random.cu:45 (default when no options) */
  uint32_t math_mode;          /* RT_MATH_FMA | RT_MATH_STRICT */
  uint64_t seed;               /* curand_init seed when use_time_seed == 0 */
  uint32_t flags;              /* RT_FLAG_* */
  uint32_t samples_in_flight;  /* samples of a pixel kept in registers per pass (1,2,4; 0 = auto) */
  uint32_t lds_chunk;          /* full-scan path: triangles staged in LDS at a time (0 = auto) */
  uint32_t bin_list;           /* binned path: candidate records per wave in LDS (0 = auto; multiple of 64) */
  uint32_t transport;          /* rt_tracer_create_multi: RT_TRANSPORT_RCCL (default) | RT_TRANSPORT_PEER */
} rt_options;

#define RT_TRANSPORT_RCCL  0u  /* finished tiles travel to the root with grouped ncclSend / ncclRecv (RCCL over xGMI) */
#define RT_TRANSPORT_PEER  1u  /* one process only: the bands' trace kernels store their BGRA8 tiles straight into the root's frame
                                  through a peer mapping (hipDeviceEnablePeerAccess): no collective, no kernels on the root.  Falls
                                  back to RCCL, with the reason in rt_tracer_group_info, when a device may not map the root's memory */

/* ---- the reference's public methods, one to one -------------------------------------- */

/* RayTracer::RayTracer, RayTracer.h:17-22.  imageSize = {width, height} pixels, fov in
 * degrees, cameraAngles in radians.  cameraPosition is stored and ignored exactly like
 * the reference (ThinLensCamera.cuh:54-57,132-141: the camera sits at the world origin). */
int  rt_tracer_create(const uint32_t imageSize[2], const float cameraPosition[3],
                      const float cameraAngles[2], float fov, float focalLength, float aperture,
                      rt_tracer** out);
int  rt_tracer_create_ex(const uint32_t imageSize[2], const float cameraPosition[3],
                         const float cameraAngles[2], float fov, float focalLength, float aperture,
                         const rt_options* options, rt_tracer** out);
/* RayTracer::~RayTracer, RayTracer.h:23: cancels and joins a running trace. */
void rt_tracer_destroy(rt_tracer* t);
/* RayTracer::Trace, RayTracer.h:25-27: asynchronous; cancels+joins a previous run, clears
 * the buffers, runs iterationCount launches of samplesPerIteration samples, fires the
 * update callback when i > 0 && updateInterval > 0 && i % updateInterval == 0 and the
 * finished callback at the end (not when stopped).  RayTracerImpl.cu:69-87,236-315. */
int  rt_tracer_trace(rt_tracer* t, uint32_t iterationCount, uint32_t samplesPerIteration,
                     uint32_t updateInterval);
/* RayTracer::Stop, RayTracer.h:28: sets the cancel flag (granularity: one kernel). */
void rt_tracer_stop(rt_tracer* t);
/* RayTracer::Resize, RayTracer.h:29: new buffers, RNG states re-created.  (Joins a running
 * trace first; the reference races here, RayTracerImpl.cu:94-103.) */
int  rt_tracer_resize(rt_tracer* t, const uint32_t size[2]);
/* RayTracer::SetCameraParameters, RayTracer.h:30-32 (fov degrees). */
void rt_tracer_set_camera_parameters(rt_tracer* t, float fov, float focalLength, float aperture);
/* RayTracer::RotateCamera, RayTracer.h:33: angles += delta (radians), matrix rebuilt. */
void rt_tracer_rotate_camera(rt_tracer* t, const float angles[2]);
/* RayTracer::UploadScene, RayTracer.h:34: count float4, three absolute vertices per
 * triangle, .w ignored; count < 3 or count % 3 != 0 is rejected and the previous scene
 * kept (RayTracerImpl.cu:121-125). */
int  rt_tracer_upload_scene(rt_tracer* t, const rt_float4* hostData, size_t count);
/* Same scene in the layout the reference's notes plan for the GPU (Documentation/gpu.meshes.txt:16-17):
 * per triangle v0, e0 = v1 - v0, e1 = v2 - v0, each .w free for a packed vertex normal
 * (rt_pack_normal).  With e0/e1 computed in fp32 the result equals rt_tracer_upload_scene. */
int  rt_tracer_upload_scene_edges(rt_tracer* t, const rt_float4* hostData, size_t count);
/* Corrected form of the reference's experimental normal packing (UnitTests/NormalPackingTest.cpp:10-23):
 * three components in [-1,1] as 8-bit fields in the fraction of one float.  unpack(pack(n)) == n for
 * every n whose components are multiples of 1/127. */
float rt_pack_normal(const float n[3]);
void  rt_unpack_normal(float packed, float n[3]);
/* RayTracer::SetUpdateCallback / SetFinishedCallback, RayTracer.h:36-37. */
void rt_tracer_set_update_callback(rt_tracer* t, rt_callback_fn fn, void* user);
void rt_tracer_set_finished_callback(rt_tracer* t, rt_callback_fn fn, void* user);

/* ---- additive extensions --------------------------------------------------------------- */

/* Block until the render thread of the last rt_tracer_trace has ended.  Returns 1 when it
 * ran to completion (finished callback fired), 0 when it was stopped or failed. */
int  rt_tracer_wait(rt_tracer* t);
/* Re-create the RNG states from an explicit seed (the reference has no seed control). */
int  rt_tracer_set_seed(rt_tracer* t, uint64_t seed);
/* Spheres: count float4 = centre xyz + radius (build-defined, Documentation/ray.sphere.png). */
int  rt_tracer_upload_spheres(rt_tracer* t, const rt_float4* spheres, size_t count);
/* Device-resident form of one Trace for throughput measurement and multi-GPU drivers:
 * enqueue clear + iterationCount trace launches + conversion on the tracer's stream, no
 * callbacks, no host synchronisation.  rt_tracer_sync waits for the stream. */
int  rt_tracer_trace_enqueue(rt_tracer* t, uint32_t iterationCount, uint32_t samplesPerIteration);
/* n_steps consecutive passes of rt_tracer_trace_enqueue, enqueued by one call (the host loop of a throughput
 * driver runs inside the library: one lock, no per-step crossing of a foreign-function boundary).  Identical to
 * n_steps calls of rt_tracer_trace_enqueue. */
int  rt_tracer_trace_enqueue_n(rt_tracer* t, uint32_t iterationCount, uint32_t samplesPerIteration, uint32_t n_steps);
int  rt_tracer_sync(rt_tracer* t);
/* One iteration of TraceFunct's loop as a building block for external drivers (the
 * multi-GPU progressive path, raytracertest_amd/dist.py): enqueue ONE trace launch of
 * `samples` spp.  clear_first != 0: the accumulators are cleared first (iteration 0,
 * RayTracerImpl.cu:242-243); emit_image != 0: the BGRA8 image is refreshed too (an update or
 * the last iteration, RayTracerImpl.cu:259-270,287-295).  No callbacks, no host sync. */
int  rt_tracer_launch(rt_tracer* t, uint32_t samples, int clear_first, int emit_image);
/* `iterations` consecutive iterations of the loop as ONE launch, bit-identical to `iterations` calls of
 * rt_tracer_launch (the per-iteration accumulate order is kept; state traffic and triangle classification
 * are paid once).  iterations <= rt_tracer_fused_iterations(t, samples) (>= 1; 1 when fusing is unavailable). */
int  rt_tracer_launch_iterations(rt_tracer* t, uint32_t samples, uint32_t iterations, int clear_first, int emit_image);
int  rt_tracer_fused_iterations(rt_tracer* t, uint32_t samples);
/* Small scenes keep each tile's candidate-triangle list (the result of the conservative classification, a
 * camera-dependent acceleration structure) in device memory.  across_traces != 0 (default): the lists stay
 * valid from one Trace to the next until the camera, the lens, the scene, the frame or the arithmetic mode
 * changes; 0: they are reused by the accumulating launches of one Trace only and every Trace classifies
 * afresh (what bench.py's headline figure uses).  Results are identical either way. */
int  rt_tracer_set_list_reuse(rt_tracer* t, int across_traces);
/* Second BGRA8 target of the emitting launches of rt_tracer_launch* / rt_tracer_trace_enqueue: a
 * device-visible buffer of at least rt_tracer_buffer_bytes(t, RT_BUF_IMAGE) bytes (device memory such
 * as a collective's send buffer, or pinned host memory) that the kernel writes alongside RT_BUF_IMAGE --
 * no copy afterwards.  NULL switches it off.  The caller orders its consumers behind the launch
 * (rt_tracer_stream) and keeps the buffer alive. */
int  rt_tracer_set_image_mirror(rt_tracer* t, void* device_visible_image);
/* Sum of the durations of the SAMPLED trace launches (HIP events on the tracer's stream, around
 * every 16th launch and every launch the caller waits for: an event pair costs ~5 us per launch) and
 * their number since the last reset; total_ms / launches = mean launch duration -- of a split launch
 * (see rt_tracer_stream_b) the upper half-frame kernel's, whose execution overlaps the lower half's.  reset_after != 0
 * clears both and makes the next launch a sampled one. */
int  rt_tracer_kernel_time(rt_tracer* t, double* total_ms, uint64_t* launches, int reset_after);
/* The same sampled launches by what they COST: a split launch counts from the start of its upper half to the end of the
 * LATER of its two halves (the lower half runs on rt_tracer_stream_b), an unsplit launch as above.  This is what the load
 * balancer uses (rt_tracer_rebalance, RowBandJob.rebalance): a band whose expensive rows lie in its lower half must not look
 * cheap.  Shares its sample set and its reset with rt_tracer_kernel_time. */
int  rt_tracer_launch_time(rt_tracer* t, double* total_ms, uint64_t* launches, int reset_after);
/* One instrumented launch (clear + trace of `samples` spp with counters; not a timed path).
 * Lane-level counters need RT_FLAG_NO_FILTER (reference-order path), wave-level ones the
 * default filtered path:
 *   out[0..3] ray-triangle tests by the reference's exit point: culled at det
 *             (Kernels.cuh:42), rejected at u (:51), rejected at v (:58), full hit (:63);
 *   out[4..7] (wave, triangle) pairs skipped by the __ballot early-outs after stage A
 *             (culling), B (u), C (v), and pairs that reached the exact stage D;
 *   out[8]    candidate triangles kept by the per-tile classification, summed over waves and
 *             rounds; out[9] classification rounds (one per wave when its list fits in LDS).
 *   out[10]   large scenes: candidate tests of a sample batch skipped by the per-sample forms; small scenes: sample
 *             batches of tiles that skipped their tests because the one candidate is certainly hit;
 *   out[11..15] small scenes: tiles without candidates / with 1 / with a certain winner (any list length) / with 2 /
 *             with more candidates. */
int  rt_tracer_trace_stats(rt_tracer* t, uint32_t samples, uint64_t out[16]);
/* Copy one of the tracer's device buffers to host memory / to another device pointer. */
int  rt_tracer_read_buffer(rt_tracer* t, int which, void* dst, size_t bytes);
int  rt_tracer_copy_buffer_to_device(rt_tracer* t, int which, void* dst_device, size_t bytes);
/* Same copy without the host synchronisation, and the tracer's HIP stream (a hipStream_t) so
 * that a driver can order its own work (e.g. the RCCL tile gather on another stream) behind it
 * with events instead of blocking the host. */
int  rt_tracer_copy_buffer_to_device_async(rt_tracer* t, int which, void* dst_device, size_t bytes);
void* rt_tracer_stream(rt_tracer* t);
/* Trace launches of frames of 128 rows or more run as two half-frame kernels, the upper half on
 * rt_tracer_stream, the lower half on this second stream (consecutive launches then overlap one half's
 * drain with the other half's work).  Every other entry point orders itself behind both; a driver that
 * orders its OWN work behind a launch without blocking either stream records one event on each and
 * waits for both (raytracertest_amd/dist.py), or calls rt_tracer_sync. */
void* rt_tracer_stream_b(rt_tracer* t);
void* rt_tracer_device_pointer(rt_tracer* t, int which);
size_t rt_tracer_buffer_bytes(rt_tracer* t, int which);
/* Launch geometry actually used: out[0]=K, out[1]=lds_chunk, out[2]=dynamic LDS bytes,
 * out[3]=grid.x, out[4]=grid.y, out[5]=n_tris, out[6]=n_spheres, out[7]=device. */
int  rt_tracer_info(rt_tracer* t, uint32_t out[8]);
const char* rt_tracer_last_error(rt_tracer* t);
const char* rt_last_error(void);          /* for failures before a tracer exists */
int  rt_device_count(void);
const char* rt_version(void);

/* ---- one frame sharded over several GPUs --------------------------------------
 * The reference builds ONE rt::RayTracer pinned to device 0 (OpenGLView/MainFrame.cpp:44-45,
 * OpenGLView/GLCanvas.cpp:259-260).  Pixels are independent and a pixel's RNG stream is keyed by its
 * global index (Random.cu:21-27), so the frame splits into contiguous row bands -- band k of n owns the rows
 * [k*H/n, (k+1)*H/n) -- that are traced with no exchange; the finished BGRA8 tiles are gathered to the
 * root (the device of band 0) with RCCL over xGMI (grouped ncclSend / ncclRecv; tiles of bands on the root
 * device are written in place by the trace kernel) and handed to the host from there. */

/* Same constructor, device list added: band k runs on devices[k] (ordinals may repeat: several bands per
 * device).  The handle is an rt_tracer like any other -- Trace / Stop / Resize / SetCameraParameters /
 * RotateCamera / UploadScene / callbacks keep their meaning, the callbacks receive the WHOLE frame (pinned
 * host memory) once per update, from one render thread, as in RayTracerImpl.cu:256-305; every device gets
 * its own host thread for the launches.  rt_tracer_read_buffer returns whole-frame buffers (RT_BUF_IMAGE =
 * the gathered frame), rt_tracer_trace_enqueue / rt_tracer_launch* include the gather,
 * rt_tracer_kernel_time reports the first band.  options->device / full_height / row_begin are not used
 * (full_height must be 0); results are bit-identical to a single tracer on the whole frame. */
int  rt_tracer_create_multi(const uint32_t imageSize[2], const float cameraPosition[3],
                            const float cameraAngles[2], float fov, float focalLength, float aperture,
                            const rt_options* options, const int32_t* devices, uint32_t n_bands,
                            rt_tracer** out);
/* One process per GPU instead (torch.distributed.run, MPI ...): every rank creates the tracer of ITS band
 * (rt_options.full_height / row_begin, rows of rank r of n = [r*H/n, (r+1)*H/n)) and joins the group with the
 * id rank 0 made and the launcher's own rendezvous distributed (collective: every rank calls it).  From then
 * on an emitting rt_tracer_trace_enqueue / rt_tracer_launch* of a member is followed by the gather of its
 * tile to rank 0, on a stream of its own, ordered by events; rt_tracer_sync waits for it as well, and rank 0
 * reads the gathered frame as RT_BUF_FRAME.  n_ranks == 1 needs no id. */
int  rt_group_unique_id(uint8_t id[RT_GROUP_ID_BYTES]);
int  rt_tracer_join_group(rt_tracer* t, uint32_t n_ranks, uint32_t rank, const uint8_t id[RT_GROUP_ID_BYTES]);
int  rt_tracer_leave_group(rt_tracer* t);
/* The same with an explicit partition: row_begin[0..n_ranks] ascending, row_begin[0] = 0, row_begin[n_ranks] =
 * full_height; rank r owns the rows [row_begin[r], row_begin[r+1]) (its tracer must have been created on, or
 * moved to -- rt_tracer_set_band -- exactly those rows). */
int  rt_tracer_join_group_bands(rt_tracer* t, uint32_t n_ranks, uint32_t rank, const uint8_t id[RT_GROUP_ID_BYTES],
                                const uint32_t* row_begin);
/* Load balance.  The reference scans every triangle for every ray, so equal rows are equal work there; with the
 * per-tile classification a band costs what its tiles' candidate lists cost, and equal rows leave the bands of a dense
 * scene uneven (C5 in 8 bands: mean/max = 0.83).  rt_balance_rows: from each band's measured cost, boundaries (multiples
 * of `granule` rows; the kernel's tiles are 8 rows high) that equalise it, assuming the cost is spread evenly inside a
 * band.  rt_tracer_rebalance applies it to a multi-device tracer from its bands' own kernel times since the last
 * rt_tracer_kernel_time reset; rt_tracer_set_band moves a band tracer of a multi-process job.  Like Resize, both
 * re-create the buffers and the RNG states of the bands (RayTracerImpl.cu:94-103); the image a Trace produces does not
 * depend on the partition. */
int  rt_balance_rows(uint32_t n_bands, const uint32_t* row_begin, const double* cost, uint32_t granule,
                     uint32_t* new_row_begin);
int  rt_tracer_rebalance(rt_tracer* t);
int  rt_tracer_set_band(rt_tracer* t, uint32_t row_begin, uint32_t rows);
/* Device time of the gathers since the last reset (root only; HIP events on the root's gather stream around
 * the exchange) and their number.  Zero for a frame whose bands all live on the root device. */
int  rt_tracer_gather_time(rt_tracer* t, double* total_ms, uint64_t* gathers, int reset_after);
/* One more exchange of the tiles as they are, without tracing: what the gather costs on its own (read it with
 * rt_tracer_gather_time).  A multi-device tracer or a group member (collective: every rank calls it). */
int  rt_tracer_gather_only(rt_tracer* t);
/* What the group is made of, as JSON text: transport ("rccl" | "peer" | "local" | "none"), ranks, the band -> rank map, the local
 * devices, and for an RCCL transport the library's version and per communicator its rank, ncclCommCount and device. */
int  rt_tracer_group_info(rt_tracer* t, char* json, size_t capacity);
/* Bands of the handle (1 for a plain tracer) and where band k runs: out = {device, first row, rows, rank}. */
int  rt_tracer_band_count(rt_tracer* t);
int  rt_tracer_band_info(rt_tracer* t, uint32_t band, uint32_t out[4]);

/* ---- single-function device harnesses (parity tests) ---------------------------------- */
/* n independent (ray, triangle) pairs through the device HitTriangle: rays n*6 (origin,
 * un-normalised direction -> rt::Ray(o, d, true)), tris n*9 (a, b, c).  eps_mode 0 =
 * kernel epsilon 1e-10f, 1 = unit-test epsilon FLT_EPSILON.  Outputs: hit n, tuv n*3,
 * normal n*3 = normalize(cross(b-a, c-a)), point n*3 = ray.point(t). */
int rt_dbg_hit_triangle(int device, uint32_t math_mode, uint32_t n, const float* rays,
                        const float* tris, int eps_mode, int32_t* hit, float* tuv, float* normal,
                        float* point);
int rt_dbg_sincos(int device, uint32_t n, const float* x, float* s, float* c);
/* 256-thread blocks of the default trace kernel (K = samples_in_flight) the occupancy API admits
 * per CU with lds_bytes of dynamic LDS (measurement aid). */
int rt_dbg_trace_occupancy(int device, int samples_in_flight, uint32_t lds_bytes);
/* fp32 VALU calibration on this device: attainable lane-FMA/s (8 fma chains per lane,
 * 8 waves per SIMD, every CU) and the shader clock held meanwhile.  Measurement aid only. */
int rt_dbg_valu_peak(int device, double* lane_fma_per_s, double* clock_ghz);
/* Exhaustive device check of the mid-range sqrt / reciprocal fast paths used by normalize: every float in
 * [2^-96, 2^96] against the generic correctly rounded expansions.  out = {values checked, sqrt mismatches,
 * reciprocal mismatches, bit pattern of a mismatching operand or 0}. */
int rt_dbg_check_midrange(int device, uint64_t out[4]);
/* Dense scenes: the header words (candidate count; 0xFFFFFFFF = the list overflowed its capacity and the tile tests its macro
 * tile's list) of the per-wave lists in HBM as the last launch left them; half 0 = an unsplit launch or the upper half of a split
 * one, 1 = the lower half.  Measurement aid and tests. */
int rt_dbg_wave_list_counts(rt_tracer* t, int half, uint32_t* dst, size_t capacity_tiles, uint32_t* n_tiles, uint32_t* capacity_per_tile);
/* The stored tile candidate lists of a small-scene tracer (after a launch that stored them): per 8x8 wave tile, in grid
 * order (4 per 32x8 block), *words_per_tile words: count | winner << 10 | certain-winner << 31, then the triangle indices. */
int rt_dbg_read_tile_lists(rt_tracer* t, uint32_t* dst, size_t capacity_words, uint32_t* words_per_tile);
/* The focal box each 8x8 wave tile of a trace launch classifies with (full tiles: the four corner pixels' focal points
 * widened by a curvature term; partial tiles: every in-image lane) and the focal points of the band's pixels as the rays
 * use them.  boxes: 8 floats per tile in grid order (4 per 32x8 block): lo[3], hi[3], corner path taken, usable;
 * focal: 3 floats per pixel.  curv_scale multiplies the curvature term for THIS launch only (1 = product; the test's
 * teeth: with 0 some pixel's focal point must fall outside its box). */
int rt_dbg_focal_boxes(rt_tracer* t, float curv_scale, float* boxes, size_t boxes_capacity, float* focal, size_t focal_capacity);
/* The conservative classification verdict by verdict (tests/test_gpu_classification.py, CLASSIFICATION.md): for each of
 * n_regions regions of the band -- level 0: the 8x8 wave tile at pixel (regions[2i], regions[2i+1]) (x a multiple of 8, band-local
 * row a multiple of 8), bounded exactly as a trace wave bounds it; level 1: the 32x8 block (x a multiple of 32), the union of
 * its four wave tiles; level 2: the 128x64 macro tile; level 3: the 32x16 region of the small scenes' two-level list builder
 * (x a multiple of 32, row a multiple of 16), the union of its eight tiles' boxes -- and for EVERY triangle of the scene, what tile_misses_triangle decides
 * and the interval ends it decides from, with every rounding allowance multiplied by slack_milli / 1000 (1000 = the product;
 * 300, 100, 30, 10, 0 exist so that the margin can be measured in the shipped library).
 *   out[region] = 16 floats: focal box lo[3], hi[3], lmin, lmax of |F - o|, usable (1) + 2 when the focal bounds of the list
 *   builder and of a large-scene trace wave (focal_bounds) agree bit for bit, + 4 when the host vouches for the corner bound (the
 *   two-level list builder is in use; level 3 is meaningless otherwise), lens radius A, orad[3], fc[3], followed by
 *   n_tris records.  forms == 0 (small-scene instantiation), 12 floats: flags (1 = kept, 2 = certainly hit by every ray of the
 *   family), det_lo, det_hi, U_lo, U_hi, V_lo, V_hi (bounds of det', U', V' = the reference's det, U = dot(tv, pv), V = dot(dir, qv)
 *   of Kernels.cuh:40,50,57 times |F - o|), q_lo, q_hi (bounds of t / |F - o|, t of Kernels.cuh:63), S_lo, S_hi (bounds of det' - U' - V'), 0.
 *   forms != 0 (large-scene instantiation with the per-sample forms), 32 floats: flags (1 = kept), the six ends, S_lo, S_hi, 3 x 0, then the
 *   forms {F1.c0, cx, cy, F2..., F3..., g1.xyz, g2.xyz, g3.xyz} each form scaled by its power of two, the gradients the fp16 values the kernel stores, 2 x 0. */
int rt_dbg_classify(rt_tracer* t, uint32_t level, uint32_t forms, uint32_t slack_milli, const uint32_t* regions, uint32_t n_regions,
                    float* out, size_t capacity_floats);
/* states n*6 {d,v0..v4} advanced in place, out n*m uniforms in (0,1] */
int rt_dbg_uniform(int device, uint32_t n, uint32_t m, uint32_t* states, float* out);
/* thin-lens rays of the tracer's current camera for n (x, y) pixels with given RNG states */
int rt_dbg_get_ray(rt_tracer* t, uint32_t n, const uint32_t* pixels, uint32_t* states, float* rays);
/* host: curand_init(seed, subsequence, 0) restated -> state[6] */
void rt_dbg_rng_init_host(uint64_t seed, uint64_t subsequence, uint32_t state[6]);

#ifdef __cplusplus
}
#endif
#endif /* RT_MI355X_H */
