// rt_kernels.hpp -- launch interface between the host runtime (rt_tracer.cpp) and the
// gfx950 kernels (rt_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtk {

// Everything rt::TraceKernel receives by value (RayTracer/Kernels.cuh:110-118), for a row
// band [row0, row0+rows) of a W x H image.
constexpr uint32_t kSuperChunk = 1024u, kSuperMaxChunks = 64u;   // super tiles: triangles per chunk list, chunks per scene at most (rt_lists.hpp)

struct TraceParams {
  float4*   render;    // rows*W RGBA float accumulators (mRenderBuffer)
  uint32_t* counts;    // rows*W sample counts          (mSampleCountBuffer)
  uint32_t* rng;       // 6 planes of rows*W u32: d, v0..v4 (mRandomStates, SoA)
  uint32_t  W, H;      // full image size (camera uses the full height)
  uint32_t  row0, rows;
  uint32_t  npix;      // rows*W
  uint32_t  samples;   // sampleCount of this launch
  float     cam[12];   // mCameraTransformation: xyz of the 4 columns
  float     half_height, aspect, focal, aperture;
  // full 8x8 tiles bound their focal points from the four corner pixels (focal_bounds): tile_curv >= the distance, per
  // component, between a focal point of the tile and the bilinear interpolant of the corners' (host, params());
  // tile_round = the magnitude the rounding allowance of that bound scales with.  tile_curv <= 0: bound over all 64 lanes.
  float     tile_curv, tile_round;
  const float4* tri_a;      // 2 float4 per triangle: (e2.xyz, e1.x), (e1.yz, v0.xy); e1 = v1-v0, e2 = v2-v0
  const float*  tri_b;      // 1 float per triangle: v0.z
  const float4* tri_color;  // abs(normalize(cross(e1,e2)))
  uint32_t  n_tris;
  const float4* spheres;    // centre xyz, radius (build-defined extension)
  uint32_t  n_spheres;
  uint32_t  chunk;          // triangles staged into LDS at a time (full-scan path)
  uint32_t  bin_list;       // candidate records per wave in LDS (binned path), multiple of 64
  uint32_t  block_list;     // block-level pre-cull list (indices) in LDS; 0 = every wave scans the scene
  unsigned long long* stats; // null in the product path; 16 counters for the instrumented launch
  uint32_t* image;     // rows*W BGRA8, written when flags & TRACE_EMIT_IMAGE (mImageBuffer)
  const float4* tri_n; // 3 per triangle: unpacked vertex normals (edge-format scenes with smooth shading), or null
#ifdef RT_TIMELINE
  unsigned long long* timeline;   // experiment builds only (tools/timeline.py): 8 u64 per wave
#endif
  uint32_t* image_host;     // optional second BGRA8 target in pinned host memory (update hand-off), or null
  uint32_t  pretest_on;     // host decision: launches OR TRACE_PRETEST into flags (large-scene kernels)
  uint32_t  iters;          // fused launches: consecutive iterations of p.samples samples (>= 1)
  // Macro-tile triangle lists (scenes larger than the per-wave list): written by macro_bin_kernel
  // once per launch, read by the trace kernel's block-level pre-cull instead of the whole scene.
  // Per macro tile of macro_w x macro_h pixels: count (or 0xFFFFFFFF = overflow, scan the scene)
  // followed by macro_cap ascending triangle indices.  null = no macro level.
  uint32_t* macro_lists;
  uint32_t  macro_cap, macro_w, macro_h, macro_nx;
  // One level above (dense scenes): super tiles of super_f x super_f macro tiles, binned by super_bin_kernel against the whole
  // scene in chunks of 1 024 triangles (rt_lists.hpp); macro_bin_kernel then tests only its super tile's lists.  Per (super
  // tile, chunk): count, then up to 1 024 ascending indices.  null = the macro level scans the scene.
  uint32_t* super_lists;
  uint32_t  super_chunks, super_f, super_nx;
  float*    macro_bounds;   // with super_lists: 8 floats per macro tile -- focal box lo[3], hi[3], ok, any (macro_bounds_kernel)
  uint32_t* tile_lists; // small scenes: per wave tile count | winner << 10 | certain << 31, then bin_list triangle indices
                        // (written by tile_lists_kernel, read by the trace kernel); null: no lists (large scene, no triangles)
  // small scenes, split launches: this kernel traces every second block row of the band -- block row 2 * blockIdx.y + row_phase --
  // so that the two kernels of a launch cost the same whatever the picture (row_il = 0: the grid's rows are the band's)
  uint32_t  row_il, row_phase;
  // list builder (two-level): tiles that will generate rays, counted per half of the band -- [0]: block rows < cost_split_brow,
  // [1]: the others (device counters, published to the host by publish_half_cost; null: not counted)
  uint32_t* half_cost;
  uint32_t  cost_split_brow;
  // small scenes: per triangle, what a pixel of a certain-winner tile accumulates in one launch of p.samples samples --
  // {sum.x, sum.y, sum.z, bits of the BGRA8 word of a freshly cleared pixel} (sure_table_kernel; null: the kernel adds)
  const float4* sure_table;
  // dense scenes (the per-sample forms): the tiles' candidate lists in HBM, written by wave_lists_kernel and read by
  // dense_trace_kernel (rt_dense.hpp): per tile slot of the launch grid (1 + wave_cap) records of 16 dwords; null: the
  // trace kernel classifies on its own (instrumented launches, frames whose lists would not fit)
  uint32_t* wave_lists;
  uint32_t  wave_cap;
  uint32_t  flags;     // TRACE_*
};

// The accumulators are logically zero (first launch after ClearRenderBuffer/ClearSampleCountBuffer,
// RayTracerImpl.cu:242-243): skip their loads and the memsets; 0.0f + x is still evaluated.
constexpr uint32_t TRACE_ZERO_ACC = 1u;
// Also write the BGRA8 image from the updated accumulators (ConverterKernel fused, Kernels.cuh:149-169).
constexpr uint32_t TRACE_EMIT_IMAGE = 2u;
// Hit selection: keep the nearest t > 0 instead of the reference's farthest t (build-defined extension).
constexpr uint32_t TRACE_NEAREST_HIT = 4u;
// (8, 16: were TRACE_LISTS_STORE / TRACE_LISTS_LOAD while the small-scene trace kernels classified on their own; the lists
//  are now always built by tile_lists_kernel and always loaded)
// large-scene kernels: per-sample conservative forms per candidate (9 more floats per LDS record)
constexpr uint32_t TRACE_PRETEST = 32u;
// small-scene kernels: do not skip the intersection tests of tiles whose list is one certainly-hit triangle (A/B, tests)
constexpr uint32_t TRACE_NO_SURE_HIT = 64u;

// jump: J^(2^k), k < 32, 160 columns x 8 words; win: the 4-bit window tables of J^(2^m), m < 6 (rt_rng_host.hpp)
hipError_t launch_rng_init(uint32_t* rng, uint32_t npix, uint32_t p0, const uint32_t seeded[6],
                           const uint32_t* jump, const uint32_t* win, hipStream_t st);
hipError_t launch_prep_triangles(bool fma, bool edges, const float4* verts, uint32_t n, float4* tri_a, float* tri_b,
                                 float4* color, float4* normals, hipStream_t st);
uint32_t trace_lds_bytes(const TraceParams& p, bool bin);
// bin: per-tile triangle classification + per-wave LDS candidate lists (rt_trace.hpp);
// !bin: every ray scans the whole list, staged into LDS in chunks of p.chunk.
hipError_t launch_trace(const TraceParams& p, bool fma, bool filter, bool bin, int K, hipStream_t st);
int trace_occupancy(int K, size_t lds);
hipError_t launch_convert(const float4* render, const uint32_t* counts, uint32_t* image, uint32_t npix,
                          hipStream_t st);

hipError_t launch_dbg_hit_triangle(bool fma, uint32_t n, const float* rays, const float* tris, int eps_mode,
                                   int* hit, float* tuv, float* normal, float* point, hipStream_t st);
bool trace_can_fuse(bool filter, bool bin);      // launches with TraceParams::iters > 1 are available
hipError_t launch_macro_bin(const TraceParams& p, bool fma, hipStream_t st);
// the level above: p.super_lists (before launch_macro_bin)
hipError_t launch_super_bin(const TraceParams& p, bool fma, hipStream_t st);
// dense scenes: the per-wave candidate lists + forms of the (half-)launch `p` into p.wave_lists (after launch_macro_bin)
hipError_t launch_wave_lists(const TraceParams& p, bool fma, hipStream_t st);
// one wave that does nothing for `us` microseconds (bounded): the stagger of the first split launch after the tracer was idle
hipError_t launch_delay(uint32_t us, hipStream_t st);
// small scenes: hands the builder's per-half counts (TraceParams::half_cost) to the host -- *host_word = upper | lower << 32 --
// and clears them for the next build
hipError_t launch_publish_half_cost(uint32_t* half_cost, unsigned long long* host_word, hipStream_t st);
// small scenes: the per-triangle table TraceParams::sure_table for launches of `samples` samples
hipError_t launch_sure_table(const float4* colors, uint32_t n_tris, uint32_t samples, float4* out, hipStream_t st);
// small scenes: the tiles' candidate lists + certain-winner verdicts of the (half-)launch `p` into p.tile_lists
hipError_t launch_tile_lists(const TraceParams& p, bool fma, hipStream_t st);
hipError_t launch_dbg_check_midrange(unsigned long long* out, hipStream_t st);
hipError_t launch_dbg_focal_boxes(bool fma, const TraceParams& p, float* boxes, float* focal, hipStream_t st);
hipError_t launch_dbg_classify(bool fma, bool forms, uint32_t slack_milli, const TraceParams& p, uint32_t level, uint32_t n_regions,
                               const uint32_t* regions, float* out, hipStream_t st);
hipError_t launch_dbg_valu_peak(uint32_t blocks, int iters, float* out, unsigned long long* clk, hipStream_t st);
hipError_t launch_dbg_sincos(uint32_t n, const float* x, float* s, float* c, hipStream_t st);
hipError_t launch_dbg_uniform(uint32_t n, uint32_t m, uint32_t* states, float* out, hipStream_t st);
hipError_t launch_dbg_get_ray(bool fma, const TraceParams& p, uint32_t n, const uint32_t* pixels,
                              uint32_t* states, float* rays, hipStream_t st);

}  // namespace rtk
