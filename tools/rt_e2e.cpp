// tools/rt_e2e.cpp -- what the reference's only caller feels: the latency of a blocking Trace(1, S, 0) through the C++ class
// (include/RayTracer/RayTracer.h), finished callback with the BGRA8 image in host memory, as OpenGLView/MainFrame.cpp's drag
// loop issues it (MainFrame.cpp:394-444: one Trace per mouse-move event; RayTracerImpl.cu:69-87,236-315).  SURVEY 8d: "reported
// separately as end-to-end" -- never the headline.  Plain g++ against the C ABI, built to raytracertest_amd/lib/rt_e2e.
//
//   rt_e2e W H samples tris.bin [traces] [fov focal aperture]     tris.bin: 3 float4 per triangle (what UploadScene takes)
// Prints one JSON object.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "RayTracer/RayTracer.h"

using clk = std::chrono::steady_clock;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }

int main(int argc, char** argv) {
  if (argc < 5) { std::fprintf(stderr, "usage: rt_e2e W H samples tris.bin [traces] [fov focal aperture]\n"); return 2; }
  const uint32_t W = static_cast<uint32_t>(atoi(argv[1])), H = static_cast<uint32_t>(atoi(argv[2])), S = static_cast<uint32_t>(atoi(argv[3]));
  const int traces = argc > 5 ? atoi(argv[5]) : 200;
  const float fov = argc > 8 ? static_cast<float>(atof(argv[6])) : 70.0f, focal = argc > 8 ? static_cast<float>(atof(argv[7])) : 3.0f,
              aperture = argc > 8 ? static_cast<float>(atof(argv[8])) : 0.05f;
  std::vector<float4> tris;
  if (FILE* f = std::fopen(argv[4], "rb")) {
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    tris.resize(static_cast<size_t>(n) / sizeof(float4));
    if (std::fread(tris.data(), sizeof(float4), tris.size(), f) != tris.size()) { std::fprintf(stderr, "short read\n"); return 2; }
    std::fclose(f);
  } else { std::fprintf(stderr, "cannot read %s\n", argv[4]); return 2; }

  rt_options opt;
  std::memset(&opt, 0, sizeof opt);
  opt.struct_size = sizeof opt;
  opt.seed = 1;
  rt::RayTracer tracer(math::uvec2(W, H), math::vec3(0.0f, 0.0f, 0.0f), math::vec2(0.0f, 0.0f), fov, focal, aperture, &opt);
  if (!tracer.Valid()) { std::printf("{\"error\": \"%s\"}\n", tracer.LastError().c_str()); return 3; }
  tracer.UploadScene(tris);

  clk::time_point t_cb;
  std::atomic<int> fired{0};
  uint64_t sink = 0;
  size_t cb_bytes = 0;
  tracer.SetFinishedCallback([&](rt::ColorPtr image, const std::size_t size) {
    t_cb = clk::now();                                    // the image is in host memory when the callback runs
    sink += image[0] + image[size / sizeof(rt::Color) / 2] + image[size / sizeof(rt::Color) - 1];
    cb_bytes = size;
    ++fired;
  });
  for (int i = 0; i < 20; ++i) { tracer.Trace(1, S, 0); tracer.Wait(); }       // warm: lists built, clocks up, render thread exists
  std::vector<double> to_cb, to_wait;
  const clk::time_point run0 = clk::now();
  for (int i = 0; i < traces; ++i) {
    const clk::time_point t0 = clk::now();
    tracer.Trace(1, S, 0);
    const bool done = tracer.Wait();
    const clk::time_point t1 = clk::now();
    if (!done) { std::printf("{\"error\": \"trace %d did not finish: %s\"}\n", i, tracer.LastError().c_str()); return 3; }
    to_cb.push_back(us(t0, t_cb));
    to_wait.push_back(us(t0, t1));
  }
  const double loop_us = us(run0, clk::now()) / traces;
  // the device part alone: the same emitting launch without the host image (rt_tracer_launch + rt_tracer_sync)
  rt_tracer* h = tracer.Handle();
  for (int i = 0; i < 10; ++i) { rt_tracer_launch(h, S, 1, 1); }
  rt_tracer_sync(h);
  std::vector<double> dev;
  for (int i = 0; i < 50; ++i) {
    const clk::time_point t0 = clk::now();
    rt_tracer_launch(h, S, 1, 1);
    rt_tracer_sync(h);
    dev.push_back(us(t0, clk::now()));
  }
  // full checksum of the last image, outside the timed loops: the bytes really are there
  uint64_t fnv = 1469598103934665603ull;
  tracer.SetFinishedCallback([&](rt::ColorPtr image, const std::size_t size) {
    for (std::size_t i = 0; i < size / sizeof(rt::Color); ++i) { fnv ^= image[i]; fnv *= 1099511628211ull; }
  });
  tracer.Trace(1, S, 0);
  tracer.Wait();
  auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  auto p10 = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 10]; };
  auto p90 = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() * 9 / 10]; };
  std::printf("{\"us\": %.1f, \"us_p10\": %.1f, \"us_p90\": %.1f, \"to_callback_us\": %.1f, \"loop_us_per_trace\": %.1f, "
              "\"launch_and_sync_without_host_image_us\": %.1f, \"traces\": %d, \"callbacks\": %d, \"image_bytes\": %zu, "
              "\"image_fnv1a\": \"%016llx\", \"sink\": %llu}\n",
              med(to_wait), p10(to_wait), p90(to_wait), med(to_cb), loop_us, med(dev), traces, fired.load(), cb_bytes,
              static_cast<unsigned long long>(fnv), static_cast<unsigned long long>(sink & 0xffff));
  return 0;
}
