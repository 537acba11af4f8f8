// rt_cli -- headless front end of the trace path: the command line of the reference's wx
// app (OpenGLView/App.cpp:62-184: -w -h -s -i -u -cx -cy -cz -cxa -cya -f -l -a, integer
// values, defaults App.cpp:11-23) without the GUI, plus what a batch run needs: a scene,
// a seed and an output file written with the reference's BMP format (Common/Bitmap.h).
//
//   rt_cli -w 1920 -h 1080 -s 16 -i 1 -f 70 -l 3 --aperture 0.05 --scene cornell32 --seed 1 -o out.bmp
//
// Build: g++ -std=c++17 -Iinclude tools/rt_cli.cpp -Lraytracertest_amd/lib -lrt_mi355x -o rt_cli
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "Common/Bitmap.h"
#include "RayTracer/RayTracer.h"

namespace {

struct Args {
  uint32_t w = 3840 / 100, h = 2160 / 100;      // App.cpp:13
  uint32_t samples = 1, iterations = 100, update = 10;
  float cx = 0, cy = 0, cz = 0, cxa = 0, cya = 0, fov = 70.0f, focal = 10.0f, aperture = 4.0f;
  uint64_t seed = 0; bool have_seed = false;
  std::string scene = "demo3", scene_file, out = "image0.bmp";
  bool quiet = false, edges = false, smooth = false, nearest = false;
};

std::vector<float4> demo3() {                   // MainFrame.cpp:230-232
  return {make_float4(0, 0, 10, 1), make_float4(0, 1, 10, 0), make_float4(1, 0, 10, 0),
          make_float4(1, 0, 10, 0), make_float4(0, 1, 10, 1), make_float4(1, 1, 10, 0),
          make_float4(0, 1, 10, 0), make_float4(0.5f, 1.5f, 10, 0), make_float4(1, 1, 10, 1)};
}

// raw little-endian float32 x,y,z,w records, 3 per triangle (what UploadScene takes)
std::vector<float4> load_f4(const std::string& path) {
  std::ifstream in(path, std::ios::binary | std::ios::ate);
  std::vector<float4> v;
  if (!in.good()) return v;
  const std::streamsize n = in.tellg();
  in.seekg(0);
  v.resize(static_cast<size_t>(n) / sizeof(float4));
  in.read(reinterpret_cast<char*>(v.data()), static_cast<std::streamsize>(v.size() * sizeof(float4)));
  return v;
}

void usage() {
  std::puts("rt_cli [-w W] [-h H] [-s samples] [-i iterations] [-u updateInterval] [-cx N -cy N -cz N]\n"
            "       [-cxa deg] [-cya deg] [-f fovDeg] [-l focalLength] [-a aperture]      (reference flags, integers)\n"
            "       [--focal F] [--aperture A] [--fov F]                                  (float forms)\n"
            "       [--scene demo3|<file.f4>] [--seed N] [-o out.bmp] [-q]\n"
            "       [--edges] (file holds (v0,e0,e1) rows, packed vertex normals in .w)  [--smooth] [--nearest]");
}

}  // namespace

int main(int argc, char** argv) {
  Args a;
  for (int i = 1; i < argc; ++i) {
    const std::string k = argv[i];
    auto next = [&](const char* what) -> const char* {
      if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", what); std::exit(2); }
      return argv[++i];
    };
    if (k == "-w") a.w = static_cast<uint32_t>(std::atol(next("-w")));
    else if (k == "-h") a.h = static_cast<uint32_t>(std::atol(next("-h")));
    else if (k == "-s") a.samples = static_cast<uint32_t>(std::atol(next("-s")));
    else if (k == "-i") a.iterations = static_cast<uint32_t>(std::atol(next("-i")));
    else if (k == "-u") a.update = static_cast<uint32_t>(std::atol(next("-u")));
    else if (k == "-cx") a.cx = static_cast<float>(std::atol(next("-cx")));
    else if (k == "-cy") a.cy = static_cast<float>(std::atol(next("-cy")));
    else if (k == "-cz") a.cz = static_cast<float>(std::atol(next("-cz")));
    else if (k == "-cxa") a.cxa = static_cast<float>(std::atol(next("-cxa"))) * 0.01745329251994329576923690768489f;  // glm::radians, App.cpp:148
    else if (k == "-cya") a.cya = static_cast<float>(std::atol(next("-cya"))) * 0.01745329251994329576923690768489f;
    else if (k == "-f") a.fov = static_cast<float>(std::atol(next("-f")));
    else if (k == "-l") a.focal = static_cast<float>(std::atol(next("-l")));
    else if (k == "-a") a.aperture = static_cast<float>(std::atol(next("-a")));
    else if (k == "--fov") a.fov = std::strtof(next("--fov"), nullptr);
    else if (k == "--focal") a.focal = std::strtof(next("--focal"), nullptr);
    else if (k == "--aperture") a.aperture = std::strtof(next("--aperture"), nullptr);
    else if (k == "--cxa-rad") a.cxa = std::strtof(next("--cxa-rad"), nullptr);
    else if (k == "--cya-rad") a.cya = std::strtof(next("--cya-rad"), nullptr);
    else if (k == "--scene") a.scene = next("--scene");
    else if (k == "--seed") { a.seed = std::strtoull(next("--seed"), nullptr, 10); a.have_seed = true; }
    else if (k == "--edges") a.edges = true;
    else if (k == "--smooth") a.smooth = true;
    else if (k == "--nearest") a.nearest = true;
    else if (k == "-o") a.out = next("-o");
    else if (k == "-q") a.quiet = true;
    else if (k == "-v") a.quiet = false;
    else if (k == "--help") { usage(); return 0; }
    else { std::fprintf(stderr, "unknown option %s\n", k.c_str()); usage(); return 2; }
  }

  rt_options opt;
  std::memset(&opt, 0, sizeof opt);
  opt.struct_size = sizeof opt;
  opt.use_time_seed = 1;                                       // Random.cu:45 unless --seed is given
  opt.flags = (a.smooth ? RT_FLAG_SMOOTH_NORMALS : 0u) | (a.nearest ? RT_FLAG_NEAREST_HIT : 0u);
  rt::RayTracer tracer(math::uvec2(a.w, a.h), math::vec3(a.cx, a.cy, a.cz), math::vec2(a.cxa, a.cya), a.fov, a.focal,
                       a.aperture, &opt);
  if (!tracer.Valid()) { std::fprintf(stderr, "rt_cli: %s\n", tracer.LastError().c_str()); return 1; }
  if (a.have_seed) tracer.SetSeed(a.seed);

  std::vector<float4> scene = (a.scene == "demo3") ? demo3() : load_f4(a.scene);
  if (scene.size() < 3 || scene.size() % 3 != 0) {
    std::fprintf(stderr, "rt_cli: scene '%s' has %zu float4 (need a positive multiple of 3)\n", a.scene.c_str(), scene.size());
    return 1;
  }
  if (a.edges) tracer.UploadSceneEdges(scene);
  else tracer.UploadScene(scene);

  uint32_t updates = 0;
  std::vector<rt::Color> finalImage;
  tracer.SetUpdateCallback([&](rt::ColorPtr, const std::size_t) { ++updates; });
  tracer.SetFinishedCallback([&](rt::ColorPtr image, const std::size_t size) {
    finalImage.assign(image, image + size / sizeof(rt::Color));
  });
  const auto t0 = std::chrono::steady_clock::now();
  tracer.Trace(a.iterations, a.samples, a.update);          // MainFrame.cpp:254-256
  const bool done = tracer.Wait();
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (!done || finalImage.size() != static_cast<size_t>(a.w) * a.h) {
    std::fprintf(stderr, "rt_cli: trace did not finish: %s\n", tracer.LastError().c_str());
    return 1;
  }
  try {
    rt::Bitmap bmp(math::uvec2(a.w, a.h), finalImage);      // MainFrame.cpp:358-359
    bmp.Write(a.out);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "rt_cli: %s\n", e.what());
    return 1;
  }
  if (!a.quiet) {
    const double rays = double(a.w) * a.h * a.iterations * a.samples;
    std::printf("%ux%u, %u x %u spp, %zu triangles, %u updates: %.2f ms end to end (%.1f Mray/s incl. host hand-off) -> %s\n",
                a.w, a.h, a.iterations, a.samples, scene.size() / 3, updates, ms, rays / ms / 1e3, a.out.c_str());
  }
  return 0;
}
