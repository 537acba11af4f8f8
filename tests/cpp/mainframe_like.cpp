// Drives rt::RayTracer the way the reference's only caller does
// (OpenGLView/MainFrame.cpp:45,108-126,219-256,293,311,438), headless.  Prints one line the
// pytest side parses: update/finished counts, image size, an FNV-1a hash of the final image.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <memory>
#include <vector>

#include "RayTracer/RayTracer.h"
#include "RayTracer/RaytracerCallback.h"

struct Frame {                       // stands in for MainFrame
  std::atomic<int> updates{0}, finished{0};
  std::size_t lastSize = 0;
  uint64_t hash = 0;
  rt::RayTracer::uptr mRayTracer;

  void TracerUpdateCallback(rt::ColorPtr deviceImageBuffer, const std::size_t size) {
    (void)deviceImageBuffer; lastSize = size; ++updates;
  }
  void TracerFinishedCallback(rt::ColorPtr deviceImageBuffer, const std::size_t size) {
    uint64_t h = 1469598103934665603ull;
    for (std::size_t i = 0; i < size / sizeof(rt::Color); ++i) { h ^= deviceImageBuffer[i]; h *= 1099511628211ull; }
    hash = h; lastSize = size; ++finished;
  }
};

int main(int argc, char** argv) {
  const uint32_t seed = argc > 1 ? static_cast<uint32_t>(atoi(argv[1])) : 1u;
  const math::uvec2 imageSize(38, 21);                       // App.cpp:13
  const math::vec3 cameraPosition(0.0f, 0.0f, 0.0f);
  const math::vec2 cameraAngles(0.0f, 0.0f);
  const int bands = argc > 2 ? atoi(argv[2]) : 0;             // > 0: the frame in `bands` row bands, spread over the devices
  Frame f;
  if (bands > 0) {
    const int n_dev = rt_device_count() > 0 ? rt_device_count() : 1;
    std::vector<int> devices;
    for (int k = 0; k < bands; ++k) devices.push_back(k % n_dev);
    f.mRayTracer = std::make_unique<rt::RayTracer>(imageSize, cameraPosition, cameraAngles, 70.0f, 10.0f, 4.0f, devices);
  } else {
    f.mRayTracer = std::make_unique<rt::RayTracer>(imageSize, cameraPosition, cameraAngles, 70.0f, 10.0f, 4.0f);
  }
  if (!f.mRayTracer->Valid()) { std::printf("CREATE_FAILED %s\n", f.mRayTracer->LastError().c_str()); return 2; }
  f.mRayTracer->SetSeed(seed);
  f.mRayTracer->SetUpdateCallback(std::bind(&Frame::TracerUpdateCallback, &f, std::placeholders::_1, std::placeholders::_2));
  f.mRayTracer->SetFinishedCallback(std::bind(&Frame::TracerFinishedCallback, &f, std::placeholders::_1, std::placeholders::_2));
  std::vector<float4> hostData{make_float4(0.0f, 0.0f, 10.0f, 1.0f), make_float4(0.0f, 1.0f, 10.0f, 0.0f), make_float4(1.0f, 0.0f, 10.0f, 0.0f),
                               make_float4(1.0f, 0.0f, 10.0f, 0.0f), make_float4(0.0f, 1.0f, 10.0f, 1.0f), make_float4(1.0f, 1.0f, 10.0f, 0.0f),
                               make_float4(0.0f, 1.0f, 10.0f, 0.0f), make_float4(0.5f, 1.5f, 10.0f, 0.0f), make_float4(1.0f, 1.0f, 10.0f, 1.0f)};
  f.mRayTracer->UploadScene(hostData);                       // MainFrame.cpp:230-233
  f.mRayTracer->RotateCamera(math::vec2(0.0f, 3.0f));        // turn towards the demo triangles (z = +10)
  f.mRayTracer->SetCameraParameters(70.0f, 10.0f, 0.5f);     // MainFrame.cpp:249
  f.mRayTracer->Trace(100, 1, 10);                           // MainFrame.cpp:254: (iterationCount, sampleCount, updateInterval)
  const bool done = f.mRayTracer->Wait();
  std::vector<uint32_t> counts;
  f.mRayTracer->ReadSampleCounts(counts);
  std::printf("RESULT done=%d updates=%d finished=%d size=%zu hash=%llu count0=%u\n", done ? 1 : 0, f.updates.load(),
              f.finished.load(), f.lastSize, static_cast<unsigned long long>(f.hash), counts.empty() ? 0u : counts[0]);
  f.mRayTracer->Resize(math::uvec2(16, 9));                  // MainFrame.cpp:293
  f.mRayTracer->Trace(2, 2, 0);
  f.mRayTracer->Stop();                                      // MainFrame.cpp:311
  f.mRayTracer->Wait();
  return 0;
}
