// RayTracer/RayTracer.h -- source-compatible rt::RayTracer (reference RayTracer/RayTracer.h:14-41).
//
// Same class name, base, public methods, parameter order, types and units as the reference,
// implemented header-only over the C ABI of librt_mi355x.so (../rt_mi355x.h): link with
// -lrt_mi355x.  Nothing throws (the reference swallows every failure, RayTracerImpl.cu:42-45,
// 307-314); LastError() tells what went wrong.  Methods below the marker are additive.
#pragma once
#include <cstdint>
#include <mutex>
#include <string>
#include <vector>

#include "../Common/Color.h"
#include "../Common/Math.h"
#include "../Common/Sptr.h"
#include "../rt_mi355x.h"
#include "RaytracerCallback.h"

namespace rt {

class RayTracer : public ISptr<RayTracer> {
public:
  RayTracer(const math::uvec2& imageSize, const math::vec3& cameraPosition, const math::vec2& cameraAngles,
            const float fov, const float focalLength, const float aperture)
      : RayTracer(imageSize, cameraPosition, cameraAngles, fov, focalLength, aperture, nullptr) {}

  ~RayTracer() { rt_tracer_destroy(mImpl); }
  RayTracer(const RayTracer&) = delete;
  RayTracer& operator=(const RayTracer&) = delete;

  void Trace(const uint32_t iterationCount, const uint32_t samplesPerIteration, const uint32_t updateInterval) {
    if (mImpl) rt_tracer_trace(mImpl, iterationCount, samplesPerIteration, updateInterval);
  }
  void Stop() { rt_tracer_stop(mImpl); }
  void Resize(const math::uvec2& size) {
    const uint32_t s[2] = {size.x, size.y};
    if (mImpl) rt_tracer_resize(mImpl, s);
  }
  void SetCameraParameters(const float fov, const float focalLength, const float aperture) {
    rt_tracer_set_camera_parameters(mImpl, fov, focalLength, aperture);
  }
  void RotateCamera(const math::vec2& angles) {
    const float a[2] = {angles.x, angles.y};
    rt_tracer_rotate_camera(mImpl, a);
  }
  void UploadScene(const std::vector<float4>& hostData) {
    static_assert(sizeof(float4) == sizeof(rt_float4), "float4 layout");
    if (mImpl) rt_tracer_upload_scene(mImpl, reinterpret_cast<const rt_float4*>(hostData.data()), hostData.size());
  }
  // (the render thread may be inside OnUpdate/OnFinished meanwhile: the std::function is swapped under a
  //  mutex and the trampolines call a copy)
  void SetUpdateCallback(rt::CallBackFunction callback) {
    const bool on = static_cast<bool>(callback);
    { std::lock_guard<std::mutex> lk(mCallbackMutex); mUpdate = std::move(callback); }
    rt_tracer_set_update_callback(mImpl, on ? &RayTracer::OnUpdate : nullptr, this);
  }
  void SetFinishedCallback(rt::CallBackFunction callback) {
    const bool on = static_cast<bool>(callback);
    { std::lock_guard<std::mutex> lk(mCallbackMutex); mFinished = std::move(callback); }
    rt_tracer_set_finished_callback(mImpl, on ? &RayTracer::OnFinished : nullptr, this);
  }

  // ---- additive extensions (not in the reference) ---------------------------------------
  RayTracer(const math::uvec2& imageSize, const math::vec3& cameraPosition, const math::vec2& cameraAngles,
            const float fov, const float focalLength, const float aperture, const rt_options* options)
      : mImpl(nullptr) {
    const uint32_t size[2] = {imageSize.x, imageSize.y};
    const float pos[3] = {cameraPosition.x, cameraPosition.y, cameraPosition.z};
    const float ang[2] = {cameraAngles.x, cameraAngles.y};
    rt_tracer_create_ex(size, pos, ang, fov, focalLength, aperture, options, &mImpl);
  }
  // The frame sharded in row bands over several GPUs of this process, band k on devices[k] (the reference
  // pins device 0, OpenGLView/GLCanvas.cpp:259-260); same methods, the callbacks receive the whole frame.
  RayTracer(const math::uvec2& imageSize, const math::vec3& cameraPosition, const math::vec2& cameraAngles,
            const float fov, const float focalLength, const float aperture, const std::vector<int>& devices,
            const rt_options* options = nullptr)
      : mImpl(nullptr) {
    const uint32_t size[2] = {imageSize.x, imageSize.y};
    const float pos[3] = {cameraPosition.x, cameraPosition.y, cameraPosition.z};
    const float ang[2] = {cameraAngles.x, cameraAngles.y};
    const std::vector<int32_t> devs(devices.begin(), devices.end());
    rt_tracer_create_multi(size, pos, ang, fov, focalLength, aperture, options, devs.data(),
                           static_cast<uint32_t>(devs.size()), &mImpl);
  }
  bool Valid() const { return mImpl != nullptr; }
  bool Wait() { return mImpl && rt_tracer_wait(mImpl) == 1; }
  void SetSeed(uint64_t seed) { if (mImpl) rt_tracer_set_seed(mImpl, seed); }
  void UploadSpheres(const std::vector<float4>& spheres) {
    if (mImpl) rt_tracer_upload_spheres(mImpl, reinterpret_cast<const rt_float4*>(spheres.data()), spheres.size());
  }
  // (v0, e0 = v1-v0, e1 = v2-v0) rows with packed vertex normals in .w (Documentation/gpu.meshes.txt:16-34;
  // pack with rt::pack, Common/NormalPacking.h); smooth shading needs RT_FLAG_SMOOTH_NORMALS in rt_options
  void UploadSceneEdges(const std::vector<float4>& hostData) {
    if (mImpl) rt_tracer_upload_scene_edges(mImpl, reinterpret_cast<const rt_float4*>(hostData.data()), hostData.size());
  }
  bool ReadRenderBuffer(std::vector<float>& rgba) {
    if (!mImpl) return false;
    rgba.resize(rt_tracer_buffer_bytes(mImpl, RT_BUF_RENDER) / sizeof(float));
    return rt_tracer_read_buffer(mImpl, RT_BUF_RENDER, rgba.data(), rgba.size() * sizeof(float)) == RT_OK;
  }
  bool ReadSampleCounts(std::vector<uint32_t>& counts) {
    if (!mImpl) return false;
    counts.resize(rt_tracer_buffer_bytes(mImpl, RT_BUF_COUNTS) / sizeof(uint32_t));
    return rt_tracer_read_buffer(mImpl, RT_BUF_COUNTS, counts.data(), counts.size() * sizeof(uint32_t)) == RT_OK;
  }
  std::string LastError() const { return mImpl ? rt_tracer_last_error(mImpl) : rt_last_error(); }
  rt_tracer* Handle() const { return mImpl; }

private:
  static void OnUpdate(uint32_t* image, size_t size, void* self) {
    RayTracer* const me = static_cast<RayTracer*>(self);
    rt::CallBackFunction f;
    { std::lock_guard<std::mutex> lk(me->mCallbackMutex); f = me->mUpdate; }
    if (f) f(image, size);
  }
  static void OnFinished(uint32_t* image, size_t size, void* self) {
    RayTracer* const me = static_cast<RayTracer*>(self);
    rt::CallBackFunction f;
    { std::lock_guard<std::mutex> lk(me->mCallbackMutex); f = me->mFinished; }
    if (f) f(image, size);
  }

  rt_tracer* mImpl;
  std::mutex mCallbackMutex;
  rt::CallBackFunction mUpdate, mFinished;
};

}  // namespace rt
