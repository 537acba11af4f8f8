// RayTracer/RaytracerCallback.h -- the callback type of the public API.
//
// rt::CallBackFunction keeps the reference's signature (RaytracerCallback.h:9): an image
// pointer and its size in BYTES (W*H*4, RayTracerImpl.cu:272,304).  Declared change: the
// pointer is HOST-readable BGRA8 (pinned memory owned by the tracer) because the OpenGL PBO
// interop is cut; the parameter keeps its reference name so existing handlers compile.
// Handlers run on the tracer's render thread.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>

#include "../Common/Color.h"

namespace rt {

using CallBackFunction = std::function<void(rt::ColorPtr deviceImageBuffer, const std::size_t size)>;

// Number of pixels a callback's `size` argument stands for.
inline std::size_t PixelCount(const std::size_t sizeInBytes) { return sizeInBytes / sizeof(rt::Color); }

}  // namespace rt
