// RayTracer/RaytracerCallback.h -- rt::CallBackFunction (reference RaytracerCallback.h:9).
// `size` is in bytes (W*H*4, RayTracerImpl.cu:272,304).  Declared change: the image pointer
// is HOST-readable (the OpenGL PBO interop is cut); the parameter keeps its reference name.
#pragma once
#include <cstddef>
#include <functional>

#include "../Common/Color.h"

namespace rt {
using CallBackFunction = std::function<void(rt::ColorPtr deviceImageBuffer, const std::size_t size)>;
}
