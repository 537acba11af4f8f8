// Common/Math.h -- the math types rt::RayTracer's signatures mention (reference
// Common/Math.h:13-31 aliases glm::uvec2/vec2/vec3).  glm is not a dependency of this
// build: with RT_USE_GLM defined the glm types are aliased exactly like the reference,
// otherwise minimal stand-alone structs with the same member names (.x .y .z, operator[])
// are provided so that callers such as OpenGLView/MainFrame.cpp:45,293,438 compile unchanged.
#pragma once
#include <cstddef>
#include <cstdint>

#ifdef RT_USE_GLM
#include <glm/glm.hpp>
namespace math {
using uvec2 = glm::uvec2;
using vec2 = glm::vec2;
using vec3 = glm::vec3;
}
#else
namespace math {

template <class T>
struct tvec2 {
  T x, y;
  constexpr tvec2() : x(0), y(0) {}
  constexpr explicit tvec2(T s) : x(s), y(s) {}
  constexpr tvec2(T x_, T y_) : x(x_), y(y_) {}
  T& operator[](std::size_t i) { return i == 0 ? x : y; }
  const T& operator[](std::size_t i) const { return i == 0 ? x : y; }
  tvec2& operator+=(const tvec2& o) { x += o.x; y += o.y; return *this; }
  friend constexpr bool operator==(const tvec2& a, const tvec2& b) { return a.x == b.x && a.y == b.y; }
};

template <class T>
struct tvec3 {
  T x, y, z;
  constexpr tvec3() : x(0), y(0), z(0) {}
  constexpr explicit tvec3(T s) : x(s), y(s), z(s) {}
  constexpr tvec3(T x_, T y_, T z_) : x(x_), y(y_), z(z_) {}
  T& operator[](std::size_t i) { return i == 0 ? x : i == 1 ? y : z; }
  const T& operator[](std::size_t i) const { return i == 0 ? x : i == 1 ? y : z; }
  friend constexpr bool operator==(const tvec3& a, const tvec3& b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
};

using uvec2 = tvec2<uint32_t>;
using vec2 = tvec2<float>;
using vec3 = tvec3<float>;

constexpr float cPi = 3.141592653589793f;

}  // namespace math
#endif

// CUDA's float4 / make_float4 as used by RayTracer::UploadScene (RayTracer.h:34) and its
// caller (MainFrame.cpp:230-232).  Skipped when HIP's vector types are already in scope.
#if !defined(HIP_INCLUDE_HIP_AMD_DETAIL_HIP_VECTOR_TYPES_H) && !defined(RT_HAVE_FLOAT4)
#define RT_HAVE_FLOAT4
struct float4 { float x, y, z, w; };
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
#endif
