// Common/NormalPacking.h -- corrected form of the reference's experimental normal packing
// (UnitTests/NormalPackingTest.cpp:10-23, Documentation/gpu.meshes.txt:20-33): three
// components in [-1,1] as 8-bit fields in the 24-bit fraction of one float, meant for the
// .w of the (v0, e0, e1) triangle rows.  Same function names as the reference's test code.
#pragma once
#include <cmath>

#include "Math.h"

namespace rt {

inline float pack(const math::vec3& normal) {
  return std::floor(normal.x * 127.0f + 127.5f) / 256.0f + std::floor(normal.y * 127.0f + 127.5f) / 65536.0f +
         std::floor(normal.z * 127.0f + 127.5f) / 16777216.0f;
}

inline math::vec3 unpack(const float packedNormal) {
  auto field = [packedNormal](float shift) {
    const float s = packedNormal * shift;
    return std::floor((s - std::floor(s)) * 256.0f) / 127.0f - 1.0f;
  };
  return math::vec3(field(1.0f), field(256.0f), field(65536.0f));   // the reference shifts by 1, 2^16, 2^24
}

}  // namespace rt
