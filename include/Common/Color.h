// Common/Color.h -- colour types of the public API (reference Common/Color.h:9-24).
// rt::Color is BGRA8 packed in a uint32: b | g << 8 | r << 16 | a << 24.
#pragma once
#include <cstdint>
#include <memory>

namespace rt {

using Channel = uint8_t;
using Color = uint32_t;
using ColorPtr = Color*;

// idx 0..3 = r, g, b, a
inline Channel GetComponent(const Color& color, const uint32_t& idx) {
  const uint32_t shift = idx == 0 ? 16u : idx == 1 ? 8u : idx == 2 ? 0u : 24u;
  return static_cast<Channel>((color >> shift) & 0xFFu);
}

inline Color GetColor(const Channel r = 0, const Channel g = 0, const Channel b = 0, const Channel a = 255) {
  return (static_cast<Color>(b) << 0) | (static_cast<Color>(g) << 8) | (static_cast<Color>(r) << 16) |
         (static_cast<Color>(a) << 24);
}

}  // namespace rt
