// Common/Bitmap.h -- rt::Bitmap, the 32-bpp top-down BI_BITFIELDS BMP writer the reference's
// "Save" button uses (reference Common/Bitmap.h:13-129, OpenGLView/MainFrame.cpp:314-366).
// Same class API (two constructors, SetPixel/GetPixel/Size/Write); the file it writes is
// byte-compatible: 14-byte file header + 36-byte image header + 88-byte colour header
// (138 bytes, packed), then W*H BGRA8 pixels top row first.
#pragma once
#include <cstdint>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "Color.h"
#include "Math.h"

namespace rt {

class Bitmap {
public:
  static constexpr uint32_t kHeaderBytes = 14u + 36u + 88u;

  Bitmap(const math::uvec2& size, const Color fillColor = 0x00000000)
      : mSize(size), mImageData(static_cast<size_t>(size.x) * size.y, fillColor) {}

  // takes the pixels over (the reference swaps the vector out of its argument, Bitmap.h:20-25)
  Bitmap(const math::uvec2& size, std::vector<Color>& pixelArray) : mSize(size), mImageData() {
    std::swap(mImageData, pixelArray);
  }

  void SetPixel(const uint32_t x, const uint32_t y, const Color& color) { mImageData[y * mSize.x + x] = color; }
  Color GetPixel(const uint32_t x, const uint32_t y) const { return mImageData[y * mSize.x + x]; }
  const math::uvec2& Size() const { return mSize; }

  // the 138 header bytes for a w x h image (little endian, no padding)
  static std::vector<uint8_t> Header(const int32_t w, const int32_t h) {
    std::vector<uint8_t> b;
    auto u16 = [&b](uint16_t v) { b.push_back(uint8_t(v)); b.push_back(uint8_t(v >> 8)); };
    auto u32 = [&b](uint32_t v) { for (int i = 0; i < 4; ++i) b.push_back(uint8_t(v >> (8 * i))); };
    const uint32_t pixels = 4u * static_cast<uint32_t>(w) * static_cast<uint32_t>(h);
    u16(0x4D42); u32(kHeaderBytes + pixels); u16(0); u16(0); u32(kHeaderBytes);          // file header
    u32(36u + 88u); u32(static_cast<uint32_t>(w)); u32(static_cast<uint32_t>(-h));      // image header: top-down
    u16(1); u16(32); u32(3); u32(0); u32(0); u32(0); u32(0);
    u32(0); u32(0x00FF0000); u32(0x0000FF00); u32(0x000000FF); u32(0xFF000000);         // colour header: masks
    u32(0x73524742);                                                                   // "sRGB"
    for (int i = 0; i < 16; ++i) u32(0);
    return b;
  }

  void Write(const std::string& path) const {
    std::ofstream ostream(path, std::ios::out | std::ios::binary);
    if (!ostream.good()) throw std::runtime_error(std::string("cannot save file to path: \"") + path + "\"");
    const std::vector<uint8_t> header = Header(static_cast<int32_t>(mSize.x), static_cast<int32_t>(mSize.y));
    ostream.write(reinterpret_cast<const char*>(header.data()), static_cast<std::streamsize>(header.size()));
    if (!mImageData.empty())
      ostream.write(reinterpret_cast<const char*>(mImageData.data()),
                    static_cast<std::streamsize>(sizeof(Color) * mImageData.size()));
  }

private:
  math::uvec2 mSize;
  std::vector<Color> mImageData;
};

}  // namespace rt
