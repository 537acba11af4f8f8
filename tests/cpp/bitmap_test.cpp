// Writes a tiny rt::Bitmap the way MainFrame.cpp:358-359 does; the pytest side compares the
// bytes with the Python writer and with the header layout of the reference's Common/Bitmap.h.
#include <vector>
#include "Common/Bitmap.h"
int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::vector<rt::Color> px{0xFF000000u, 0xFFFF0000u, 0xFF00FF00u, 0xFF0000FFu, 0x80123456u, 0x00000000u};
  rt::Bitmap bmp(math::uvec2(3, 2), px);
  bmp.SetPixel(2, 1, rt::GetColor(1, 2, 3, 4));
  if (bmp.GetPixel(2, 1) != 0x04010203u || bmp.Size().x != 3u) return 3;
  bmp.Write(argv[1]);
  rt::Bitmap filled(math::uvec2(2, 2), rt::Color(0xFFFFFFFFu));
  filled.Write(std::string(argv[1]) + ".filled");
  return 0;
}
