"""32-bpp top-down BI_BITFIELDS BMP, byte-compatible with the reference's rt::Bitmap
(Common/Bitmap.h:45-123: 14-byte file header + 36-byte image header + 88-byte colour
header, packed, then W*H BGRA8 pixels, top row first; used by MainFrame.cpp:314-366)."""
import struct

import numpy as np

HEADER_BYTES = 14 + 36 + 88


def bmp_header(width, height):
    pixels = 4 * width * height
    file_header = struct.pack("<HIHHI", 0x4D42, HEADER_BYTES + pixels, 0, 0, HEADER_BYTES)
    image_header = struct.pack("<IiiHHIIiiI", 36 + 88, width, -height, 1, 32, 3, 0, 0, 0, 0)
    color_header = struct.pack("<IIIIII", 0, 0x00FF0000, 0x0000FF00, 0x000000FF, 0xFF000000, 0x73524742) + bytes(64)
    out = file_header + image_header + color_header
    assert len(out) == HEADER_BYTES
    return out


def write_bmp(path, image):
    """image: (H, W) uint32 BGRA8-in-u32 (rt::Color), row 0 = top row."""
    img = np.ascontiguousarray(image, dtype="<u4")
    h, w = img.shape
    with open(path, "wb") as f:
        f.write(bmp_header(w, h))
        f.write(img.tobytes())


def read_bmp(path):
    """Inverse of write_bmp (only this exact format)."""
    data = open(path, "rb").read()
    sig, size, _, _, off = struct.unpack_from("<HIHHI", data, 0)
    hsz, w, h, planes, bpp, comp = struct.unpack_from("<IiiHHI", data, 14)
    if (sig, off, hsz, planes, bpp, comp) != (0x4D42, HEADER_BYTES, 124, 1, 32, 3) or h >= 0 or size != len(data):
        raise ValueError("not a reference-format BMP")
    return np.frombuffer(data, dtype="<u4", offset=off).reshape(-h, w).copy()
