// PNG reader/writer for the two apps (zlib only): what the reference gets from cv::imread / cv::imwrite.
//   read_gray8   = cv::imread(path, 0)   -> 8-bit gray (colour converted, 16-bit reduced to 8)
//                  (apps/PhotoconsistencyFrameAlignment/PhotoconsistencyFrameAlignment.cpp:74,78;
//                   phovo/include/CImageReader.h:73-76)
//   read_unchanged16 = cv::imread(path, -1) of a 16-bit single-channel depth PNG (...:75,79; CImageReader.h:54-57)
// Supported: non-interlaced PNG, bit depth 8 or 16, colour types 0 (gray), 2 (RGB), 3 (palette, 8-bit),
// 4 (gray+alpha), 6 (RGBA).  Anything else is reported, not guessed.
#ifndef PHOVO_APPS_PNG_IO_H
#define PHOVO_APPS_PNG_IO_H

#include <cstdint>
#include <string>
#include <vector>

namespace phovo_io {

struct Image8 { int width = 0, height = 0; std::vector<uint8_t> pixels; };
struct Image16 { int width = 0, height = 0; std::vector<uint16_t> pixels; };

// Return false and fill `error` on failure.
bool read_gray8(const std::string &path, Image8 *out, std::string *error);
bool read_unchanged16(const std::string &path, Image16 *out, std::string *error);
bool write_gray8(const std::string &path, int width, int height, const uint8_t *pixels, std::string *error);
bool write_gray16(const std::string &path, int width, int height, const uint16_t *pixels, std::string *error);

}  // namespace phovo_io
#endif
