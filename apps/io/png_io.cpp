#include "png_io.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace phovo_io {

namespace {

struct Raw {
  int width = 0, height = 0, bit_depth = 0, color_type = 0, channels = 0;
  std::vector<uint8_t> rows;        // unfiltered scanlines, big-endian samples
  std::vector<uint8_t> palette;     // RGB triples
};

uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

bool load(const std::string &path, Raw *raw, std::string *err)
{
  FILE *f = std::fopen(path.c_str(), "rb");
  if (!f) { *err = "cannot open " + path; return false; }
  std::vector<uint8_t> buf;
  uint8_t tmp[65536];
  size_t n;
  while ((n = std::fread(tmp, 1, sizeof(tmp), f)) > 0) {
    buf.insert(buf.end(), tmp, tmp + n);
    if (n < sizeof(tmp)) break;               // end of file or error: do not read again
  }
  const bool read_error = std::ferror(f) != 0;
  std::fclose(f);
  if (read_error) { *err = path + ": read error"; return false; }
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (buf.size() < 8 || std::memcmp(buf.data(), sig, 8) != 0) { *err = path + ": not a PNG file"; return false; }
  std::vector<uint8_t> idat;
  size_t pos = 8;
  bool have_ihdr = false, interlaced = false;
  while (pos + 12 <= buf.size()) {
    const uint32_t len = be32(&buf[pos]);
    const char *type = reinterpret_cast<const char *>(&buf[pos + 4]);
    if (pos + 12 + (size_t)len > buf.size()) { *err = path + ": truncated chunk"; return false; }
    const uint8_t *data = &buf[pos + 8];
    if (!std::memcmp(type, "IHDR", 4)) {
      if (len < 13) { *err = path + ": bad IHDR"; return false; }
      raw->width = (int)be32(data); raw->height = (int)be32(data + 4);
      raw->bit_depth = data[8]; raw->color_type = data[9];
      interlaced = data[12] != 0;
      have_ihdr = true;
    } else if (!std::memcmp(type, "PLTE", 4)) {
      raw->palette.assign(data, data + len);
    } else if (!std::memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!std::memcmp(type, "IEND", 4)) {
      break;
    }
    pos += 12 + (size_t)len;
  }
  if (!have_ihdr || raw->width <= 0 || raw->height <= 0) { *err = path + ": missing IHDR"; return false; }
  if (interlaced) { *err = path + ": interlaced PNG is not supported"; return false; }
  switch (raw->color_type) {
    case 0: raw->channels = 1; break;
    case 2: raw->channels = 3; break;
    case 3: raw->channels = 1; break;
    case 4: raw->channels = 2; break;
    case 6: raw->channels = 4; break;
    default: *err = path + ": unknown colour type"; return false;
  }
  if (!(raw->bit_depth == 8 || raw->bit_depth == 16) || (raw->color_type == 3 && raw->bit_depth != 8)) {
    *err = path + ": only 8- and 16-bit samples are supported";
    return false;
  }
  const size_t bpp = (size_t)raw->channels * raw->bit_depth / 8;
  const size_t stride = bpp * (size_t)raw->width;
  // a header can claim any size: refuse what no camera produces before allocating for it
  if (raw->width > 65535 || raw->height > 65535 || (stride + 1) * (size_t)raw->height > ((size_t)1 << 30)) {
    *err = path + ": image dimensions out of range";
    return false;
  }
  std::vector<uint8_t> infl((stride + 1) * (size_t)raw->height);
  uLongf out_len = (uLongf)infl.size();
  if (uncompress(infl.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != infl.size()) {
    *err = path + ": zlib inflate failed";
    return false;
  }
  raw->rows.assign(stride * (size_t)raw->height, 0);
  for (int y = 0; y < raw->height; y++) {            // undo the per-scanline filters (PNG spec, section 9)
    const uint8_t ft = infl[(stride + 1) * (size_t)y];
    const uint8_t *in = &infl[(stride + 1) * (size_t)y + 1];
    uint8_t *cur = &raw->rows[stride * (size_t)y];
    const uint8_t *up = y ? &raw->rows[stride * (size_t)(y - 1)] : nullptr;
    for (size_t i = 0; i < stride; i++) {
      const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
      int v = in[i];
      switch (ft) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) / 2; break;
        case 4: {
          const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
          v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
          break;
        }
        default: *err = path + ": bad filter type"; return false;
      }
      cur[i] = (uint8_t)v;
    }
  }
  return true;
}

// 8-bit value of sample `ch` of pixel x in a scanline (16-bit samples are reduced by dropping the low
// byte, as cv::imread does without IMREAD_ANYDEPTH).
inline int sample8(const Raw &r, const uint8_t *row, int x, int ch)
{
  const size_t bps = (size_t)r.bit_depth / 8;
  return row[((size_t)x * r.channels + ch) * bps];
}

bool store(const std::string &path, int w, int h, int bit_depth, const uint8_t *be_rows, std::string *err)
{
  const size_t stride = (size_t)w * bit_depth / 8;
  std::vector<uint8_t> rawdata((stride + 1) * (size_t)h);
  for (int y = 0; y < h; y++) {
    rawdata[(stride + 1) * (size_t)y] = 0;
    std::memcpy(&rawdata[(stride + 1) * (size_t)y + 1], be_rows + stride * (size_t)y, stride);
  }
  uLongf clen = compressBound((uLong)rawdata.size());
  std::vector<uint8_t> comp(clen);
  if (compress2(comp.data(), &clen, rawdata.data(), (uLong)rawdata.size(), 6) != Z_OK) { *err = "zlib deflate failed"; return false; }
  FILE *f = std::fopen(path.c_str(), "wb");
  if (!f) { *err = "cannot write " + path; return false; }
  auto chunk = [&](const char *type, const uint8_t *d, uint32_t len) {
    uint8_t hdr[8] = {(uint8_t)(len >> 24), (uint8_t)(len >> 16), (uint8_t)(len >> 8), (uint8_t)len,
                      (uint8_t)type[0], (uint8_t)type[1], (uint8_t)type[2], (uint8_t)type[3]};
    std::fwrite(hdr, 1, 8, f);
    if (len) std::fwrite(d, 1, len, f);
    uLong crc = crc32(0L, hdr + 4, 4);
    if (len) crc = crc32(crc, d, len);
    const uint8_t c[4] = {(uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc};
    std::fwrite(c, 1, 4, f);
  };
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  std::fwrite(sig, 1, 8, f);
  uint8_t ihdr[13] = {(uint8_t)(w >> 24), (uint8_t)(w >> 16), (uint8_t)(w >> 8), (uint8_t)w,
                      (uint8_t)(h >> 24), (uint8_t)(h >> 16), (uint8_t)(h >> 8), (uint8_t)h,
                      (uint8_t)bit_depth, 0, 0, 0, 0};
  chunk("IHDR", ihdr, 13);
  chunk("IDAT", comp.data(), (uint32_t)clen);
  chunk("IEND", nullptr, 0);
  std::fclose(f);
  return true;
}

}  // namespace

bool read_gray8(const std::string &path, Image8 *out, std::string *err)
{
  Raw r;
  if (!load(path, &r, err)) return false;
  out->width = r.width; out->height = r.height;
  out->pixels.resize((size_t)r.width * r.height);
  const size_t stride = (size_t)r.channels * r.bit_depth / 8 * (size_t)r.width;
  for (int y = 0; y < r.height; y++) {
    const uint8_t *row = &r.rows[stride * (size_t)y];
    for (int x = 0; x < r.width; x++) {
      int R, G, B;
      if (r.color_type == 0 || r.color_type == 4) {
        out->pixels[(size_t)y * r.width + x] = (uint8_t)sample8(r, row, x, 0);
        continue;
      } else if (r.color_type == 3) {
        const size_t idx = row[x];
        if (3 * idx + 2 >= r.palette.size()) { *err = path + ": palette index out of range"; return false; }
        R = r.palette[3 * idx]; G = r.palette[3 * idx + 1]; B = r.palette[3 * idx + 2];
      } else {
        R = sample8(r, row, x, 0); G = sample8(r, row, x, 1); B = sample8(r, row, x, 2);
      }
      // OpenCV's PNG decoder asks libpng for the conversion: png_set_rgb_to_gray(png, 1, 0.299, 0.587),
      // i.e. 15-bit fixed point coefficients 9797 / 19234 / 3737 (recalled from OpenCV 2.4 + libpng 1.2+;
      // not checkable in this image, which has neither).
      out->pixels[(size_t)y * r.width + x] = (uint8_t)((9797 * R + 19234 * G + 3737 * B + 16384) >> 15);
    }
  }
  return true;
}

bool read_unchanged16(const std::string &path, Image16 *out, std::string *err)
{
  Raw r;
  if (!load(path, &r, err)) return false;
  if (r.channels != 1 || r.color_type == 3) { *err = path + ": depth image must be single-channel gray"; return false; }
  out->width = r.width; out->height = r.height;
  out->pixels.resize((size_t)r.width * r.height);
  const size_t bps = (size_t)r.bit_depth / 8, stride = bps * (size_t)r.width;
  for (int y = 0; y < r.height; y++) {
    const uint8_t *row = &r.rows[stride * (size_t)y];
    for (int x = 0; x < r.width; x++)
      out->pixels[(size_t)y * r.width + x] =
          bps == 2 ? (uint16_t)((row[2 * x] << 8) | row[2 * x + 1]) : (uint16_t)row[x];
  }
  return true;
}

bool write_gray8(const std::string &path, int w, int h, const uint8_t *px, std::string *err)
{
  return store(path, w, h, 8, px, err);
}

bool write_gray16(const std::string &path, int w, int h, const uint16_t *px, std::string *err)
{
  std::vector<uint8_t> be((size_t)w * h * 2);
  for (size_t i = 0; i < (size_t)w * h; i++) { be[2 * i] = (uint8_t)(px[i] >> 8); be[2 * i + 1] = (uint8_t)px[i]; }
  return store(path, w, h, 16, be.data(), err);
}

}  // namespace phovo_io
