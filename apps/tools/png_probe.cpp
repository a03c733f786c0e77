// Test helper: decodes a PNG with apps/io/png_io and dumps "W H\n" + raw samples to a file, or re-encodes.
//   png_probe gray8 in.png out.raw | png_probe raw16 in.png out.raw | png_probe copy8 in.png out.png | png_probe copy16 in.png out.png
#include <cstdio>
#include <cstring>
#include <iostream>

#include "io/png_io.h"

int main(int argc, char **argv)
{
  if (argc < 4) { std::cerr << "usage: png_probe gray8|raw16|copy8|copy16 in out" << std::endl; return 2; }
  std::string err;
  const std::string mode(argv[1]);
  if (mode == "gray8" || mode == "copy8") {
    phovo_io::Image8 im;
    if (!phovo_io::read_gray8(argv[2], &im, &err)) { std::cerr << err << std::endl; return 1; }
    if (mode == "copy8") {
      if (!phovo_io::write_gray8(argv[3], im.width, im.height, im.pixels.data(), &err)) { std::cerr << err << std::endl; return 1; }
      return 0;
    }
    FILE *f = std::fopen(argv[3], "wb");
    std::fprintf(f, "%d %d\n", im.width, im.height);
    std::fwrite(im.pixels.data(), 1, im.pixels.size(), f);
    std::fclose(f);
  } else {
    phovo_io::Image16 im;
    if (!phovo_io::read_unchanged16(argv[2], &im, &err)) { std::cerr << err << std::endl; return 1; }
    if (mode == "copy16") {
      if (!phovo_io::write_gray16(argv[3], im.width, im.height, im.pixels.data(), &err)) { std::cerr << err << std::endl; return 1; }
      return 0;
    }
    FILE *f = std::fopen(argv[3], "wb");
    std::fprintf(f, "%d %d\n", im.width, im.height);
    std::fwrite(im.pixels.data(), 2, im.pixels.size(), f);      // host (little) endian
    std::fclose(f);
  }
  return 0;
}
