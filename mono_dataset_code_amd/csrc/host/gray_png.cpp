#include "gray_png.h"
#include "image_codecs_internal.h"

#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace mdc_host {
namespace {

bool slurp(const std::string& path, std::vector<unsigned char>& buf) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  if (n < 0) { fclose(f); return false; }
  buf.resize((size_t)n);
  bool ok = n == 0 || fread(buf.data(), 1, (size_t)n, f) == (size_t)n;
  fclose(f);
  return ok;
}

GrayImage decode_png(const std::vector<unsigned char>& buf) {
  GrayImage none;
  PngAny any;
  std::string err;
  if (!png_decode_any(buf.data(), buf.size(), any, &err)) return none;
  GrayImage im;
  im.width = any.w;
  im.height = any.h;
  im.channels = any.channels;
  if (any.channels != 1) return im;  // bits stays 0: not a single-channel image
  im.bits = any.bits;
  im.px.assign(any.px.begin(), any.px.end());
  return im;
}

GrayImage decode_pgm(const std::vector<unsigned char>& buf) {
  GrayImage none;
  int w = 0, h = 0, maxv = 0, used = 0;
  if (sscanf((const char*)buf.data(), "P5 %d %d %d%n", &w, &h, &maxv, &used) != 3 || w <= 0 || h <= 0) return none;
  size_t off = (size_t)used + 1, bps = maxv > 255 ? 2 : 1;
  if (off + (size_t)w * h * bps > buf.size()) return none;
  GrayImage im;
  im.width = w; im.height = h; im.channels = 1; im.bits = bps == 2 ? 16 : 8;
  im.px.resize((size_t)w * h);
  for (size_t i = 0; i < (size_t)w * h; i++)
    im.px[i] = bps == 1 ? buf[off + i] : (unsigned short)(buf[off + 2 * i] << 8 | buf[off + 2 * i + 1]);
  return im;
}

}  // namespace

GrayImage read_gray_image(const std::string& path) {
  std::vector<unsigned char> buf;
  if (!slurp(path, buf) || buf.size() < 16) return GrayImage();
  static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (!memcmp(buf.data(), sig, 8)) return decode_png(buf);
  buf.push_back(0);
  if (buf[0] == 'P' && buf[1] == '5') return decode_pgm(buf);
  return GrayImage();
}

}  // namespace mdc_host
