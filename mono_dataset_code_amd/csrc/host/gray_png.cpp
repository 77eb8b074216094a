#include "gray_png.h"

#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace mdc_host {
namespace {

unsigned be32(const unsigned char* p) { return (unsigned)p[0] << 24 | (unsigned)p[1] << 16 | (unsigned)p[2] << 8 | p[3]; }

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

int paeth(int a, int b, int c) {
  int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

GrayImage decode_png(const std::vector<unsigned char>& buf) {
  GrayImage none;
  size_t pos = 8;
  unsigned w = 0, h = 0;
  int depth = 0, ctype = -1, interlace = 0;
  std::vector<unsigned char> idat;
  while (pos + 12 <= buf.size()) {
    unsigned len = be32(&buf[pos]);
    const unsigned char* tag = &buf[pos + 4];
    if (pos + 12 + (size_t)len > buf.size()) return none;
    const unsigned char* body = &buf[pos + 8];
    if (!memcmp(tag, "IHDR", 4) && len >= 13) {
      w = be32(body); h = be32(body + 4);
      depth = body[8]; ctype = body[9]; interlace = body[12];
    } else if (!memcmp(tag, "IDAT", 4)) {
      idat.insert(idat.end(), body, body + len);
    } else if (!memcmp(tag, "IEND", 4)) {
      break;
    }
    pos += 12 + (size_t)len;
  }
  if (w == 0 || h == 0 || ctype != 0 || interlace != 0 || (depth != 8 && depth != 16)) return none;
  const size_t bpp = (size_t)depth / 8, stride = (size_t)w * bpp;
  std::vector<unsigned char> raw((stride + 1) * h);
  uLongf got = (uLongf)raw.size();
  if (uncompress(raw.data(), &got, idat.data(), (uLong)idat.size()) != Z_OK || got != raw.size()) return none;
  std::vector<unsigned char> prev(stride, 0), cur(stride);
  GrayImage im;
  im.width = (int)w; im.height = (int)h; im.bits = depth;
  im.px.resize((size_t)w * h);
  for (unsigned y = 0; y < h; y++) {
    const unsigned char* line = &raw[(stride + 1) * y];
    const int ft = line[0];
    for (size_t i = 0; i < stride; i++) {
      int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0, x = line[1 + i];
      int v;
      switch (ft) {
        case 0: v = x; break;
        case 1: v = x + a; break;
        case 2: v = x + b; break;
        case 3: v = x + ((a + b) >> 1); break;
        case 4: v = x + paeth(a, b, c); break;
        default: return none;
      }
      cur[i] = (unsigned char)v;
    }
    for (unsigned x = 0; x < w; x++)
      im.px[(size_t)y * w + x] = depth == 8 ? cur[x] : (unsigned short)(cur[2 * x] << 8 | cur[2 * x + 1]);
    prev.swap(cur);
  }
  return im;
}

GrayImage decode_pgm(const std::vector<unsigned char>& buf) {
  GrayImage none;
  int w = 0, h = 0, maxv = 0, used = 0;
  if (sscanf((const char*)buf.data(), "P5 %d %d %d%n", &w, &h, &maxv, &used) != 3 || w <= 0 || h <= 0) return none;
  size_t off = (size_t)used + 1, bps = maxv > 255 ? 2 : 1;
  if (off + (size_t)w * h * bps > buf.size()) return none;
  GrayImage im;
  im.width = w; im.height = h; im.bits = bps == 2 ? 16 : 8;
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
