#include "image_codecs.h"
#include "image_codecs_internal.h"
#include "mdc_hip.h"

#include <zlib.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace mdc_host {

bool read_file(const std::string& path, std::vector<unsigned char>& buf) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  if (n < 0) {
    fclose(f);
    return false;
  }
  buf.resize((size_t)n);
  const bool ok = n == 0 || fread(buf.data(), 1, (size_t)n, f) == (size_t)n;
  fclose(f);
  return ok;
}

namespace {

bool fail(std::string* err, const char* msg) {
  if (err) *err = msg;
  return false;
}

// ---------------------------------------------------------------------------------------------
// PNG: 8-bit grayscale, non-interlaced (what the dataset's lossless frames are)
// ---------------------------------------------------------------------------------------------
unsigned be32(const unsigned char* p) { return (unsigned)p[0] << 24 | (unsigned)p[1] << 16 | (unsigned)p[2] << 8 | p[3]; }

int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

bool png_gray8(const unsigned char* d, size_t n, unsigned char* out, size_t cap, int* w, int* h, std::string* err) {
  size_t pos = 8;
  unsigned W = 0, H = 0;
  int depth = 0, ctype = -1, interlace = 0;
  // scratch that survives between frames of one thread: fresh multi-megabyte vectors per frame mean an mmap, page
  // faults and an munmap each, and the kernel's address-space lock then serialises the decode threads
  static thread_local std::vector<unsigned char> idat, raw;
  idat.clear();
  while (pos + 12 <= n) {
    const unsigned len = be32(d + pos);
    const unsigned char* tag = d + pos + 4;
    if (pos + 12 + (size_t)len > n) return fail(err, "PNG: truncated chunk");
    const unsigned char* body = d + pos + 8;
    if (!memcmp(tag, "IHDR", 4) && len >= 13) {
      W = be32(body);
      H = be32(body + 4);
      depth = body[8];
      ctype = body[9];
      interlace = body[12];
    } else if (!memcmp(tag, "IDAT", 4)) {
      idat.insert(idat.end(), body, body + len);
    } else if (!memcmp(tag, "IEND", 4)) {
      break;
    }
    pos += 12 + (size_t)len;
  }
  if (W == 0 || H == 0 || W > 65535 || H > 65535) return fail(err, "PNG: no IHDR");
  *w = (int)W;
  *h = (int)H;
  if ((size_t)W * H > cap) return fail(err, "frame larger than the buffer");
  if (ctype != 0 || depth != 8 || interlace != 0) {  // any other PNG flavour: general decoder + OpenCV's conversion to 8-bit gray
    PngAny im;
    if (!png_decode_any(d, n, im, err)) return false;
    png_any_to_gray8(im, out);
    return true;
  }
  const size_t stride = W;
  raw.resize((stride + 1) * H);
  uLongf got = (uLongf)raw.size();
  if (uncompress(raw.data(), &got, idat.data(), (uLong)idat.size()) != Z_OK || got != raw.size()) return fail(err, "PNG: bad IDAT stream");
  const unsigned char* prev = nullptr;
  for (unsigned y = 0; y < H; y++) {
    const unsigned char* line = &raw[(stride + 1) * y];
    unsigned char* cur = out + (size_t)y * W;
    const int ft = line[0];
    for (size_t i = 0; i < stride; i++) {
      const int a = i ? cur[i - 1] : 0, b = prev ? prev[i] : 0, c = (i && prev) ? prev[i - 1] : 0, x = line[1 + i];
      int v;
      switch (ft) {
        case 0: v = x; break;
        case 1: v = x + a; break;
        case 2: v = x + b; break;
        case 3: v = x + ((a + b) >> 1); break;
        case 4: v = x + paeth(a, b, c); break;
        default: return fail(err, "PNG: bad filter type");
      }
      cur[i] = (unsigned char)v;
    }
    prev = cur;
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// PGM (P5, maxval <= 255)
// ---------------------------------------------------------------------------------------------
bool pgm_gray8(const unsigned char* d, size_t n, unsigned char* out, size_t cap, int* w, int* h, std::string* err) {
  // header: "P5" ws width ws height ws maxval single-ws, '#' comments allowed between tokens
  size_t p = 2;
  int vals[3] = {0, 0, 0};
  for (int k = 0; k < 3; k++) {
    for (;;) {
      while (p < n && (d[p] == ' ' || d[p] == '\t' || d[p] == '\r' || d[p] == '\n')) p++;
      if (p < n && d[p] == '#') {
        while (p < n && d[p] != '\n') p++;
        continue;
      }
      break;
    }
    if (p >= n || d[p] < '0' || d[p] > '9') return fail(err, "PGM: bad header");
    long v = 0;
    while (p < n && d[p] >= '0' && d[p] <= '9' && v < 100000000) v = v * 10 + (d[p++] - '0');
    vals[k] = (int)v;
  }
  p++;  // the single whitespace byte after maxval
  if (vals[0] <= 0 || vals[1] <= 0) return fail(err, "PGM: bad size");
  *w = vals[0];
  *h = vals[1];
  if (vals[2] != 65535 && (vals[2] > 255 || vals[2] <= 0)) return fail(err, "PGM: only maxval <= 255 or 65535 is supported");
  const size_t px = (size_t)vals[0] * vals[1];
  if (px > cap) return fail(err, "frame larger than the buffer");
  if (vals[2] == 65535) {  // 16-bit samples (big endian): the high byte, as OpenCV's 8-bit read of such a file
    if (p + 2 * px > n) return fail(err, "PGM: truncated");
    for (size_t i = 0; i < px; i++) out[i] = d[p + 2 * i];
    return true;
  }
  if (p + px > n) return fail(err, "PGM: truncated");
  memcpy(out, d + p, px);
  return true;
}

// ---------------------------------------------------------------------------------------------
// Baseline JPEG
// ---------------------------------------------------------------------------------------------
const unsigned char kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                   30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
  bool present = false;
  unsigned char vals[256];
  uint16_t look[512];  // codes of <= 9 bits: (length << 8) | symbol, 0 = longer code
  int maxcode[18];     // largest code of length l (or -1), maxcode[17] = sentinel
  int valoff[17];      // vals index of the first code of length l minus that code
  // AC tables only: for a 9-bit window that holds a whole (code, magnitude bits) pair: value << 8 | run << 4 | bits used;
  // 0 = not such a window (long code, long magnitude, EOB or ZRL)
  int16_t fast_ac[512];
  unsigned char def[16 + 256];  // the table as the file defines it (16 counts + the symbols): identity of the table
  int def_len = 0;
};

bool build_huff(Huff& t, const unsigned char* bits /*[1..16] at bits[0..15]*/, const unsigned char* vals, int nvals) {
  memset(t.look, 0, sizeof t.look);
  memcpy(t.vals, vals, (size_t)nvals);
  int code = 0, k = 0;
  for (int l = 1; l <= 16; l++) {
    t.valoff[l] = k - code;
    const int cnt = bits[l - 1];
    if (k + cnt > 256 || code + cnt > (1 << l)) return false;
    for (int i = 0; i < cnt; i++, k++, code++)
      if (l <= 9) {
        const int first = code << (9 - l);
        for (int f = 0; f < (1 << (9 - l)); f++) t.look[first + f] = (uint16_t)(l << 8 | vals[k]);
      }
    t.maxcode[l] = cnt ? code - 1 : -1;
    code <<= 1;
  }
  t.maxcode[17] = 0x7fffffff;
  t.present = k == nvals;
  for (int w = 0; w < 512; w++) {
    t.fast_ac[w] = 0;
    const int e = t.look[w];
    if (!e) continue;
    const int len = e >> 8, rs = e & 255, run = rs >> 4, sz = rs & 15;
    if (sz == 0 || len + sz > 9) continue;
    int v = (w >> (9 - len - sz)) & ((1 << sz) - 1);  // the magnitude bits that follow the code inside the window
    if (v < (1 << (sz - 1))) v += (int)((~0u) << sz) + 1;  // EXTEND
    if (v >= -128 && v <= 127) t.fast_ac[w] = (int16_t)(v * 256 + run * 16 + (len + sz));
  }
  return t.present;
}

struct Bits {  // entropy-coded segment reader: FF00 unstuffing, stops (feeding zeros) at a marker
  const unsigned char* p;
  const unsigned char* end;
  uint64_t acc = 0;
  int cnt = 0;
  bool hit_marker = false;
  void fill() {
    while (cnt <= 56) {
      unsigned b = 0;
      if (!hit_marker && p < end) {
        b = *p;
        if (b == 0xff) {
          if (p + 1 < end && p[1] == 0) p += 2;
          else {
            hit_marker = true;
            b = 0;
          }
        } else p++;
      }
      acc |= (uint64_t)b << (56 - cnt);
      cnt += 8;
    }
  }
  int peek(int n) { return (int)(acc >> (64 - n)); }
  void skip(int n) {
    acc <<= n;
    cnt -= n;
  }
  int get(int n) {
    if (n == 0) return 0;
    if (cnt < n) fill();
    const int v = peek(n);
    skip(n);
    return v;
  }
  void reset_at(const unsigned char* q) {
    p = q;
    acc = 0;
    cnt = 0;
    hit_marker = false;
  }
};

inline int decode_sym(Bits& b, const Huff& t) {
  if (b.cnt < 16) b.fill();
  const int e = t.look[b.peek(9)];
  if (e) {
    b.skip(e >> 8);
    return e & 255;
  }
  int code = b.peek(10), l = 10;
  while (code > t.maxcode[l]) {
    if (++l > 16) return -1;
    code = b.peek(l);
  }
  b.skip(l);
  const int idx = code + t.valoff[l];
  return (idx >= 0 && idx < 256) ? t.vals[idx] : -1;
}

inline int extend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }

// libjpeg's jidctint.c ("islow"), 8x8: the accurate integer inverse DCT every libjpeg / libjpeg-turbo
// build uses by default -- same constants, same two passes, same rounding, so the samples agree bit for bit.
inline int descale(long x, int n) { return (int)((x + (1L << (n - 1))) >> n); }
inline unsigned char clamp_sample(int x) {
  x += 128;
  return (unsigned char)(x < 0 ? 0 : (x > 255 ? 255 : x));
}
void idct_islow(const int* coef /* dequantised, natural order */, unsigned char* out, size_t stride, bool dc_only) {
  if (dc_only) {  // both passes collapse: DESCALE(dc << 2, 5) everywhere (the shortcuts of jidctint.c applied twice)
    const unsigned char v = clamp_sample(descale((long)coef[0] * 4, 5));
    for (int r = 0; r < 8; r++) memset(out + (size_t)r * stride, v, 8);
    return;
  }
  const long F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299,
             F1_847 = 15137, F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
  const int CB = 13, P1 = 2;
  int ws[64];
  for (int c = 0; c < 8; c++) {
    const int* in = coef + c;
    if (!(in[8] | in[16] | in[24] | in[32] | in[40] | in[48] | in[56])) {
      const int dc = in[0] * (1 << P1);
      for (int r = 0; r < 8; r++) ws[r * 8 + c] = dc;
      continue;
    }
    long z2 = in[16], z3 = in[48];
    long z1 = (z2 + z3) * F0_541;
    long tmp2 = z1 + z3 * (-F1_847), tmp3 = z1 + z2 * F0_765;
    z2 = in[0];
    z3 = in[32];
    long tmp0 = (z2 + z3) * (1L << CB), tmp1 = (z2 - z3) * (1L << CB);
    const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[56];
    tmp1 = in[40];
    tmp2 = in[24];
    tmp3 = in[8];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    long z4 = tmp1 + tmp3;
    const long z5 = (z3 + z4) * F1_175;
    tmp0 *= F0_298;
    tmp1 *= F2_053;
    tmp2 *= F3_072;
    tmp3 *= F1_501;
    z1 *= -F0_899;
    z2 *= -F2_562;
    z3 *= -F1_961;
    z4 *= -F0_390;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    ws[0 * 8 + c] = descale(tmp10 + tmp3, CB - P1);
    ws[7 * 8 + c] = descale(tmp10 - tmp3, CB - P1);
    ws[1 * 8 + c] = descale(tmp11 + tmp2, CB - P1);
    ws[6 * 8 + c] = descale(tmp11 - tmp2, CB - P1);
    ws[2 * 8 + c] = descale(tmp12 + tmp1, CB - P1);
    ws[5 * 8 + c] = descale(tmp12 - tmp1, CB - P1);
    ws[3 * 8 + c] = descale(tmp13 + tmp0, CB - P1);
    ws[4 * 8 + c] = descale(tmp13 - tmp0, CB - P1);
  }
  for (int r = 0; r < 8; r++) {
    const int* w = ws + r * 8;
    unsigned char* o = out + (size_t)r * stride;
    if (!(w[1] | w[2] | w[3] | w[4] | w[5] | w[6] | w[7])) {  // jidctint.c's row shortcut: same value as the full pass
      memset(o, clamp_sample(descale(w[0], 5)), 8);
      continue;
    }
    long z2 = w[2], z3 = w[6];
    long z1 = (z2 + z3) * F0_541;
    long tmp2 = z1 + z3 * (-F1_847), tmp3 = z1 + z2 * F0_765;
    long tmp0 = ((long)w[0] + w[4]) * (1L << CB), tmp1 = ((long)w[0] - w[4]) * (1L << CB);
    const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = w[7];
    tmp1 = w[5];
    tmp2 = w[3];
    tmp3 = w[1];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    long z4 = tmp1 + tmp3;
    const long z5 = (z3 + z4) * F1_175;
    tmp0 *= F0_298;
    tmp1 *= F2_053;
    tmp2 *= F3_072;
    tmp3 *= F1_501;
    z1 *= -F0_899;
    z2 *= -F2_562;
    z3 *= -F1_961;
    z4 *= -F0_390;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    const int S = CB + P1 + 3;
    o[0] = clamp_sample(descale(tmp10 + tmp3, S));
    o[7] = clamp_sample(descale(tmp10 - tmp3, S));
    o[1] = clamp_sample(descale(tmp11 + tmp2, S));
    o[6] = clamp_sample(descale(tmp11 - tmp2, S));
    o[2] = clamp_sample(descale(tmp12 + tmp1, S));
    o[5] = clamp_sample(descale(tmp12 - tmp1, S));
    o[3] = clamp_sample(descale(tmp13 + tmp0, S));
    o[4] = clamp_sample(descale(tmp13 - tmp0, S));
  }
}

struct Comp {
  int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0;
};

bool jpeg_gray8(const unsigned char* d, size_t n, unsigned char* out, size_t cap, int* w, int* h, std::string* err, JpegCoefSink* sink = nullptr) {
  uint16_t qt[4][64];
  bool have_qt[4] = {false, false, false, false};
  Huff dc[4], ac[4];
  Comp comp[4];
  int ncomp = 0, W = 0, H = 0, restart = 0;
  size_t p = 2;
  bool have_sof = false;
  JpegColorMarkers color;
  while (p + 4 <= n) {
    if (d[p] != 0xff) return fail(err, "JPEG: marker expected");
    while (p < n && d[p] == 0xff) p++;  // fill bytes
    if (p >= n) break;
    const int m = d[p++];
    if (m == 0xd8 || (m >= 0xd0 && m <= 0xd7) || m == 0x01) continue;
    if (m == 0xd9) break;
    if (p + 2 > n) return fail(err, "JPEG: truncated");
    const size_t len = (size_t)d[p] << 8 | d[p + 1];
    if (len < 2 || p + len > n) return fail(err, "JPEG: bad segment length");
    const unsigned char* s = d + p + 2;
    const size_t sl = len - 2;
    color.see(m, s, sl);
    if (m == 0xdb) {  // DQT
      size_t q = 0;
      while (q < sl) {
        const int pq = s[q] >> 4, tq = s[q] & 15;
        q++;
        if (tq > 3 || q + (pq ? 128 : 64) > sl) return fail(err, "JPEG: bad DQT");
        for (int i = 0; i < 64; i++, q += pq ? 2 : 1) qt[tq][kZigzag[i]] = pq ? (uint16_t)(s[q] << 8 | s[q + 1]) : s[q];
        have_qt[tq] = true;
      }
    } else if (m == 0xc4) {  // DHT
      size_t q = 0;
      while (q + 17 <= sl) {
        const int tc = s[q] >> 4, th = s[q] & 15;
        int cnt = 0;
        for (int i = 0; i < 16; i++) cnt += s[q + 1 + i];
        if (th > 3 || tc > 1 || cnt > 256 || q + 17 + (size_t)cnt > sl) return fail(err, "JPEG: bad DHT");
        if (!build_huff(tc ? ac[th] : dc[th], s + q + 1, s + q + 17, cnt)) return fail(err, "JPEG: bad Huffman table");
        q += 17 + (size_t)cnt;
      }
    } else if (m == 0xc0 || m == 0xc1) {  // SOF0 / SOF1: sequential, Huffman
      if (sl < 6 || s[0] != 8) return fail(err, "JPEG: only 8-bit samples are supported");
      H = s[1] << 8 | s[2];
      W = s[3] << 8 | s[4];
      ncomp = s[5];
      if ((ncomp != 1 && ncomp != 3) || sl < 6 + 3 * (size_t)ncomp || W <= 0 || H <= 0) return fail(err, "JPEG: unsupported frame header");
      for (int i = 0; i < ncomp; i++) {
        comp[i].id = s[6 + 3 * i];
        comp[i].h = s[7 + 3 * i] >> 4;
        comp[i].v = s[7 + 3 * i] & 15;
        comp[i].tq = s[8 + 3 * i] & 3;
        if (comp[i].h < 1 || comp[i].h > 4 || comp[i].v < 1 || comp[i].v > 4) return fail(err, "JPEG: bad sampling factors");
      }
      have_sof = true;
      *w = W;
      *h = H;
    } else if (m == 0xc2) {  // progressive: its own decoder (image_codecs_ext.cpp)
      return jpeg_progressive_gray8(d, n, out, cap, w, h, err, sink);
    } else if (m >= 0xc3 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc) {
      return fail(err, "JPEG: lossless, hierarchical and arithmetic-coded files are not supported");
    } else if (m == 0xdd) {  // DRI
      if (sl >= 2) restart = s[0] << 8 | s[1];
    } else if (m == 0xda) {  // SOS: the one scan of a baseline file
      if (!have_sof) return fail(err, "JPEG: scan before frame header");
      if (!sink && (size_t)W * H > cap) return fail(err, "frame larger than the buffer");
      const int ns = s[0];
      if (ns != ncomp || sl < 1 + 2 * (size_t)ns + 3) return fail(err, "JPEG: only single-scan files are supported");
      for (int i = 0; i < ns; i++) {
        int k = -1;
        for (int c = 0; c < ncomp; c++)
          if (comp[c].id == s[1 + 2 * i]) k = c;
        if (k != i) return fail(err, "JPEG: unexpected component order");
        comp[k].td = s[2 + 2 * i] >> 4;
        comp[k].ta = s[2 + 2 * i] & 15;
        if (comp[k].td > 3 || comp[k].ta > 3 || !dc[comp[k].td].present || !ac[comp[k].ta].present || !have_qt[comp[k].tq])
          return fail(err, "JPEG: scan refers to a missing table");
        comp[k].pred = 0;
      }
      // geometry: a single-component scan is non-interleaved (one block per MCU)
      const int hmax = ncomp == 1 ? 1 : std::max(comp[0].h, std::max(comp[1].h, comp[2].h));
      const int vmax = ncomp == 1 ? 1 : std::max(comp[0].v, std::max(comp[1].v, comp[2].v));
      const int yh = ncomp == 1 ? 1 : comp[0].h, yv = ncomp == 1 ? 1 : comp[0].v;
      if (ncomp == 3 && (yh != hmax || yv != vmax)) return fail(err, "JPEG: luma is subsampled; unsupported");
      // An RGB-encoded file (Adobe transform 0, or component ids R G B): gray is a weighted sum of ALL three components, not
      // component 0 -- no luma record for the device, and on the host every component is inverted (1 x 1 sampling only).
      const bool rgb = color.is_rgb(ncomp, comp[0].id, comp[1].id, comp[2].id);
      if (rgb && sink) return fail(err, "JPEG: RGB-encoded file: no luma coefficient record");
      if (rgb && (hmax != 1 || vmax != 1)) return fail(err, "JPEG: RGB-encoded file with subsampled components is not supported");
      const int mcu_w = 8 * hmax, mcu_h = 8 * vmax;
      const int mx = (W + mcu_w - 1) / mcu_w, my = (H + mcu_h - 1) / mcu_h;
      const size_t pw = (size_t)mx * mcu_w;  // padded luma row (luma has the full resolution)
      static thread_local std::vector<unsigned char> rows;
      if (!sink) rows.resize(pw * mcu_h * (rgb ? 3 : 1));  // rgb: the R, G and B block rows one after the other
      if (sink) {  // coefficient output: quantised luma coefficients, natural order, [block row][block][64]; no inverse DCT here
        sink->w = W;
        sink->h = H;
        if (sink->pitch_blocks && sink->pitch_blocks < mx * yh) return fail(err, "coefficient row pitch too small for this file");
        sink->blocks_w = sink->pitch_blocks ? sink->pitch_blocks : mx * yh;
        sink->blocks_rows = my * yv;
        if ((size_t)sink->blocks_w * sink->blocks_rows > sink->cap_blocks) return fail(err, "frame larger than the coefficient buffer");
        for (int i = 0; i < 64; i++) sink->quant[i] = qt[comp[0].tq][i];
      }
      Bits b;
      b.p = d + p + len;
      b.end = d + n;
      int coef[64];
      int to_restart = restart;
      for (int y = 0; y < my; y++) {
        for (int x = 0; x < mx; x++) {
          if (restart && to_restart == 0) {  // RSTn: byte-align, skip the marker, reset predictions
            const unsigned char* q = b.p;
            while (q + 1 < b.end && !(q[0] == 0xff && q[1] >= 0xd0 && q[1] <= 0xd7)) q++;
            if (q + 1 >= b.end) return fail(err, "JPEG: missing restart marker");
            b.reset_at(q + 2);
            for (int c = 0; c < ncomp; c++) comp[c].pred = 0;
            to_restart = restart;
          }
          for (int c = 0; c < ncomp; c++) {
            const int nb = ncomp == 1 ? 1 : comp[c].h * comp[c].v;
            for (int k = 0; k < nb; k++) {
              const bool luma = c == 0 || rgb;  // the component is kept (rgb: all three; then sink == nullptr)
              int16_t* blk = nullptr;  // (coefficient output) this luma block
              if (luma && sink) {
                const int bx = ncomp == 1 ? 0 : k % comp[c].h, by = ncomp == 1 ? 0 : k / comp[c].h;
                blk = sink->coef + ((size_t)(y * yv + by) * sink->blocks_w + (size_t)x * yh + bx) * 64;
                memset(blk, 0, 64 * sizeof(int16_t));
              } else if (luma) {
                memset(coef, 0, sizeof coef);
              }
              const uint16_t* q = qt[comp[c].tq];
              int t = decode_sym(b, dc[comp[c].td]);
              if (t < 0 || t > 11) return fail(err, "JPEG: bad DC code");
              comp[c].pred += t ? extend(b.get(t), t) : 0;
              if (blk) blk[0] = (int16_t)comp[c].pred;
              else if (luma) coef[0] = comp[c].pred * q[0];
              bool dc_only = true;
              const Huff& act = ac[comp[c].ta];
              for (int i = 1; i < 64;) {
                if (b.cnt < 16) b.fill();
                const int fa = act.fast_ac[b.peek(9)];
                if (fa) {  // code + magnitude in one lookup
                  i += (fa >> 4) & 15;
                  if (i > 63) return fail(err, "JPEG: coefficient index out of range");
                  b.skip(fa & 15);
                  if (blk) blk[kZigzag[i]] = (int16_t)(fa >> 8);
                  else if (luma) coef[kZigzag[i]] = (fa >> 8) * q[kZigzag[i]];
                  dc_only = false;
                  i++;
                  continue;
                }
                const int rs = decode_sym(b, act);
                if (rs < 0) return fail(err, "JPEG: bad AC code");
                const int r = rs >> 4, sz = rs & 15;
                if (sz == 0) {
                  if (r != 15) break;  // EOB
                  i += 16;
                  continue;
                }
                i += r;
                if (i > 63) return fail(err, "JPEG: coefficient index out of range");
                const int v = extend(b.get(sz), sz);
                if (blk) blk[kZigzag[i]] = (int16_t)v;
                else if (luma) coef[kZigzag[i]] = v * q[kZigzag[i]];
                dc_only = false;
                i++;
              }
              if (luma && !sink) {
                const int bx = ncomp == 1 ? 0 : k % comp[c].h, by = ncomp == 1 ? 0 : k / comp[c].h;
                idct_islow(coef, rows.data() + (rgb ? (size_t)c * pw * mcu_h : 0) + (size_t)by * 8 * pw + (size_t)x * mcu_w + (size_t)bx * 8, pw, dc_only);
              }
            }
          }
          if (restart) to_restart--;
        }
        const int y0 = y * mcu_h, ny = std::min(mcu_h, H - y0);
        if (!sink && !rgb)
          for (int r = 0; r < ny; r++) memcpy(out + (size_t)(y0 + r) * W, rows.data() + (size_t)r * pw, (size_t)W);
        if (!sink && rgb)  // libjpeg's rgb_gray_convert (jdcolor.c): (FIX(0.299) R + FIX(0.587) G + FIX(0.114) B + ONE_HALF) >> 16
          for (int r = 0; r < ny; r++) {
            const unsigned char *R = rows.data() + (size_t)r * pw, *G = R + pw * mcu_h, *B = G + pw * mcu_h;
            unsigned char* o = out + (size_t)(y0 + r) * W;
            for (int xx = 0; xx < W; xx++) o[xx] = (unsigned char)((19595 * R[xx] + 38470 * G[xx] + 7471 * B[xx] + 32768) >> 16);
          }
      }
      return true;
    }
    p += len;
  }
  return fail(err, "JPEG: no scan found");
}

}  // namespace

void jpeg_idct_islow(const int* coef, unsigned char* out, size_t stride, bool dc_only) { idct_islow(coef, out, stride, dc_only); }

bool decode_jpeg_coefs(const unsigned char* d, size_t n, JpegCoefSink* sink, std::string* err) {
  if (!sink || !sink->coef) return fail(err, "no coefficient buffer");
  if (n < 4 || d[0] != 0xff || d[1] != 0xd8) return fail(err, "not a JPEG file");
  int w = 0, h = 0;
  return jpeg_gray8(d, n, nullptr, 0, &w, &h, err, sink);
}

namespace {

// The device's form of one Huffman table (mdc_jpeg_huff, include/mdc_hip.h): one lookup of the next 11 bits gives code length,
// run and size -- and the value itself where the magnitude bits lie inside the window; codes of 12..16 bits go through a
// 32-entry subtable per prefix.
bool build_device_table(const Huff& t, bool is_ac, mdc_jpeg_huff* dst, std::string* err) {
  memset(dst, 0, sizeof *dst);
  // symbol and length of the code the 16-bit window `w16` starts with (0 = none)
  auto code_of = [&](int w16, int* sym) {
    const int e = t.look[w16 >> 7];
    if (e) {
      *sym = e & 255;
      return e >> 8;
    }
    for (int l = 10; l <= 16; l++) {
      const int code = w16 >> (16 - l);
      if (code <= t.maxcode[l]) {
        const int idx = code + t.valoff[l];
        if (idx < 0 || idx > 255) return 0;
        *sym = t.vals[idx];
        return l;
      }
    }
    return 0;
  };
  const int k = is_ac ? 1 : 0;
  int nsub = 0;
  for (int w11 = 0; w11 < 2048; w11++) {
    int sym = 0;
    const int l0 = code_of(w11 << 5, &sym);  // (with the 5 bits below the window zero: right for every code of <= 11 bits)
    uint32_t e = 0;
    bool is_short = l0 >= 1 && l0 <= 11;
    if (is_short) {  // confirm: the code must not depend on the bits below the window
      int sym1 = 0;
      is_short = code_of(w11 << 5 | 31, &sym1) == l0 && sym1 == sym;
    }
    if (is_short) {
      const int run = k ? sym >> 4 : 0, sz = k ? sym & 15 : sym;
      if (!k && sym > 15) {
        dst->t1[w11] = 0;
        continue;
      }
      e = (uint32_t)l0 | (uint32_t)run << 5 | (uint32_t)sz << 9;
      if (sz && l0 + sz <= 11) {
        int v = (w11 >> (11 - l0 - sz)) & ((1 << sz) - 1);
        if (v < (1 << (sz - 1))) v += (int)((~0u) << sz) + 1;  // EXTEND
        e |= 1u << 13 | (uint32_t)(uint16_t)(int16_t)v << 16;
      }
    } else {  // longer codes below this prefix?
      uint32_t sub[32];
      bool any = false;
      for (int sfx = 0; sfx < 32; sfx++) {
        int s2 = 0;
        const int l = code_of(w11 << 5 | sfx, &s2);
        sub[sfx] = 0;
        if (l >= 12 && l <= 16 && (k || s2 <= 15)) {
          sub[sfx] = (uint32_t)l | (uint32_t)(k ? s2 >> 4 : 0) << 5 | (uint32_t)(k ? s2 & 15 : s2) << 9;
          any = true;
        }
      }
      if (any) {
        if (nsub >= MDC_JPEG_HUFF_SUBTABLES) return fail(err, "JPEG stream: too many long Huffman codes for the device tables");
        memcpy(dst->t2[nsub], sub, sizeof sub);
        e = 31u | (uint32_t)nsub << 16;
        nsub++;
      }
    }
    dst->t1[w11] = e;
  }
  return true;
}

// The files of a sequence carry the same tables (an encoder's defaults): built once per decode thread and table definition, then
// copied -- building them was 55 of the 86 us jpeg_stream took for a 265-KB file (31 now).  Slots: DC / AC x luma / chroma.
struct DeviceTableCache {
  bool valid[4] = {false, false, false, false};
  int len[4] = {0, 0, 0, 0};
  unsigned char def[4][16 + 256];
  mdc_jpeg_huff tab[4];
};
bool device_table(const Huff& t, bool is_ac, int slot, mdc_jpeg_huff* dst, std::string* err) {
  static thread_local DeviceTableCache cache;
  if (cache.valid[slot] && cache.len[slot] == t.def_len && memcmp(cache.def[slot], t.def, (size_t)t.def_len) == 0) {
    *dst = cache.tab[slot];
    return true;
  }
  cache.valid[slot] = false;
  if (!build_device_table(t, is_ac, dst, err)) return false;
  cache.len[slot] = t.def_len;
  memcpy(cache.def[slot], t.def, (size_t)t.def_len);
  cache.tab[slot] = *dst;
  cache.valid[slot] = true;
  return true;
}

}  // namespace

bool jpeg_stream(const unsigned char* d, size_t n, unsigned char* stream, size_t cap, size_t* used, int* w, int* h, std::string* err) {
  if (n < 4 || d[0] != 0xff || d[1] != 0xd8) return fail(err, "not a JPEG file");
  if (!stream || cap < sizeof(mdc_jpeg_stream_header) + 32 || (reinterpret_cast<uintptr_t>(stream) & 3) != 0) return fail(err, "stream buffer too small");
  uint16_t qt[4][64];
  bool have_qt[4] = {false, false, false, false};
  Huff dc[4], ac[4];
  int W = 0, H = 0, ncomp = 0, restart = 0;
  struct Comp {
    int id, h, v, tq;
  } comp[3] = {{0, 1, 1, 0}, {0, 1, 1, 0}, {0, 1, 1, 0}};
  bool have_sof = false;
  JpegColorMarkers color;
  size_t p = 2;
  while (p + 4 <= n) {
    if (d[p] != 0xff) return fail(err, "JPEG: marker expected");
    while (p < n && d[p] == 0xff) p++;
    if (p >= n) break;
    const int m = d[p++];
    if (m == 0xd8 || m == 0x01) continue;
    if (m >= 0xd0 && m <= 0xd7) return fail(err, "JPEG: restart marker outside a scan");
    if (m == 0xd9) break;
    if (p + 2 > n) return fail(err, "JPEG: truncated");
    const size_t len = (size_t)d[p] << 8 | d[p + 1];
    if (len < 2 || p + len > n) return fail(err, "JPEG: bad segment length");
    const unsigned char* s = d + p + 2;
    const size_t sl = len - 2;
    color.see(m, s, sl);
    if (m == 0xdb) {
      size_t q = 0;
      while (q < sl) {
        const int pq = s[q] >> 4, t = s[q] & 15;
        q++;
        if (t > 3 || q + (pq ? 128 : 64) > sl) return fail(err, "JPEG: bad DQT");
        for (int i = 0; i < 64; i++, q += pq ? 2 : 1) qt[t][kZigzag[i]] = pq ? (uint16_t)(s[q] << 8 | s[q + 1]) : s[q];
        have_qt[t] = true;
      }
    } else if (m == 0xc4) {
      size_t q = 0;
      while (q + 17 <= sl) {
        const int tc = s[q] >> 4, th = s[q] & 15;
        int cnt = 0;
        for (int i = 0; i < 16; i++) cnt += s[q + 1 + i];
        if (th > 3 || tc > 1 || cnt > 256 || q + 17 + (size_t)cnt > sl) return fail(err, "JPEG: bad DHT");
        if (!build_huff(tc ? ac[th] : dc[th], s + q + 1, s + q + 17, cnt)) return fail(err, "JPEG: bad Huffman table");
        Huff& hh = tc ? ac[th] : dc[th];
        memcpy(hh.def, s + q + 1, 16 + (size_t)cnt);
        hh.def_len = 16 + cnt;
        q += 17 + (size_t)cnt;
      }
    } else if (m == 0xc0 || m == 0xc1) {
      if (sl < 9 || s[0] != 8) return fail(err, "JPEG: only 8-bit samples are supported");
      H = s[1] << 8 | s[2];
      W = s[3] << 8 | s[4];
      ncomp = s[5];
      if ((ncomp != 1 && ncomp != 3) || sl < 6 + 3 * (size_t)ncomp || W <= 0 || H <= 0) return fail(err, "JPEG stream: unsupported frame header");
      for (int i = 0; i < ncomp; i++) {
        comp[i].id = s[6 + 3 * i];
        comp[i].h = s[7 + 3 * i] >> 4;
        comp[i].v = s[7 + 3 * i] & 15;
        comp[i].tq = s[8 + 3 * i] & 3;
        if (comp[i].h < 1 || comp[i].h > 4 || comp[i].v < 1 || comp[i].v > 4) return fail(err, "JPEG: bad sampling factors");
      }
      if (ncomp == 1) comp[0].h = comp[0].v = 1;
      // the device decodes luma h x v + one block of each chroma component per MCU (4:4:4, 4:2:2, 4:2:0, 4:1:1, ...)
      if (ncomp == 3 && (comp[1].h != 1 || comp[1].v != 1 || comp[2].h != 1 || comp[2].v != 1)) return fail(err, "JPEG stream: chroma sampling is not 1 x 1");
      have_sof = true;
    } else if (m >= 0xc2 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc) {
      return fail(err, "JPEG stream: not a sequential Huffman file");
    } else if (m == 0xdd) {
      if (sl >= 2) restart = s[0] << 8 | s[1];
    } else if (m == 0xda) {
      if (!have_sof) return fail(err, "JPEG: scan before frame header");
      if (sl < 1 || s[0] != ncomp || sl < 1 + 2 * (size_t)ncomp + 3) return fail(err, "JPEG stream: the components are not in one scan");
      // the device keeps component 0 only: right for Y Cb Cr, wrong for an RGB-encoded file (gray = weighted sum of R, G, B) -> host decoder
      if (color.is_rgb(ncomp, comp[0].id, comp[1].id, comp[2].id)) return fail(err, "JPEG stream: RGB-encoded file (no luma component)");
      int td[3] = {0, 0, 0}, ta[3] = {0, 0, 0};
      for (int i = 0; i < ncomp; i++) {
        if (s[1 + 2 * i] != comp[i].id) return fail(err, "JPEG stream: scan components out of frame order");
        td[i] = s[2 + 2 * i] >> 4;
        ta[i] = s[2 + 2 * i] & 15;
        if (td[i] > 3 || ta[i] > 3 || !dc[td[i]].present || !ac[ta[i]].present) return fail(err, "JPEG: scan refers to a missing table");
      }
      if (!have_qt[comp[0].tq]) return fail(err, "JPEG: scan refers to a missing table");
      // one table pair for both chroma components (every encoder's choice; two different pairs would need a fifth and sixth table)
      if (ncomp == 3 && (td[1] != td[2] || ta[1] != ta[2])) return fail(err, "JPEG stream: Cb and Cr use different Huffman tables");
      const int hY = comp[0].h, vY = comp[0].v;
      const int mx = (W + 8 * hY - 1) / (8 * hY), my = (H + 8 * vY - 1) / (8 * vY);
      const long mcus = (long)mx * my;
      const long n_iv = restart ? (mcus + restart - 1) / restart : 1;
      if (n_iv >= (1l << 24)) return fail(err, "JPEG stream: too many restart intervals");
      mdc_jpeg_stream_header* hd = reinterpret_cast<mdc_jpeg_stream_header*>(stream);
      size_t off = sizeof *hd + (ncomp == 3 ? 2 * sizeof(mdc_jpeg_huff) : 0);
      const size_t starts_off = off;
      if (restart) off += (size_t)n_iv * 4;
      off = (off + 15) & ~(size_t)15;
      if (off + 32 > cap) return fail(err, "JPEG stream: does not fit the buffer");
      memset(hd, 0, sizeof *hd);
      hd->magic = MDC_JPEG_STREAM_MAGIC;
      hd->w = (uint32_t)W;
      hd->h = (uint32_t)H;
      hd->restart_interval = (uint32_t)restart;
      hd->n_intervals = (uint32_t)n_iv;
      hd->comp_info = (uint32_t)ncomp | (uint32_t)hY << 8 | (uint32_t)vY << 12;
      hd->ecs_offset = (uint32_t)off;
      for (int i = 0; i < 64; i++) hd->quant[i] = qt[comp[0].tq][i];
      if (!device_table(dc[td[0]], false, 0, &hd->dc, err) || !device_table(ac[ta[0]], true, 1, &hd->ac, err)) return false;
      if (ncomp == 3) {
        mdc_jpeg_huff* chroma = reinterpret_cast<mdc_jpeg_huff*>(hd + 1);
        if (!device_table(dc[td[1]], false, 2, &chroma[0], err) || !device_table(ac[ta[1]], true, 3, &chroma[1], err)) return false;
      }
      uint32_t* starts = reinterpret_cast<uint32_t*>(stream + starts_off);
      memset(stream + starts_off, 0, off - starts_off);
      // entropy-coded segment without its byte stuffing and its restart markers; ends at the first other marker (EOI)
      const unsigned char* q = d + p + len;
      const unsigned char* end = d + n;
      unsigned char* const o0 = stream + off;
      unsigned char* o = o0;
      unsigned char* const o_end = stream + cap - 16;
      long iv = 0;  // intervals begun
      int expect_rst = 0;
      if (restart) starts[iv] = 0;
      iv = 1;
      while (q < end) {
        const unsigned char* ff = static_cast<const unsigned char*>(memchr(q, 0xff, (size_t)(end - q)));
        const size_t run = ff ? (size_t)(ff - q) : (size_t)(end - q);
        if (o + run + 1 > o_end) return fail(err, "JPEG stream: does not fit the buffer");
        memcpy(o, q, run);
        o += run;
        q += run;
        if (!ff) break;
        if (q + 1 < end && q[1] == 0x00) {  // stuffed zero: a data byte FF
          *o++ = 0xff;
          q += 2;
        } else if (q + 1 < end && q[1] == 0xff) {  // fill byte
          q++;
        } else if (q + 1 < end && q[1] >= 0xd0 && q[1] <= 0xd7) {
          // RSTm: the next interval begins at the next byte (what came before it is padded to a byte with 1-bits)
          if (!restart || q[1] != 0xd0 + expect_rst) return fail(err, "JPEG stream: unexpected restart marker");
          if (iv >= n_iv) return fail(err, "JPEG stream: more restart intervals than the frame has");
          expect_rst = (expect_rst + 1) & 7;
          starts[iv++] = (uint32_t)(o - o0);
          q += 2;
        } else {
          break;  // EOI (or any other marker): end of the scan
        }
      }
      if (restart && iv != n_iv) return fail(err, "JPEG stream: fewer restart intervals than the frame has");
      const size_t ecs = (size_t)(o - o0);
      if (ecs == 0 || ecs >= (1u << 28)) return fail(err, "JPEG stream: empty scan");
      memset(o, 0, 16);
      hd->ecs_bytes = (uint32_t)ecs;
      *used = off + ecs + 16;
      *w = W;
      *h = H;
      return true;
    }
    p += len;
  }
  return fail(err, "JPEG: no scan found");
}

bool decode_gray8(const unsigned char* d, size_t n, unsigned char* out, size_t cap, int* w, int* h, std::string* err) {
  *w = *h = 0;
  static const unsigned char png_sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (n >= 16 && !memcmp(d, png_sig, 8)) return png_gray8(d, n, out, cap, w, h, err);
  if (n >= 4 && d[0] == 0xff && d[1] == 0xd8) return jpeg_gray8(d, n, out, cap, w, h, err);
  if (n >= 8 && d[0] == 'P' && d[1] == '5') return pgm_gray8(d, n, out, cap, w, h, err);
  return fail(err, "unknown image format (PNG, PGM P5 and JPEG are supported)");
}

}  // namespace mdc_host
