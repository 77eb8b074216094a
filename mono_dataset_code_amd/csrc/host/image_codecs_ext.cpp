// The less common inputs cv::imread(..., CV_LOAD_IMAGE_GRAYSCALE) accepts for a frame (reference
// src/BenchmarkDatasetReader.h:252,274) and cv::imread(..., CV_LOAD_IMAGE_UNCHANGED) for the vignette
// (src/PhotometricUndistorter.cpp:120), without OpenCV:
//
//   PNG   every colour type (gray, gray + alpha, RGB, RGBA, palette), every bit depth (1, 2, 4, 8, 16), Adam7
//         interlacing.  Conversion to 8-bit gray as OpenCV's PNG reader configures libpng for a grayscale read:
//           16-bit samples      -> the high byte                 (png_set_strip_16; after the colour conversion)
//           alpha               -> dropped                       (png_set_strip_alpha)
//           1/2/4-bit gray      -> scaled to 8 bits (x 255, 85, 17)  (png_set_expand_gray_1_2_4_to_8)
//           palette             -> RGB, then as RGB              (png_set_palette_to_rgb)
//           RGB                 -> libpng's png_set_rgb_to_gray(1, 0.299, 0.587): 15-bit fixed point with TRUNCATED
//                                  coefficients 9797 / 19234 / 3737 (sum 32768); 8-bit samples
//                                  (9797 R + 19234 G + 3737 B) >> 15 (no rounding), 16-bit samples + 16384 before the
//                                  shift; R == G == B passes through.  (Not PIL's "L": (19595 R + 38470 G + 7471 B
//                                  + 32768) >> 16 -- the two differ by at most 1; tests/test_reader_cpu.py states both.)
//         (The common case -- 8-bit gray, non-interlaced -- never comes here: image_codecs.cpp decodes it in place.)
//   JPEG  progressive (SOF2): spectral selection + successive approximation, DC / AC first and refinement scans,
//         restart intervals; luma only for YCbCr files (libjpeg's JCS_GRAYSCALE), chroma-only scans are skipped, the
//         interleaved DC scans are parsed for all components to stay in step.  Coefficients -> the same islow
//         inverse DCT as the baseline decoder -> bit-identical to libjpeg / libjpeg-turbo.
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "image_codecs.h"
#include "image_codecs_internal.h"

namespace mdc_host {
namespace {

bool fail(std::string* err, const char* msg) {
  if (err) *err = msg;
  return false;
}
unsigned be32(const unsigned char* p) { return (unsigned)p[0] << 24 | (unsigned)p[1] << 16 | (unsigned)p[2] << 8 | p[3]; }
int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// PNG, general
// ---------------------------------------------------------------------------------------------------------
bool png_decode_any(const unsigned char* d, size_t n, PngAny& im, std::string* err) {
  im = PngAny();
  size_t pos = 8;
  unsigned W = 0, H = 0;
  int depth = 0, ctype = -1, interlace = 0;
  std::vector<unsigned char> idat, plte;
  bool have_ihdr = false;
  while (pos + 12 <= n) {
    const unsigned len = be32(d + pos);
    const unsigned char* tag = d + pos + 4;
    if (len > n || pos + 12 + (size_t)len > n) return fail(err, "PNG: truncated chunk");
    const unsigned char* body = d + pos + 8;
    if (!memcmp(tag, "IHDR", 4) && len >= 13) {
      W = be32(body);
      H = be32(body + 4);
      depth = body[8];
      ctype = body[9];
      interlace = body[12];
      have_ihdr = true;
      if (body[10] != 0 || body[11] != 0) return fail(err, "PNG: unknown compression / filter method");
    } else if (!memcmp(tag, "PLTE", 4)) {
      plte.assign(body, body + len);
    } else if (!memcmp(tag, "IDAT", 4)) {
      idat.insert(idat.end(), body, body + len);
    } else if (!memcmp(tag, "IEND", 4)) {
      break;
    }
    pos += 12 + (size_t)len;
  }
  if (!have_ihdr || W == 0 || H == 0 || W > 65535 || H > 65535) return fail(err, "PNG: no IHDR");
  im.w = (int)W;
  im.h = (int)H;
  int fch;  // channels in the file
  switch (ctype) {
    case 0: fch = 1; if (depth != 1 && depth != 2 && depth != 4 && depth != 8 && depth != 16) return fail(err, "PNG: bad bit depth"); break;
    case 2: fch = 3; if (depth != 8 && depth != 16) return fail(err, "PNG: bad bit depth"); break;
    case 3: fch = 1; if (depth != 1 && depth != 2 && depth != 4 && depth != 8) return fail(err, "PNG: bad bit depth"); break;
    case 4: fch = 2; if (depth != 8 && depth != 16) return fail(err, "PNG: bad bit depth"); break;
    case 6: fch = 4; if (depth != 8 && depth != 16) return fail(err, "PNG: bad bit depth"); break;
    default: return fail(err, "PNG: bad colour type");
  }
  if (interlace > 1) return fail(err, "PNG: bad interlace method");
  if (ctype == 3 && plte.size() < 3) return fail(err, "PNG: palette image without PLTE");
  im.channels = ctype == 3 ? 3 : fch;
  im.bits = depth == 16 ? 16 : 8;
  im.palette = ctype == 3;
  if ((size_t)W * H * (size_t)im.channels > ((size_t)1 << 28)) return fail(err, "PNG: image too large");
  // passes: non-interlaced = one pass over the whole image
  static const int X0[7] = {0, 4, 0, 2, 0, 1, 0}, Y0[7] = {0, 0, 4, 0, 2, 0, 1}, DX[7] = {8, 8, 4, 4, 2, 2, 1}, DY[7] = {8, 8, 8, 4, 4, 2, 2};
  const int npass = interlace ? 7 : 1;
  size_t total = 0;
  unsigned pw[7], ph[7];
  for (int p = 0; p < npass; p++) {
    pw[p] = interlace ? (W + DX[p] - 1 - X0[p]) / DX[p] : W;
    ph[p] = interlace ? (H + DY[p] - 1 - Y0[p]) / DY[p] : H;
    if (interlace && ((unsigned)X0[p] >= W || (unsigned)Y0[p] >= H)) pw[p] = ph[p] = 0;
    if (pw[p] && ph[p]) total += (size_t)ph[p] * (1 + ((size_t)pw[p] * fch * depth + 7) / 8);
  }
  std::vector<unsigned char> raw(total);
  uLongf got = (uLongf)raw.size();
  if (idat.empty() || uncompress(raw.data(), &got, idat.data(), (uLong)idat.size()) != Z_OK || got != raw.size())
    return fail(err, "PNG: bad IDAT stream");
  im.px.assign((size_t)W * H * im.channels, 0);
  const size_t bpp = std::max<size_t>(1, (size_t)fch * depth / 8);
  size_t off = 0;
  std::vector<unsigned char> prev, cur;
  for (int p = 0; p < npass; p++) {
    if (!pw[p] || !ph[p]) continue;
    const size_t stride = ((size_t)pw[p] * fch * depth + 7) / 8;
    prev.assign(stride, 0);
    cur.assign(stride, 0);
    for (unsigned y = 0; y < ph[p]; y++) {
      const unsigned char* line = &raw[off];
      off += stride + 1;
      const int ft = line[0];
      if (ft > 4) return fail(err, "PNG: bad filter type");
      for (size_t i = 0; i < stride; i++) {
        const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0, x = line[1 + i];
        int v;
        switch (ft) {
          case 0: v = x; break;
          case 1: v = x + a; break;
          case 2: v = x + b; break;
          case 3: v = x + ((a + b) >> 1); break;
          default: v = x + paeth(a, b, c); break;
        }
        cur[i] = (unsigned char)v;
      }
      const size_t oy = interlace ? (size_t)Y0[p] + (size_t)y * DY[p] : y;
      for (unsigned x = 0; x < pw[p]; x++) {
        const size_t ox = interlace ? (size_t)X0[p] + (size_t)x * DX[p] : x;
        uint16_t* o = &im.px[(oy * W + ox) * im.channels];
        if (depth == 16) {
          for (int ch = 0; ch < fch; ch++) o[ch] = (uint16_t)(cur[((size_t)x * fch + ch) * 2] << 8 | cur[((size_t)x * fch + ch) * 2 + 1]);
        } else if (depth == 8) {
          if (ctype == 3) {
            const size_t idx = cur[x];
            if (idx * 3 + 2 >= plte.size()) return fail(err, "PNG: palette index out of range");
            o[0] = plte[idx * 3];
            o[1] = plte[idx * 3 + 1];
            o[2] = plte[idx * 3 + 2];
          } else {
            for (int ch = 0; ch < fch; ch++) o[ch] = cur[(size_t)x * fch + ch];
          }
        } else {  // 1, 2, 4 bits: one channel (gray or palette index), most significant bits first
          const size_t bit = (size_t)x * depth;
          const unsigned v = (cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1);
          if (ctype == 3) {
            if ((size_t)v * 3 + 2 >= plte.size()) return fail(err, "PNG: palette index out of range");
            o[0] = plte[v * 3];
            o[1] = plte[v * 3 + 1];
            o[2] = plte[v * 3 + 2];
          } else {
            o[0] = (uint16_t)(v * (255u / ((1u << depth) - 1)));  // x 255, 85, 17
          }
        }
      }
      prev.swap(cur);
    }
  }
  return true;
}

// 8-bit gray as OpenCV's grayscale read of that PNG (see the file comment)
void png_any_to_gray8(const PngAny& im, unsigned char* out) {
  const size_t npx = (size_t)im.w * im.h;
  const int ch = im.channels;
  for (size_t i = 0; i < npx; i++) {
    const uint16_t* s = &im.px[i * ch];
    unsigned v;
    if (ch <= 2) {
      v = s[0];
    } else {
      const unsigned r = s[0], g = s[1], b = s[2];
      if (r == g && r == b) v = r;
      else if (im.bits == 16) v = (9797u * r + 19234u * g + 3737u * b + 16384u) >> 15;
      else v = (9797u * r + 19234u * g + 3737u * b) >> 15;
    }
    out[i] = (unsigned char)(im.bits == 16 ? v >> 8 : v);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Progressive JPEG (luma only)
// ---------------------------------------------------------------------------------------------------------
namespace {

const unsigned char kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                   30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct PHuff {
  bool present = false;
  unsigned char vals[256];
  int maxcode[18], valoff[17], mincode[17];
};
bool build_phuff(PHuff& t, const unsigned char* bits, const unsigned char* vals, int nvals) {
  memcpy(t.vals, vals, (size_t)nvals);
  int code = 0, k = 0;
  for (int l = 1; l <= 16; l++) {
    t.valoff[l] = k - code;
    const int cnt = bits[l - 1];
    if (k + cnt > 256 || code + cnt > (1 << l)) return false;
    t.maxcode[l] = cnt ? code + cnt - 1 : -1;
    k += cnt;
    code = (code + cnt) << 1;
  }
  t.maxcode[17] = 0x7fffffff;
  t.present = k == nvals;
  return t.present;
}

struct PBits {
  const unsigned char* p;
  const unsigned char* end;
  uint32_t acc = 0;
  int cnt = 0;
  bool hit_marker = false;
  int bit() {
    if (cnt == 0) {
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
      acc = b;
      cnt = 8;
    }
    cnt--;
    return (acc >> cnt) & 1;
  }
  int get(int n) {
    int v = 0;
    for (int i = 0; i < n; i++) v = v << 1 | bit();
    return v;
  }
  void reset_at(const unsigned char* q) {
    p = q;
    cnt = 0;
    hit_marker = false;
  }
};
int decode_psym(PBits& b, const PHuff& t) {
  int code = 0;
  for (int l = 1; l <= 16; l++) {
    code = code << 1 | b.bit();
    if (code <= t.maxcode[l]) {
      const int idx = code + t.valoff[l];
      return (idx >= 0 && idx < 256) ? t.vals[idx] : -1;
    }
  }
  return -1;
}
inline int pextend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }

struct PComp {
  int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0;
};

}  // namespace

bool jpeg_progressive_gray8(const unsigned char* d, size_t n, unsigned char* out, size_t cap, int* w, int* h, std::string* err, JpegCoefSink* sink) {
  uint16_t qt[4][64];
  bool have_qt[4] = {false, false, false, false};
  PHuff dc[4], ac[4];
  PComp comp[4];
  int ncomp = 0, W = 0, H = 0, restart = 0;
  bool have_sof = false, saw_luma_scan = false;
  int hmax = 1, vmax = 1, mx = 0, my = 0;  // MCU grid
  int lbw = 0, lbh = 0;                    // luma blocks: allocated grid (MCU padded)
  int lcw = 0, lch = 0;                    // luma blocks a non-interleaved luma scan covers
  std::vector<int16_t> coef;               // luma coefficients, [block][64] natural order
  JpegColorMarkers color;
  size_t p = 2;
  while (p + 4 <= n) {
    if (d[p] != 0xff) return fail(err, "JPEG: marker expected");
    while (p < n && d[p] == 0xff) p++;
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
    // only component 0 is kept below: wrong for an RGB-encoded file (cv::imread weighs R, G and B) -> refused, loudly
    if (m == 0xda && color.is_rgb(ncomp, comp[0].id, comp[1].id, comp[2].id))
      return fail(err, "JPEG: progressive RGB-encoded files (Adobe transform 0 / component ids R G B) are not supported");
    if (m == 0xdb) {
      size_t q = 0;
      while (q < sl) {
        const int pq = s[q] >> 4, tq = s[q] & 15;
        q++;
        if (tq > 3 || q + (pq ? 128 : 64) > sl) return fail(err, "JPEG: bad DQT");
        for (int i = 0; i < 64; i++, q += pq ? 2 : 1) qt[tq][kZigzag[i]] = pq ? (uint16_t)(s[q] << 8 | s[q + 1]) : s[q];
        have_qt[tq] = true;
      }
    } else if (m == 0xc4) {
      size_t q = 0;
      while (q + 17 <= sl) {
        const int tc = s[q] >> 4, th = s[q] & 15;
        int cnt = 0;
        for (int i = 0; i < 16; i++) cnt += s[q + 1 + i];
        if (th > 3 || tc > 1 || cnt > 256 || q + 17 + (size_t)cnt > sl) return fail(err, "JPEG: bad DHT");
        if (!build_phuff(tc ? ac[th] : dc[th], s + q + 1, s + q + 17, cnt)) return fail(err, "JPEG: bad Huffman table");
        q += 17 + (size_t)cnt;
      }
    } else if (m == 0xc2) {
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
        hmax = std::max(hmax, comp[i].h);
        vmax = std::max(vmax, comp[i].v);
      }
      if (ncomp == 1) comp[0].h = comp[0].v = hmax = vmax = 1;
      if (comp[0].h != hmax || comp[0].v != vmax) return fail(err, "JPEG: luma is subsampled; unsupported");
      *w = W;
      *h = H;
      if (!sink && (size_t)W * H > cap) return fail(err, "frame larger than the buffer");
      mx = (W + 8 * hmax - 1) / (8 * hmax);
      my = (H + 8 * vmax - 1) / (8 * vmax);
      lbw = mx * hmax;
      lbh = my * vmax;
      lcw = (W + 7) / 8;
      lch = (H + 7) / 8;
      if (sink) {  // checked BEFORE the scans' coefficient store is sized from the header (65535 x 65535 would ask for 8.6 GB)
        if (sink->pitch_blocks && sink->pitch_blocks < lbw) return fail(err, "coefficient row pitch too small for this file");
        if ((size_t)(sink->pitch_blocks ? sink->pitch_blocks : lbw) * lbh > sink->cap_blocks) return fail(err, "frame larger than the coefficient buffer");
      }
      coef.assign((size_t)lbw * lbh * 64, 0);
      have_sof = true;
    } else if (m == 0xc0 || m == 0xc1 || (m >= 0xc3 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc)) {
      return fail(err, "JPEG: not a progressive Huffman file");
    } else if (m == 0xdd) {
      if (sl >= 2) restart = s[0] << 8 | s[1];
    } else if (m == 0xda) {
      if (!have_sof) return fail(err, "JPEG: scan before frame header");
      const int ns = s[0];
      if (ns < 1 || ns > ncomp || sl < 1 + 2 * (size_t)ns + 3) return fail(err, "JPEG: bad scan header");
      int sc[4];
      bool has_luma = false;
      for (int i = 0; i < ns; i++) {
        int k = -1;
        for (int c = 0; c < ncomp; c++)
          if (comp[c].id == s[1 + 2 * i]) k = c;
        if (k < 0) return fail(err, "JPEG: scan names an unknown component");
        sc[i] = k;
        comp[k].td = s[2 + 2 * i] >> 4;
        comp[k].ta = s[2 + 2 * i] & 15;
        if (comp[k].td > 3 || comp[k].ta > 3) return fail(err, "JPEG: bad table index");
        if (k == 0) has_luma = true;
      }
      const int Ss = s[1 + 2 * ns], Se = s[2 + 2 * ns], Ah = s[3 + 2 * ns] >> 4, Al = s[3 + 2 * ns] & 15;
      if (Ss > Se || Se > 63 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1) || Al > 13) return fail(err, "JPEG: bad progression parameters");
      const unsigned char* ecs = d + p + len;
      if (has_luma) {
        saw_luma_scan = true;
        PBits b;
        b.p = ecs;
        b.end = d + n;
        for (int c = 0; c < ncomp; c++) comp[c].pred = 0;
        int eobrun = 0, to_restart = restart;
        auto do_restart = [&]() -> bool {
          const unsigned char* q = b.p;
          while (q + 1 < b.end && !(q[0] == 0xff && q[1] >= 0xd0 && q[1] <= 0xd7)) q++;
          if (q + 1 >= b.end) return false;
          b.reset_at(q + 2);
          for (int c = 0; c < ncomp; c++) comp[c].pred = 0;
          eobrun = 0;
          to_restart = restart;
          return true;
        };
        if (Ss == 0) {  // DC scan: interleaved over the scan's components (MCU order) or luma alone
          for (int i = 0; i < ns; i++)
            if (!Ah && !dc[comp[sc[i]].td].present) return fail(err, "JPEG: scan refers to a missing table");
          const bool inter = ns > 1;
          const int bw = inter ? mx : lcw, bh = inter ? my : lch;
          for (int y = 0; y < bh; y++)
            for (int x = 0; x < bw; x++) {
              if (restart && to_restart == 0 && !do_restart()) return fail(err, "JPEG: missing restart marker");
              for (int i = 0; i < ns; i++) {
                PComp& cc = comp[sc[i]];
                const int nb = inter ? cc.h * cc.v : 1;
                for (int k = 0; k < nb; k++) {
                  int16_t* blk = nullptr;
                  if (sc[i] == 0) {
                    const int bx = inter ? x * cc.h + k % cc.h : x, by = inter ? y * cc.v + k / cc.h : y;
                    blk = &coef[((size_t)by * lbw + bx) * 64];
                  }
                  if (!Ah) {
                    const int t = decode_psym(b, dc[cc.td]);
                    if (t < 0 || t > 11) return fail(err, "JPEG: bad DC code");
                    cc.pred += t ? pextend(b.get(t), t) : 0;
                    if (blk) blk[0] = (int16_t)(cc.pred * (1 << Al));
                  } else {
                    const int bit = b.bit();
                    if (blk && bit) blk[0] = (int16_t)(blk[0] | (1 << Al));
                  }
                }
              }
              if (restart) to_restart--;
            }
        } else {  // AC scan of the luma component alone
          const PHuff& act = ac[comp[0].ta];
          if (!act.present) return fail(err, "JPEG: scan refers to a missing table");
          const int p1 = 1 << Al, m1 = -(1 << Al);
          for (int y = 0; y < lch; y++)
            for (int x = 0; x < lcw; x++) {
              if (restart && to_restart == 0 && !do_restart()) return fail(err, "JPEG: missing restart marker");
              int16_t* blk = &coef[((size_t)y * lbw + x) * 64];
              if (!Ah) {  // first pass over this band
                if (eobrun > 0) eobrun--;
                else
                  for (int k = Ss; k <= Se; k++) {
                    const int rs = decode_psym(b, act);
                    if (rs < 0) return fail(err, "JPEG: bad AC code");
                    const int r = rs >> 4, sz = rs & 15;
                    if (sz) {
                      k += r;
                      if (k > Se) return fail(err, "JPEG: coefficient index out of range");
                      blk[kZigzag[k]] = (int16_t)(pextend(b.get(sz), sz) * (1 << Al));
                    } else if (r == 15) {
                      k += 15;
                    } else {
                      eobrun = (1 << r) - 1;
                      if (r) eobrun += b.get(r);
                      break;
                    }
                  }
              } else {  // refinement (ITU T.81 G.1.2.3, as libjpeg's decode_mcu_AC_refine)
                int k = Ss;
                if (eobrun == 0) {
                  for (; k <= Se; k++) {
                    const int rs = decode_psym(b, act);
                    if (rs < 0) return fail(err, "JPEG: bad AC code");
                    int r = rs >> 4, sz = rs & 15;
                    if (sz) {
                      if (sz != 1) return fail(err, "JPEG: bad refinement code");
                      sz = b.bit() ? p1 : m1;
                    } else if (r != 15) {
                      eobrun = 1 << r;
                      if (r) eobrun += b.get(r);
                      break;
                    }
                    do {
                      int16_t* cf = &blk[kZigzag[k]];
                      if (*cf != 0) {
                        if (b.bit() && (*cf & p1) == 0) *cf = (int16_t)(*cf + (*cf >= 0 ? p1 : m1));
                      } else if (--r < 0) {
                        break;
                      }
                      k++;
                    } while (k <= Se);
                    if (sz && k <= Se) blk[kZigzag[k]] = (int16_t)sz;
                  }
                }
                if (eobrun > 0) {
                  for (; k <= Se; k++) {
                    int16_t* cf = &blk[kZigzag[k]];
                    if (*cf != 0 && b.bit() && (*cf & p1) == 0) *cf = (int16_t)(*cf + (*cf >= 0 ? p1 : m1));
                  }
                  eobrun--;
                }
              }
              if (restart) to_restart--;
            }
        }
      }
      // on to the next marker after the entropy-coded segment (RSTn and stuffed FF00 belong to it)
      size_t q = (size_t)(ecs - d);
      while (q + 1 < n && !(d[q] == 0xff && d[q + 1] != 0 && !(d[q + 1] >= 0xd0 && d[q + 1] <= 0xd7))) q++;
      p = q;
      continue;
    }
    p += len;
  }
  if (!have_sof || !saw_luma_scan) return fail(err, "JPEG: no scan found");
  if (!have_qt[comp[0].tq]) return fail(err, "JPEG: scan refers to a missing table");
  const uint16_t* q = qt[comp[0].tq];
  if (sink) {  // coefficient output (see JpegCoefSink): the scans' result as it is
    sink->w = W;
    sink->h = H;
    if (sink->pitch_blocks && sink->pitch_blocks < lbw) return fail(err, "coefficient row pitch too small for this file");
    sink->blocks_w = sink->pitch_blocks ? sink->pitch_blocks : lbw;
    sink->blocks_rows = lbh;
    if ((size_t)sink->blocks_w * lbh > sink->cap_blocks) return fail(err, "frame larger than the coefficient buffer");
    for (int r = 0; r < lbh; r++)
      memcpy(sink->coef + (size_t)r * sink->blocks_w * 64, coef.data() + (size_t)r * lbw * 64, (size_t)lbw * 64 * sizeof(int16_t));
    for (int i = 0; i < 64; i++) sink->quant[i] = q[i];
    return true;
  }
  // coefficients -> samples: dequantise, islow IDCT, crop
  std::vector<unsigned char> rows((size_t)lbw * 8 * 8);
  int cf[64];
  for (int by = 0; by < lch; by++) {
    for (int bx = 0; bx < lcw; bx++) {
      const int16_t* blk = &coef[((size_t)by * lbw + bx) * 64];
      bool dc_only = true;
      for (int i = 0; i < 64; i++) {
        cf[i] = blk[i] * q[i];
        if (i && blk[i]) dc_only = false;
      }
      jpeg_idct_islow(cf, rows.data() + (size_t)bx * 8, (size_t)lbw * 8, dc_only);
    }
    const int y0 = by * 8, ny = std::min(8, H - y0);
    for (int r = 0; r < ny; r++) memcpy(out + (size_t)(y0 + r) * W, rows.data() + (size_t)r * lbw * 8, (size_t)W);
  }
  return true;
}

}  // namespace mdc_host
