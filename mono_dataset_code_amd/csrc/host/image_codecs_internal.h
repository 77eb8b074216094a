// Shared between image_codecs.cpp (the common formats, decoded in place) and image_codecs_ext.cpp (every other PNG
// flavour, progressive JPEG) and gray_png.cpp (the vignette image).  Not installed.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace mdc_host {

struct PngAny {
  int w = 0, h = 0;
  int channels = 0;  // 1 gray, 2 gray + alpha, 3 RGB (also expanded palettes), 4 RGBA
  int bits = 0;      // 8 (1/2/4-bit gray already scaled to 8) or 16
  bool palette = false;
  std::vector<uint16_t> px;  // interleaved samples
};
bool png_decode_any(const unsigned char* data, size_t n, PngAny& im, std::string* err);
void png_any_to_gray8(const PngAny& im, unsigned char* out);  // w*h bytes, OpenCV's grayscale read of that PNG

// JPEG entropy decoding WITHOUT the inverse DCT: the quantised luma coefficients of a file, natural order, 64 int16 per
// 8x8 block, [blocks_rows][blocks_w][64] (the grid is padded to whole MCUs; the image covers the first ceil(h/8) rows and
// ceil(w/8) blocks of a row), + the luma quantisation table (natural order).  What the GPU stage of the reader takes
// (mdc_jpeg_idct_batch_device dequantises, inverts and crops on the device).  `coef` / `cap_blocks` are the caller's.
struct JpegCoefSink {
  int16_t* coef = nullptr;
  size_t cap_blocks = 0;
  int pitch_blocks = 0;  // in: row pitch in blocks the caller wants (0 = the file's own MCU-padded width)
  uint16_t quant[64];
  int w = 0, h = 0, blocks_w = 0, blocks_rows = 0;  // out: blocks_w = the row pitch used
};
bool decode_jpeg_coefs(const unsigned char* data, size_t n, JpegCoefSink* sink, std::string* err);

// GPU Huffman stage (include/mdc_hip.h: mdc_jpeg_stream_header): markers parsed, decode tables built, entropy-coded segment
// copied without its byte stuffing into `stream` (capacity cap) -- header, ecs bytes, 16 zero bytes; *used = bytes written.
// false (err says why) for what the device decoder does not take: more than one component, progressive / arithmetic /
// lossless files, restart intervals, 16-bit quantisation values above what a record holds, a stream that does not fit.
bool jpeg_stream(const unsigned char* data, size_t n, unsigned char* stream, size_t cap, size_t* used, int* w, int* h, std::string* err);

// Colour space of a three-component file as libjpeg decides it (jdapimin.c: default_decompress_parms): a JFIF marker means YCbCr;
// else an Adobe marker's transform byte (0 = RGB, 1 = YCbCr); else the component ids (1 2 3 = YCbCr, 'R' 'G' 'B' = RGB); else YCbCr.
// For an RGB file cv::imread(..., GRAYSCALE) returns 0.299 R + 0.587 G + 0.114 B (libjpeg's rgb_gray_convert), not component 0.
struct JpegColorMarkers {
  bool jfif = false, adobe = false;
  int adobe_transform = 0;
  void see(int marker, const unsigned char* s, size_t sl) {
    if (marker == 0xe0 && sl >= 14 && s[0] == 'J' && s[1] == 'F' && s[2] == 'I' && s[3] == 'F' && s[4] == 0) jfif = true;
    if (marker == 0xee && sl >= 12 && s[0] == 'A' && s[1] == 'd' && s[2] == 'o' && s[3] == 'b' && s[4] == 'e') {
      adobe = true;
      adobe_transform = s[11];
    }
  }
  bool is_rgb(int ncomp, int id0, int id1, int id2) const {
    if (ncomp != 3) return false;
    if (jfif) return false;
    if (adobe) return adobe_transform == 0;
    return id0 == 'R' && id1 == 'G' && id2 == 'B';
  }
};

bool jpeg_progressive_gray8(const unsigned char* data, size_t n, unsigned char* out, size_t cap, int* w, int* h, std::string* err,
                            JpegCoefSink* sink = nullptr);
// libjpeg's islow inverse DCT on dequantised coefficients in natural order (image_codecs.cpp)
void jpeg_idct_islow(const int* coef, unsigned char* out, size_t stride, bool dc_only);

}  // namespace mdc_host
