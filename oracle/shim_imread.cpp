// TEST INFRASTRUCTURE ONLY -- cv::imread / cv::imdecode for the OpenCV stub.
//
// Decodes 8/16-bit PNG through libpng (an implementation independent of the
// product's own PNG reader, so the two check each other) and binary PGM (P5).
// Mirrors what cv::imread(..., CV_LOAD_IMAGE_UNCHANGED) hands to
// /root/reference/src/PhotometricUndistorter.cpp:120-147: a continuous
// single-channel matrix of type CV_8U or CV_16U with host-endian samples; a
// colour PNG comes back with a type that is neither (as OpenCV's CV_8UC3).
#include "opencv2/core/core.hpp"
#include <png.h>
#include <cstdio>

namespace cv {

static Mat read_pgm(FILE* f) {
  Mat m;
  int w = 0, h = 0, maxv = 0;
  if (fscanf(f, "P5 %d %d %d", &w, &h, &maxv) != 3) return m;
  fgetc(f);
  int bps = maxv > 255 ? 2 : 1;
  m.store = std::make_shared<std::vector<uchar>>((size_t)w * h * bps);
  if (fread(m.store->data(), 1, m.store->size(), f) != m.store->size()) return Mat();
  if (bps == 2) {  // PGM is big-endian
    uchar* p = m.store->data();
    for (size_t i = 0; i < (size_t)w * h; i++) std::swap(p[2 * i], p[2 * i + 1]);
  }
  m.rows = h; m.cols = w; m.data = m.store->data();
  m.type_ = bps == 2 ? CV_16U : CV_8U;
  return m;
}

Mat imread(const std::string& file, int /*flags*/) {
  Mat m;
  FILE* f = fopen(file.c_str(), "rb");
  if (!f) return m;
  unsigned char sig[8];
  if (fread(sig, 1, 8, f) != 8) { fclose(f); return m; }
  if (sig[0] == 'P' && sig[1] == '5') {
    rewind(f);
    m = read_pgm(f);
    fclose(f);
    return m;
  }
  if (png_sig_cmp(sig, 0, 8)) { fclose(f); return m; }
  png_structp png = png_create_read_struct(PNG_LIBPNG_VER_STRING, 0, 0, 0);
  png_infop info = png_create_info_struct(png);
  if (setjmp(png_jmpbuf(png))) {
    png_destroy_read_struct(&png, &info, 0);
    fclose(f);
    return Mat();
  }
  png_init_io(png, f);
  png_set_sig_bytes(png, 8);
  png_read_info(png, info);
  int w = png_get_image_width(png, info), h = png_get_image_height(png, info);
  int depth = png_get_bit_depth(png, info), ct = png_get_color_type(png, info);
  if (depth < 8) { png_set_packing(png); depth = 8; }
  if (depth == 16) png_set_swap(png);  // to little-endian host order
  png_set_interlace_handling(png);
  png_read_update_info(png, info);
  size_t rb = png_get_rowbytes(png, info);
  m.store = std::make_shared<std::vector<uchar>>(rb * h);
  std::vector<png_bytep> rows(h);
  for (int y = 0; y < h; y++) rows[y] = m.store->data() + rb * y;
  png_read_image(png, rows.data());
  png_destroy_read_struct(&png, &info, 0);
  fclose(f);
  m.rows = h; m.cols = w; m.data = m.store->data();
  if (ct == PNG_COLOR_TYPE_GRAY) m.type_ = depth == 16 ? CV_16U : CV_8U;
  else m.type_ = CV_8UC3;  // anything the reference does not handle
  return m;
}

// cv::imdecode(cv::Mat(readbytes, 1, CV_8U, databuffer), GRAYSCALE) of src/BenchmarkDatasetReader.h:274: the bytes of
// one archive entry.  PNG (libpng reading from memory) and binary PGM; other formats (JPEG: its exact output depends on
// the libjpeg build OpenCV links) come back empty, as an undecodable file does in OpenCV.
namespace {
struct MemCursor {
  const uchar* p;
  size_t left;
};
void mem_read(png_structp png, png_bytep out, png_size_t n) {
  MemCursor* c = (MemCursor*)png_get_io_ptr(png);
  if (n > c->left) png_error(png, "read past the end");
  memcpy(out, c->p, n);
  c->p += n;
  c->left -= n;
}
}  // namespace

Mat imdecode(const Mat& buf, int /*flags*/) {
  Mat m;
  const size_t n = (size_t)buf.rows * (size_t)buf.cols;
  if (!buf.data || n < 16) return m;
  if (buf.data[0] == 'P' && buf.data[1] == '5') {
    int w = 0, h = 0, maxv = 0, used = 0;
    std::string head((const char*)buf.data, std::min<size_t>(n, 64));
    if (sscanf(head.c_str(), "P5 %d %d %d%n", &w, &h, &maxv, &used) != 3 || maxv > 255) return m;
    if ((size_t)used + 1 + (size_t)w * h > n) return m;
    m.store = std::make_shared<std::vector<uchar>>(buf.data + used + 1, buf.data + used + 1 + (size_t)w * h);
    m.rows = h; m.cols = w; m.data = m.store->data(); m.type_ = CV_8U;
    return m;
  }
  if (png_sig_cmp(buf.data, 0, 8)) return m;
  png_structp png = png_create_read_struct(PNG_LIBPNG_VER_STRING, 0, 0, 0);
  png_infop info = png_create_info_struct(png);
  MemCursor cur = {buf.data, n};
  if (setjmp(png_jmpbuf(png))) {
    png_destroy_read_struct(&png, &info, 0);
    return Mat();
  }
  png_set_read_fn(png, &cur, mem_read);
  png_read_info(png, info);
  const int w = png_get_image_width(png, info), h = png_get_image_height(png, info);
  const int depth = png_get_bit_depth(png, info), ct = png_get_color_type(png, info);
  if (ct != PNG_COLOR_TYPE_GRAY || depth != 8) {  // GRAYSCALE load of anything else would need OpenCV's conversions
    png_destroy_read_struct(&png, &info, 0);
    return Mat();
  }
  png_set_interlace_handling(png);
  png_read_update_info(png, info);
  const size_t rb = png_get_rowbytes(png, info);
  m.store = std::make_shared<std::vector<uchar>>(rb * h);
  std::vector<png_bytep> rows(h);
  for (int y = 0; y < h; y++) rows[y] = m.store->data() + rb * y;
  png_read_image(png, rows.data());
  png_destroy_read_struct(&png, &info, 0);
  m.rows = h; m.cols = w; m.data = m.store->data(); m.type_ = CV_8U;
  return m;
}

}  // namespace cv
