// C facade over the drop-in classes -- see include/mdc_host.h.
#include "mdc_host.h"

#include <new>
#include <cstdint>
#include <cstring>

#include "BenchmarkDatasetReader.h"
#include "FOVUndistorter.h"
#include "image_codecs.h"
#include "image_codecs_internal.h"
#include "MdcBind.h"
#include "PhotometricUndistorter.h"

struct mdch_fov { UndistorterFOV* u; };
struct mdch_photo { PhotometricUndistorter* p; };

// Friend of both classes: read-only view of their tables for the facade.
struct MdcHostAccess {
  static const float* rx(const UndistorterFOV& u) { return u.remap_x_; }
  static const float* ry(const UndistorterFOV& u) { return u.remap_y_; }
  static const float* calib_out(const UndistorterFOV& u) { return u.calib_out_; }
  static const float* calib_in(const UndistorterFOV& u) { return u.calib_in_; }
  static bool has_gpu(const UndistorterFOV& u) { return u.gpu_ != 0; }
  static bool has_gpu(const PhotometricUndistorter& p) { return p.gpu_ != 0; }
  static bool valid_gamma(const PhotometricUndistorter& p) { return p.valid_gamma_; }
  static bool valid_vignette(const PhotometricUndistorter& p) { return p.valid_vignette_; }
  static const float* vmap(const PhotometricUndistorter& p) { return p.vignette_; }
  static const float* vinv(const PhotometricUndistorter& p) { return p.vignette_inv_; }
  static const float* ginv(const PhotometricUndistorter& p) { return p.ginv_; }
  static int w(const PhotometricUndistorter& p) { return p.w_; }
  static int h(const PhotometricUndistorter& p) { return p.h_; }
};

int mdc_bind_objects(mdc_ctx* ctx, const UndistorterFOV* fov, const PhotometricUndistorter* photo) {
  if (!ctx) return MDC_ERR_ARG;
  int rc = MDC_OK;
  if (photo) {
    const PhotometricUndistorter& p = *photo;
    rc = mdc_set_photometric(ctx, MdcHostAccess::valid_gamma(p) ? MdcHostAccess::ginv(p) : 0,
                             MdcHostAccess::valid_vignette(p) ? MdcHostAccess::vinv(p) : 0, MdcHostAccess::w(p),
                             MdcHostAccess::h(p));
    if (rc != MDC_OK) return rc;
  }
  if (fov) {
    const UndistorterFOV& u = *fov;
    if (u.isValid())
      rc = mdc_set_remap(ctx, MdcHostAccess::rx(u), MdcHostAccess::ry(u), u.getInputDims()[0], u.getInputDims()[1],
                         u.getOutputDims()[0], u.getOutputDims()[1]);
    else rc = mdc_set_remap(ctx, 0, 0, 0, 0, 0, 0);
  }
  return rc;
}

void mdc_fov_model_of(const UndistorterFOV& u, mdc_fov_model* m) {
  for (int i = 0; i < 5; i++) {
    m->in_calib[i] = MdcHostAccess::calib_in(u)[i];
    m->out_calib[i] = MdcHostAccess::calib_out(u)[i];
  }
  m->in_w = u.getInputDims()[0];
  m->in_h = u.getInputDims()[1];
  m->out_w = u.getOutputDims()[0];
  m->out_h = u.getOutputDims()[1];
}

extern "C" {

void mdch_fov_model(const mdch_fov* h, mdc_fov_model* m) try { mdc_fov_model_of(*h->u, m); } catch (...) {}

mdch_fov* mdch_fov_create(const char* camera_txt) try {
  mdch_fov* h = new mdch_fov;
  h->u = new UndistorterFOV(camera_txt);
  return h;
} catch (...) { return {}; }  // no exception leaves the C facade
void mdch_fov_destroy(mdch_fov* h) try {
  if (!h) return;
  delete h->u;
  delete h;
} catch (...) {}
int mdch_fov_valid(const mdch_fov* h) try { return h->u->isValid() ? 1 : 0; } catch (...) { return {}; }
int mdch_fov_has_gpu(const mdch_fov* h) try { return MdcHostAccess::has_gpu(*h->u) ? 1 : 0; } catch (...) { return {}; }
void mdch_fov_dims(const mdch_fov* h, int d[4]) try {
  d[0] = h->u->getInputDims()[0];
  d[1] = h->u->getInputDims()[1];
  d[2] = h->u->getOutputDims()[0];
  d[3] = h->u->getOutputDims()[1];
} catch (...) {}
void mdch_fov_intrinsics(const mdch_fov* h, float* o) try {
  const Eigen::Matrix3f a = h->u->getK_rect(), b = h->u->getK_org();
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      o[r * 3 + c] = a(r, c);
      o[9 + r * 3 + c] = b(r, c);
    }
  const Eigen::VectorXf v = h->u->getOriginalCalibration();
  for (int i = 0; i < 5; i++) o[18 + i] = v[i];
  o[23] = h->u->getOmega();
  for (int i = 0; i < 5; i++) o[24 + i] = MdcHostAccess::calib_out(*h->u)[i];
} catch (...) {}
int mdch_fov_remap(const mdch_fov* h, float* rx, float* ry) try {
  if (!h->u->isValid() || !MdcHostAccess::rx(*h->u)) return 0;
  const size_t n = (size_t)h->u->getOutputDims()[0] * h->u->getOutputDims()[1];
  memcpy(rx, MdcHostAccess::rx(*h->u), n * sizeof(float));
  memcpy(ry, MdcHostAccess::ry(*h->u), n * sizeof(float));
  return 1;
} catch (...) { return {}; }  // no exception leaves the C facade
void mdch_fov_distort(mdch_fov* h, float* x, float* y, int n) try { h->u->distortCoordinates(x, y, n); } catch (...) {}
void mdch_fov_undistort_f32(const mdch_fov* h, const float* in, float* out, int n_in, int n_out) try {
  h->u->undistort<float>(in, out, n_in, n_out);
} catch (...) {}
void mdch_fov_undistort_u8(const mdch_fov* h, const unsigned char* in, float* out, int n_in, int n_out) try {
  h->u->undistort<unsigned char>(in, out, n_in, n_out);
} catch (...) {}

mdch_photo* mdch_photo_create(const char* pcalib, const char* vignette, int w, int h) try {
  mdch_photo* p = new mdch_photo;
  p->p = new PhotometricUndistorter(pcalib, vignette, w, h);
  return p;
} catch (...) { return {}; }  // no exception leaves the C facade
void mdch_photo_destroy(mdch_photo* p) try {
  if (!p) return;
  delete p->p;
  delete p;
} catch (...) {}
int mdch_photo_valid(const mdch_photo* p) try {
  return (MdcHostAccess::valid_gamma(*p->p) ? 1 : 0) | (MdcHostAccess::valid_vignette(*p->p) ? 2 : 0);
} catch (...) { return {}; }  // no exception leaves the C facade
int mdch_photo_has_gpu(const mdch_photo* p) try { return MdcHostAccess::has_gpu(*p->p) ? 1 : 0; } catch (...) { return {}; }
int mdch_photo_ginv(mdch_photo* p, float* o) try {
  const float* g = p->p->getGInv();
  if (!g) return 0;
  memcpy(o, g, 256 * sizeof(float));
  return 1;
} catch (...) { return {}; }  // no exception leaves the C facade
int mdch_photo_g(mdch_photo* p, float* o) try {
  const float* g = p->p->getG();
  if (!g) return 0;
  memcpy(o, g, 256 * sizeof(float));
  return 1;
} catch (...) { return {}; }  // no exception leaves the C facade
int mdch_photo_vignette(const mdch_photo* p, float* map, float* inv) try {
  if (!MdcHostAccess::valid_vignette(*p->p)) return 0;
  const size_t n = (size_t)MdcHostAccess::w(*p->p) * MdcHostAccess::h(*p->p);
  if (map) memcpy(map, MdcHostAccess::vmap(*p->p), n * sizeof(float));
  if (inv) memcpy(inv, MdcHostAccess::vinv(*p->p), n * sizeof(float));
  return 1;
} catch (...) { return {}; }  // no exception leaves the C facade
void mdch_photo_unmap(mdch_photo* p, unsigned char* in, float* out, int n, int g, int v, int o) try {
  p->p->unMapImage(in, out, n, g != 0, v != 0, o != 0);
} catch (...) {}

int mdch_bind(mdc_ctx* ctx, const mdch_fov* fov, const mdch_photo* photo) try {
  return mdc_bind_objects(ctx, fov ? fov->u : 0, photo ? photo->p : 0);
} catch (const std::bad_alloc&) { return MDC_ERR_NOMEM; } catch (...) { return MDC_ERR_HIP; }  // a status, never "0 = ok", for a failed call

// Must stay in step with BlobHeader in csrc/mdc_capi.hip (checked by tests/test_multi_gpu.py
// on the GPU: pack == export after bind).
namespace {
struct PackedHeader {
  uint32_t magic, version;
  int32_t in_w, in_h, rm_in_w, rm_in_h, out_w, out_h;
  int32_t valid_gamma, valid_vignette, valid_remap, pad;
};
}  // namespace

int mdch_pack_tables(const mdch_fov* fov, const mdch_photo* photo, void* blob, size_t cap, size_t* size) try {
  if (!size) return MDC_ERR_ARG;
  PackedHeader h;
  memset(&h, 0, sizeof h);
  h.magic = 0x4d444331u;
  h.version = 1;
  const PhotometricUndistorter* p = photo ? photo->p : 0;
  const UndistorterFOV* u = fov ? fov->u : 0;
  if (p) {
    h.in_w = MdcHostAccess::w(*p);
    h.in_h = MdcHostAccess::h(*p);
    h.valid_gamma = MdcHostAccess::valid_gamma(*p);
    h.valid_vignette = MdcHostAccess::valid_vignette(*p);
  }
  if (u && u->isValid()) {
    h.valid_remap = 1;
    h.rm_in_w = u->getInputDims()[0];
    h.rm_in_h = u->getInputDims()[1];
    h.out_w = u->getOutputDims()[0];
    h.out_h = u->getOutputDims()[1];
  }
  const size_t nv = h.valid_vignette ? (size_t)h.in_w * h.in_h : 0;
  const size_t nr = h.valid_remap ? (size_t)h.out_w * h.out_h : 0;
  const size_t need = sizeof h + 256 * 4 + nv * 4 + 2 * nr * 4;
  *size = need;
  if (!blob) return MDC_OK;
  if (cap < need) return MDC_ERR_ARG;
  char* q = (char*)blob;
  memcpy(q, &h, sizeof h);
  q += sizeof h;
  float zeros[256];
  memset(zeros, 0, sizeof zeros);
  memcpy(q, (p && h.valid_gamma) ? MdcHostAccess::ginv(*p) : zeros, 256 * 4);
  q += 256 * 4;
  if (nv) memcpy(q, MdcHostAccess::vinv(*p), nv * 4);
  q += nv * 4;
  if (nr) {
    memcpy(q, MdcHostAccess::rx(*u), nr * 4);
    memcpy(q + nr * 4, MdcHostAccess::ry(*u), nr * 4);
  }
  return MDC_OK;
} catch (const std::bad_alloc&) { return MDC_ERR_NOMEM; } catch (...) { return MDC_ERR_HIP; }  // (mdch_pack_tables: *size / the blob may be unwritten)


// ---- DatasetReader ---------------------------------------------------------------------------------
struct mdch_reader { DatasetReader* r; };

mdch_reader* mdch_reader_create(const char* folder) try {
  mdch_reader* h = new mdch_reader;
  h->r = new DatasetReader(folder);
  return h;
} catch (...) { return {}; }  // no exception leaves the C facade
void mdch_reader_destroy(mdch_reader* h) try {
  if (!h) return;
  delete h->r;
  delete h;
} catch (...) {}
int mdch_reader_num_images(mdch_reader* h) try { return h->r->getNumImages(); } catch (...) { return {}; }
double mdch_reader_timestamp(mdch_reader* h, int id) try { return h->r->getTimestamp(id); } catch (...) { return {}; }
float mdch_reader_exposure(mdch_reader* h, int id) try { return h->r->getExposure(id); } catch (...) { return {}; }
void mdch_reader_dims(mdch_reader* h, int d[4]) try {
  d[0] = h->r->getUndistorter()->getInputDims()[0];
  d[1] = h->r->getUndistorter()->getInputDims()[1];
  d[2] = h->r->getUndistorter()->getOutputDims()[0];
  d[3] = h->r->getUndistorter()->getOutputDims()[1];
} catch (...) {}
int mdch_reader_get_image(mdch_reader* h, int id, int rectify, int g, int v, int o, float* out, long cap, int meta[3],
                          double* stamp, float* exposure) try {
  ExposureImage* img = h->r->getImage(id, rectify != 0, g != 0, v != 0, o != 0);
  if (!img) return 0;
  const long n = (long)img->w * img->h;
  const int ok = n <= cap;
  if (ok) memcpy(out, img->image, (size_t)n * sizeof(float));
  meta[0] = img->w;
  meta[1] = img->h;
  meta[2] = img->id;
  *stamp = img->timestamp;
  *exposure = img->exposure_time;
  delete img;  // caller owns the ExposureImage (main_playbackDataset.cpp:82,116)
  return ok;
} catch (...) { return {}; }  // no exception leaves the C facade
int mdch_reader_get_images_device(mdch_reader* h, int first, int count, int rectify, int g, int v, int o, const mdc_device_outputs* out,
                                  unsigned char* valid) try {
  return h->r->getImagesDevice(first, count, rectify != 0, g != 0, v != 0, o != 0, out, valid);
} catch (...) { return 0; }
mdc_ctx* mdch_reader_context(mdch_reader* h) { return h->r->getContext(); }
int mdch_reader_device(mdch_reader* h) { return h->r->getDevice(); }
int mdch_reader_get_images(mdch_reader* h, int first, int count, int rectify, int g, int v, int o, float* out,
                           long frame_floats, unsigned char* ok) try {
  if (count <= 0) return 0;
  ExposureImage** imgs = new ExposureImage*[count];
  const int n = h->r->getImages(first, count, rectify != 0, g != 0, v != 0, o != 0, imgs);
  for (int i = 0; i < count; i++) {
    ok[i] = 0;
    if (!imgs[i]) continue;
    if ((long)imgs[i]->w * imgs[i]->h <= frame_floats && imgs[i]->id == first + i) {
      memcpy(out + (size_t)i * frame_floats, imgs[i]->image, (size_t)imgs[i]->w * imgs[i]->h * sizeof(float));
      ok[i] = 1;
    }
    delete imgs[i];
  }
  delete[] imgs;
  return n;
} catch (...) { return {}; }  // no exception leaves the C facade
int mdch_reader_get_raw(mdch_reader* h, int id, unsigned char* out, long cap, int wh[2]) try {
  const unsigned char* p = h->r->getImageRaw(id, &wh[0], &wh[1]);
  if (!p || (long)wh[0] * wh[1] > cap) return 0;
  memcpy(out, p, (size_t)wh[0] * wh[1]);
  return 1;
} catch (...) { return {}; }  // no exception leaves the C facade
void mdch_reader_set_threads(mdch_reader* h, int n) try { h->r->setDecodeThreads(n); } catch (...) {}
void mdch_reader_set_prefetch(mdch_reader* h, int n) try { h->r->setPrefetch(n); } catch (...) {}
void mdch_reader_set_lookahead(mdch_reader* h, int frames) try { h->r->setResultLookahead(frames); } catch (...) {}
void mdch_reader_set_gpu_jpeg(mdch_reader* h, int stage) try { h->r->setGpuJpegStage(stage == 1 ? 1 : (stage ? 2 : 0)); } catch (...) {}
const char* mdch_reader_last_error(mdch_reader* h) try { return h->r->lastError(); } catch (...) { return {}; }
void mdch_reader_prefetch_stats(mdch_reader* h, long hm[2]) try { h->r->getPrefetchStats(&hm[0], &hm[1]); } catch (...) {}
int mdch_reader_device_stats(mdch_reader* h, int lane, int64_t device_frames[2], double wait_gpu_seconds[2]) try {
  int dev = -1;
  long frames = 0;
  if (lane < 0 || lane >= h->r->getDeviceCount()) return 0;
  h->r->getDeviceStats(lane, &dev, &frames, &wait_gpu_seconds[0], &wait_gpu_seconds[1]);
  device_frames[0] = dev;
  device_frames[1] = frames;
  return 1;
} catch (...) { return {}; }

size_t mdch_jpeg_record_bytes(int w, int h, int pitch_rows[2]) try {
  // MCUs are 1..4 x 1..4 blocks: a pitch / row count rounded up to a multiple of 12 blocks (lcm of 1, 2, 3, 4) holds every
  // sampling layout, a luma factor of 3 included
  const int pitch = ((w + 7) / 8 + 11) / 12 * 12, rows = ((h + 7) / 8 + 11) / 12 * 12;
  if (pitch_rows) {
    pitch_rows[0] = pitch;
    pitch_rows[1] = rows;
  }
  return 128 + (size_t)pitch * rows * 128;
} catch (...) { return {}; }  // no exception leaves the C facade

int mdch_decode_jpeg_record(const unsigned char* data, size_t n, void* record, size_t record_bytes, int pitch_blocks, int dims[4], char* err,
                            size_t errcap) try {
  std::string e;
  bool ok = false;
  if (record && record_bytes >= 128 + 128 && dims) {
    mdc_host::JpegCoefSink sink;
    sink.coef = reinterpret_cast<int16_t*>(static_cast<unsigned char*>(record) + 128);
    sink.cap_blocks = (record_bytes - 128) / 128;
    sink.pitch_blocks = pitch_blocks;
    ok = mdc_host::decode_jpeg_coefs(data, n, &sink, &e);
    if (ok) {
      memcpy(record, sink.quant, 128);
      dims[0] = sink.w;
      dims[1] = sink.h;
      dims[2] = sink.blocks_w;
      dims[3] = sink.blocks_rows;
    }
  } else {
    e = "bad record buffer";
  }
  if (err && errcap) {
    strncpy(err, e.c_str(), errcap - 1);
    err[errcap - 1] = 0;
  }
  return ok ? 1 : 0;
} catch (...) { return {}; }  // no exception leaves the C facade

long long mdch_jpeg_stream(const unsigned char* data, size_t n, void* stream, size_t cap, int wh[2], char* err, size_t errcap) try {
  std::string e;
  size_t used = 0;
  int w = 0, h = 0;
  const bool ok = data && stream && wh && mdc_host::jpeg_stream(data, n, static_cast<unsigned char*>(stream), cap, &used, &w, &h, &e);
  if (ok) {
    wh[0] = w;
    wh[1] = h;
  } else if (e.empty()) {
    e = "bad argument";
  }
  if (err && errcap) {
    strncpy(err, e.c_str(), errcap - 1);
    err[errcap - 1] = 0;
  }
  return ok ? (long long)used : 0;
} catch (...) { return {}; }  // no exception leaves the C facade

int mdch_decode_gray8(const unsigned char* data, size_t n, unsigned char* out, size_t cap, int wh[2], char* err, size_t errcap) try {
  std::string e;
  const bool ok = mdc_host::decode_gray8(data, n, out, cap, &wh[0], &wh[1], &e);
  if (err && errcap) {
    strncpy(err, e.c_str(), errcap - 1);
    err[errcap - 1] = 0;
  }
  return ok ? 1 : 0;
} catch (...) { return {}; }  // no exception leaves the C facade

}  // extern "C"
