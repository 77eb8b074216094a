/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the photometric + FOV undistortion
 * hot path of tum-vision/mono_dataset_code.
 *
 * A plain-C restatement of the reference's algorithm, one function per
 * reference function, each citing the reference file:line it follows
 * (paths relative to /root/reference/).  It exists so that parity tests have a
 * checker that travels to the GPU box (where /root/reference is absent).
 *
 * Pinning: the reference ships no tests or golden vectors for this path
 * (SURVEY.md section 4), so this restatement is pinned against the reference
 * ITSELF: tests/test_oracle_vs_ref.py compares every function here, bit for
 * bit, with oracle/_ref/libmdc_ref.so (the reference's sources compiled where
 * they lie), and tests/golden/ holds vectors generated from that library by
 * tests/golden/make_golden.py.  The pyramid (orc_pyramid_level) has no
 * counterpart in the reference: PARITY UNPINNED for that function only.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this.  The product never does.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off (x86-64 baseline: SSE2 scalar
 * float, no FMA, FLT_EVAL_METHOD == 0 -- the arithmetic the reference's
 * CMake flag set "-O3 -DNDEBUG -std=c++0x" produces, CMakeLists.txt:16-18).
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAXF(a, b) (((a) < (b)) ? (b) : (a)) /* std::max(a,b) */

/* ------------------------------------------------------------------------ */
/* FOV model: src/FOVUndistorter.cpp:280-319 (UndistorterFOV::distortCoordinates) */
/* in_calib = fx fy cx cy omega relative to the input size (camera.txt line 1);  */
/* out_calib = the *normalised* output calibration the ctor leaves behind (:214-218). */
void orc_distort_coordinates(const float in_calib[5], int in_w, int in_h, const float out_calib[5],
                             int out_w, int out_h, float* xs, float* ys, int n) {
  float dist = in_calib[4];
  float d2t = 2.0f * tan(dist / 2.0f); /* double tan, narrowed: :290 */

  float fx = in_calib[0] * in_w; /* :293-296 */
  float fy = in_calib[1] * in_h;
  float cx = in_calib[2] * in_w - 0.5;
  float cy = in_calib[3] * in_h - 0.5;

  float ofx = out_calib[0] * out_w; /* :298-301 */
  float ofy = out_calib[1] * out_h;
  float ocx = out_calib[2] * out_w - 0.5f;
  float ocy = out_calib[3] * out_h - 0.5f;

  for (int i = 0; i < n; i++) { /* :303-318 */
    float x = xs[i];
    float y = ys[i];
    float ix = (x - ocx) / ofx;
    float iy = (y - ocy) / ofy;

    float r = sqrtf(ix * ix + iy * iy);
    float fac = (r == 0 || dist == 0) ? 1 : atanf(r * d2t) / (dist * r);

    ix = fx * fac * ix + cx;
    iy = fy * fac * iy + cy;

    xs[i] = ix;
    ys[i] = iy;
  }
}

/*
 * Output-intrinsics selection + remap build: src/FOVUndistorter.cpp:128-268.
 *   mode: -1 = "crop" (:151), -2 = "full" (:173), 0 = explicit out_calib_in (:206).
 *   out_calib (5): receives the normalised output calibration (:214-218).
 *   remap_x/remap_y (out_w*out_h): source coordinates, (-1,-1) = black (:223-251).
 *   k_rect/k_org (9, row-major): :257-268.
 * Returns 1 if any black pixel was produced (the reference prints a warning), else 0.
 */
int orc_fov_setup(const float in_calib[5], int in_w, int in_h, int mode, const float out_calib_in[5],
                  int out_w, int out_h, float out_calib[5], float* remap_x, float* remap_y,
                  float k_rect[9], float k_org[9]) {
  float dist = in_calib[4];
  float d2t = 2.0f * tan(dist / 2.0f); /* :132 */

  float fx = in_calib[0] * in_w; /* :135-138 */
  float fy = in_calib[1] * in_h;
  float cx = in_calib[2] * in_w - 0.5;
  float cy = in_calib[3] * in_h - 0.5;

  float ofx, ofy, ocx, ocy;

  if (in_calib[4] == 0) { /* :144-150 */
    ofx = in_calib[0] * out_w;
    ofy = in_calib[1] * out_h;
    ocx = (in_calib[2] * out_w) - 0.5;
    ocy = (in_calib[3] * out_h) - 0.5;
  } else if (mode == -1) { /* crop :151-172 */
    float left_radius = (cx) / fx;
    float right_radius = (in_w - 1 - cx) / fx;
    float top_radius = (cy) / fy;
    float bottom_radius = (in_h - 1 - cy) / fy;

    float trans_left_radius = tan(left_radius * dist) / d2t;
    float trans_right_radius = tan(right_radius * dist) / d2t;
    float trans_top_radius = tan(top_radius * dist) / d2t;
    float trans_bottom_radius = tan(bottom_radius * dist) / d2t;

    ofy = fy * ((top_radius + bottom_radius) / (trans_top_radius + trans_bottom_radius)) *
          ((float)out_h / (float)in_h);
    ocy = (trans_top_radius / top_radius) * ofy * cy / fy;

    ofx = fx * ((left_radius + right_radius) / (trans_left_radius + trans_right_radius)) *
          ((float)out_w / (float)in_w);
    ocx = (trans_left_radius / left_radius) * ofx * cx / fx;
  } else if (mode == -2) { /* full :173-205 */
    float left_radius = cx / fx;
    float right_radius = (in_w - 1 - cx) / fx;
    float top_radius = cy / fy;
    float bottom_radius = (in_h - 1 - cy) / fy;

    float tl_radius = sqrt(left_radius * left_radius + top_radius * top_radius);
    float tr_radius = sqrt(right_radius * right_radius + top_radius * top_radius);
    float bl_radius = sqrt(left_radius * left_radius + bottom_radius * bottom_radius);
    float br_radius = sqrt(right_radius * right_radius + bottom_radius * bottom_radius);

    float trans_tl_radius = tan(tl_radius * dist) / d2t;
    float trans_tr_radius = tan(tr_radius * dist) / d2t;
    float trans_bl_radius = tan(bl_radius * dist) / d2t;
    float trans_br_radius = tan(br_radius * dist) / d2t;

    float hor = ORC_MAXF(br_radius, tr_radius) + ORC_MAXF(bl_radius, tl_radius);
    float vert = ORC_MAXF(tr_radius, tl_radius) + ORC_MAXF(bl_radius, br_radius);

    float trans_hor = ORC_MAXF(trans_br_radius, trans_tr_radius) + ORC_MAXF(trans_bl_radius, trans_tl_radius);
    float trans_vert = ORC_MAXF(trans_tr_radius, trans_tl_radius) + ORC_MAXF(trans_bl_radius, trans_br_radius);

    ofy = fy * ((vert) / (trans_vert)) * ((float)out_h / (float)in_h);
    ocy = ORC_MAXF(trans_tl_radius / tl_radius, trans_tr_radius / tr_radius) * ofy * cy / fy;

    ofx = fx * ((hor) / (trans_hor)) * ((float)out_w / (float)in_w);
    ocx = ORC_MAXF(trans_bl_radius / bl_radius, trans_tl_radius / tl_radius) * ofx * cx / fx;
  } else { /* explicit :206-212 */
    ofx = out_calib_in[0] * out_w;
    ofy = out_calib_in[1] * out_h;
    ocx = out_calib_in[2] * out_w - 0.5;
    ocy = out_calib_in[3] * out_h - 0.5;
  }

  out_calib[0] = ofx / out_w; /* :214-218 */
  out_calib[1] = ofy / out_h;
  out_calib[2] = (ocx + 0.5) / out_w;
  out_calib[3] = (ocy + 0.5) / out_h;
  out_calib[4] = 0;

  for (int y = 0; y < out_h; y++) /* :226-231 */
    for (int x = 0; x < out_w; x++) {
      remap_x[x + y * out_w] = x;
      remap_y[x + y * out_w] = y;
    }
  orc_distort_coordinates(in_calib, in_w, in_h, out_calib, out_w, out_h, remap_x, remap_y,
                          out_h * out_w); /* :232 */

  int has_black = 0;
  for (int i = 0; i < out_w * out_h; i++) { /* :235-251 */
    if (remap_x[i] == 0) remap_x[i] = 0.01;
    if (remap_y[i] == 0) remap_y[i] = 0.01;
    if (remap_x[i] == in_w - 1) remap_x[i] = in_w - 1.01;
    if (remap_y[i] == in_h - 1) remap_y[i] = in_h - 1.01;

    if (!(remap_x[i] > 0 && remap_y[i] > 0 && remap_x[i] < in_w - 1 && remap_y[i] < in_h - 1)) {
      has_black = 1;
      remap_x[i] = -1;
      remap_y[i] = -1;
    }
  }

  for (int i = 0; i < 9; i++) k_rect[i] = k_org[i] = (i % 4 == 0) ? 1.f : 0.f; /* :257-268 */
  k_rect[0] = out_calib[0] * out_w;
  k_rect[4] = out_calib[1] * out_h;
  k_rect[2] = out_calib[2] * out_w - 0.5;
  k_rect[5] = out_calib[3] * out_h - 0.5;
  k_org[0] = in_calib[0] * in_w;
  k_org[4] = in_calib[1] * in_h;
  k_org[2] = in_calib[2] * in_w - 0.5;
  k_org[5] = in_calib[3] * in_h - 0.5;
  return has_black;
}

static int orc_getline(FILE* f, char* buf, int cap) {
  /* std::getline: reads up to '\n', drops it; a missing file line gives "" */
  int n = 0, c;
  while ((c = fgetc(f)) != EOF && c != '\n')
    if (n < cap - 1) buf[n++] = (char)c;
  buf[n] = 0;
  return n;
}

/*
 * camera.txt parsing: src/FOVUndistorter.cpp:54-123.
 * Returns 1 and fills everything when the object would be valid; 0 otherwise
 * (unreadable file :56, bad l1/l2 :78, "none" :96, unparsable l3 :107, bad l4 :119).
 * mode as in orc_fov_setup.
 */
int orc_parse_camera(const char* path, float in_calib[5], int* in_w, int* in_h, int* mode,
                     float out_calib_in[5], int* out_w, int* out_h) {
  FILE* f = fopen(path, "r");
  if (!f) return 0;
  char l1[1024], l2[1024], l3[1024], l4[1024];
  orc_getline(f, l1, 1024);
  orc_getline(f, l2, 1024);
  orc_getline(f, l3, 1024);
  orc_getline(f, l4, 1024);
  fclose(f);

  if (!(sscanf(l1, "%f %f %f %f %f", &in_calib[0], &in_calib[1], &in_calib[2], &in_calib[3], &in_calib[4]) == 5 &&
        sscanf(l2, "%d %d", in_w, in_h) == 2))
    return 0;

  *mode = 0;
  if (strcmp(l3, "crop") == 0) *mode = -1;
  else if (strcmp(l3, "full") == 0) *mode = -2;
  else if (strcmp(l3, "none") == 0) return 0;
  else if (sscanf(l3, "%f %f %f %f %f", &out_calib_in[0], &out_calib_in[1], &out_calib_in[2], &out_calib_in[3],
                  &out_calib_in[4]) != 5)
    return 0;

  if (sscanf(l4, "%d %d", out_w, out_h) != 2) return 0;
  return 1;
}

/* ------------------------------------------------------------------------ */
/*
 * Response table: src/PhotometricUndistorter.cpp:74-109.
 * raw[n] = the floats of pcalib.txt's first line.  Returns 1 (validGamma) and
 * fills ginv[256] / g[256]; returns 0 on n != 256 (:74) or a non-strictly-
 * increasing table (:81-88).  g[] entries the reference leaves uninitialised
 * (:94-106, no bracketing s) are left untouched here too.
 */
int orc_photo_gamma(const float* raw, int n, float ginv[256], float g[256]) {
  if (n != 256) return 0;
  for (int i = 0; i < 256; i++) ginv[i] = raw[i];
  for (int i = 0; i < 255; i++)
    if (ginv[i + 1] <= ginv[i]) return 0;
  float min = ginv[0];
  float max = ginv[255];
  for (int i = 0; i < 256; i++) ginv[i] = 255.0 * (ginv[i] - min) / (max - min); /* :91 */

  for (int i = 1; i < 255; i++) /* :94-106 */
    for (int s = 1; s < 255; s++)
      if (ginv[s] <= i && ginv[s + 1] >= i) {
        g[i] = s + (i - ginv[s]) / (ginv[s + 1] - ginv[s]);
        break;
      }
  g[0] = 0;
  g[255] = 255;
  return 1;
}

/* pcalib.txt first line -> floats (std::istream_iterator<float> semantics, :70-73):
 * whitespace separated, stops at the first token that is not a float. */
int orc_parse_pcalib(const char* path, float* raw, int cap) {
  FILE* f = fopen(path, "r");
  if (!f) return -1;
  static char line[1 << 16];
  orc_getline(f, line, sizeof line);
  fclose(f);
  int n = 0;
  char* p = line;
  while (n < cap) {
    char* e;
    float v = strtof(p, &e);
    if (e == p) break;
    raw[n++] = v;
    p = e;
  }
  return n;
}

/* Vignette normalisation: src/PhotometricUndistorter.cpp:130-152.
 * px is w*h samples, 8-bit (bits=8) or host-endian 16-bit (bits=16). */
void orc_photo_vignette(const void* px, int bits, int n, float* vmap, float* vinv) {
  float maxV = 0;
  if (bits == 8) {
    const unsigned char* p = (const unsigned char*)px;
    for (int i = 0; i < n; i++)
      if (p[i] > maxV) maxV = p[i];
    for (int i = 0; i < n; i++) vmap[i] = p[i] / maxV;
  } else {
    const unsigned short* p = (const unsigned short*)px;
    for (int i = 0; i < n; i++)
      if (p[i] > maxV) maxV = p[i];
    for (int i = 0; i < n; i++) vmap[i] = p[i] / maxV;
  }
  for (int i = 0; i < n; i++) vinv[i] = 1.0f / vmap[i];
}

/*
 * Per-frame photometric stage: src/PhotometricUndistorter.cpp:165-212.
 * valid_gamma / valid_vignette are the object's flags; g/v/o the call's.
 * Flag degradation as :173-189 (the reference's printf notices are omitted).
 */
void orc_unmap(const unsigned char* in, float* out, int n, const float ginv[256], const float* vinv,
               int valid_gamma, int valid_vignette, int g, int v, int o) {
  if (!valid_gamma && g) g = 0;
  if (!valid_vignette && v) v = 0;
  if (!g && v) { v = 0; g = 0; }

  if (!g && !v)
    for (int i = 0; i < n; i++) out[i] = in[i];
  if (g && !v)
    for (int i = 0; i < n; i++) out[i] = ginv[in[i]];
  if (g && v)
    for (int i = 0; i < n; i++) out[i] = ginv[in[i]] * vinv[i];
  if (o)
    for (int i = 0; i < n; i++)
      if (in[i] == 255) out[i] = NAN;
}

/* Per-frame geometric stage: src/FOVUndistorter.cpp:341-367 (T = float). */
void orc_undistort_f32(const float* input, float* output, const float* remap_x, const float* remap_y,
                       int in_w, int n_out) {
  for (int idx = 0; idx < n_out; idx++) {
    float xx = remap_x[idx];
    float yy = remap_y[idx];
    if (xx < 0) output[idx] = 0;
    else {
      int xxi = xx;
      int yyi = yy;
      xx -= xxi;
      yy -= yyi;
      float xxyy = xx * yy;
      const float* src = input + xxi + yyi * in_w;
      output[idx] = xxyy * src[1 + in_w] + (yy - xxyy) * src[in_w] + (xx - xxyy) * src[1] +
                    (1 - xx - yy + xxyy) * src[0];
    }
  }
}

/* Same, T = unsigned char (:370): taps promote exactly to float. */
void orc_undistort_u8(const unsigned char* input, float* output, const float* remap_x, const float* remap_y,
                      int in_w, int n_out) {
  for (int idx = 0; idx < n_out; idx++) {
    float xx = remap_x[idx];
    float yy = remap_y[idx];
    if (xx < 0) output[idx] = 0;
    else {
      int xxi = xx;
      int yyi = yy;
      xx -= xxi;
      yy -= yyi;
      float xxyy = xx * yy;
      const unsigned char* src = input + xxi + yyi * in_w;
      output[idx] = xxyy * src[1 + in_w] + (yy - xxyy) * src[in_w] + (xx - xxyy) * src[1] +
                    (1 - xx - yy + xxyy) * src[0];
    }
  }
}

/*
 * The composition site: DatasetReader::getImage, src/BenchmarkDatasetReader.h:207-241,
 * on a raw u8 frame.  tmp = internalTempBuffer (in_w*in_h floats, :145).
 * have_remap = UndistorterFOV::isValid(); an invalid undistorter leaves `out`
 * untouched when rectify is requested (FOVUndistorter.cpp:325).
 */
void orc_get_image(const unsigned char* raw, float* out, float* tmp, int in_w, int in_h, int out_w, int out_h,
                   const float ginv[256], const float* vinv, int valid_gamma, int valid_vignette,
                   const float* remap_x, const float* remap_y, int have_remap, int rectify, int g, int v, int o) {
  int n = in_w * in_h;
  if (g || v || o) {
    if (!rectify) orc_unmap(raw, out, n, ginv, vinv, valid_gamma, valid_vignette, g, v, o);
    else {
      orc_unmap(raw, tmp, n, ginv, vinv, valid_gamma, valid_vignette, g, v, o);
      if (have_remap) orc_undistort_f32(tmp, out, remap_x, remap_y, in_w, out_w * out_h);
    }
  } else {
    if (rectify) {
      if (have_remap) orc_undistort_u8(raw, out, remap_x, remap_y, in_w, out_w * out_h);
    } else
      for (int i = 0; i < n; i++) out[i] = raw[i];
  }
}

/*
 * One level of the box pyramid (BASELINE.json config 5).  NOT IN THE REFERENCE --
 * PARITY UNPINNED; this is the definition:
 * dst(x,y) = 0.25f * (((a+b)+c)+d), a=(2x,2y) b=(2x+1,2y) c=(2x,2y+1) d=(2x+1,2y+1),
 * dst is (w/2) x (h/2) (floor), NaN propagates.
 */
void orc_pyramid_level(const float* src, int w, int h, float* dst) {
  int w2 = w / 2, h2 = h / 2;
  for (int y = 0; y < h2; y++)
    for (int x = 0; x < w2; x++) {
      const float* p = src + 2 * x + (size_t)(2 * y) * w;
      dst[x + (size_t)y * w2] = 0.25f * (((p[0] + p[1]) + p[w]) + p[w + 1]);
    }
}

/* Synthetic frame generator of SURVEY.md section 8(d): uniform bytes from the
 * murmur3 finaliser of (seed + global pixel index). */
static unsigned orc_fmix32(unsigned h) {
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}
void orc_synth_frames(unsigned char* out, long long first_frame, long long nframes, int npix, unsigned seed) {
  for (long long f = 0; f < nframes; f++)
    for (int i = 0; i < npix; i++)
      out[f * npix + i] = (unsigned char)(orc_fmix32(seed + (unsigned)((first_frame + f) * (long long)npix + i)) >> 24);
}

/* CPU timing of orc_get_image over a batch (single thread; "port" baseline). */
#include <time.h>
double orc_time_path(const unsigned char* frames, int nframes, int passes, int in_w, int in_h, int out_w, int out_h,
                     const float ginv[256], const float* vinv, const float* remap_x, const float* remap_y,
                     int rectify, int g, int v, int o, double* checksum) {
  size_t nout = rectify ? (size_t)out_w * out_h : (size_t)in_w * in_h;
  float* tmp = (float*)malloc(sizeof(float) * in_w * in_h);
  float* out = (float*)malloc(sizeof(float) * nout);
  struct timespec t0, t1;
  double s = 0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int p = 0; p < passes; p++)
    for (int f = 0; f < nframes; f++) {
      orc_get_image(frames + (size_t)f * in_w * in_h, out, tmp, in_w, in_h, out_w, out_h, ginv, vinv, 1, 1, remap_x,
                    remap_y, 1, rectify, g, v, o);
      float x = out[(size_t)(f * 7919) % nout];
      if (x == x) s += x;
    }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  free(tmp);
  free(out);
  if (checksum) *checksum = s;
  return (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
}

/* ---------------------------------------------------------------------------------------------
 * vignetteCalib solver: the two accumulate half-iterations of src/main_vignetteCalib.cpp:395-527
 * (alternating least squares for the plane's colour and the per-pixel vignette factor).
 * Pinned bit for bit against the reference's own loop text (oracle/vcal_extract.py ->
 * oracle/_ref/libvcal_ref.so) in tests/test_vcal.py.
 * Images, plane->image coordinates and factors are stacked: images[img*wI*hI + ..],
 * p2x/p2y[img*np + pi], np = gw*gh plane points.  oth2 is the reference's int (:397-398).
 * ------------------------------------------------------------------------------------------- */
/* getInterpolatedElement, src/main_vignetteCalib.cpp:52-70 */
static float orc_interp(const float* mat, float x, float y, int width) {
  int ix = (int)x;
  int iy = (int)y;
  float dx = x - ix;
  float dy = y - iy;
  float dxdy = dx * dy;
  const float* bp = mat + ix + iy * width;
  float res = dxdy * bp[1 + width] + (dy - dxdy) * bp[width] + (dx - dxdy) * bp[1] + (1 - dx - dy + dxdy) * bp[0];
  return res;
}

/* "optimize planeColor", :400-448: FF/FC are rebuilt, planeColor is read (residuals) and then replaced. */
void orc_vcal_plane_step(const float* images, const float* p2x, const float* p2y, int n, int wI, int hI, int np,
                         float* planeColor, float* planeColorFF, float* planeColorFC, const float* vignetteFactor,
                         int oth2, double* E_out, double* R_out) {
  double E = 0, R = 0;
  memset(planeColorFF, 0, (size_t)np * sizeof(float)); /* :401-402 */
  memset(planeColorFC, 0, (size_t)np * sizeof(float));
  for (int img = 0; img < n; img++) { /* :406 */
    const float* plane2imgX = p2x + (size_t)img * np;
    const float* plane2imgY = p2y + (size_t)img * np;
    const float* image = images + (size_t)img * wI * hI;
    for (int pi = 0; pi < np; pi++) { /* :412 */
      if (isnan(plane2imgX[pi])) continue;
      float color = orc_interp(image, plane2imgX[pi], plane2imgY[pi], wI); /* :417-418 */
      float fac = orc_interp(vignetteFactor, plane2imgX[pi], plane2imgY[pi], wI);
      if (isnan(fac)) continue;
      if (isnan(color)) continue;
      double residual = (double)((color - planeColor[pi] * fac) * (color - planeColor[pi] * fac)); /* :423, float product */
      if (fabs(residual) > oth2) { /* :424-429 (abs resolves to the double overload in the reference build) */
        E += oth2;
        R++;
        continue;
      }
      planeColorFF[pi] += fac * fac; /* :432-433 */
      planeColorFC[pi] += color * fac;
      if (isnan(planeColor[pi])) continue; /* :435 */
      E += residual;
      R++;
    }
  }
  for (int pi = 0; pi < np; pi++) { /* :441-447 */
    if (planeColorFF[pi] < 1) planeColor[pi] = NAN;
    else planeColor[pi] = planeColorFC[pi] / planeColorFF[pi];
  }
  *E_out = E;
  *R_out = R;
}

/* "optimize vignette", :455-527: TT/CT are rebuilt by bilinear scatter, vignetteFactor is read and then
 * replaced and normalised to a maximum of 1. */
void orc_vcal_vignette_step(const float* images, const float* p2x, const float* p2y, int n, int wI, int hI, int np,
                            const float* planeColor, float* vignetteFactor, float* vignetteFactorTT,
                            float* vignetteFactorCT, int oth2, double* E_out, double* R_out) {
  double E = 0, R = 0;
  memset(vignetteFactorTT, 0, (size_t)hI * wI * sizeof(float)); /* :457-458 */
  memset(vignetteFactorCT, 0, (size_t)hI * wI * sizeof(float));
  for (int img = 0; img < n; img++) { /* :461 */
    const float* plane2imgX = p2x + (size_t)img * np;
    const float* plane2imgY = p2y + (size_t)img * np;
    const float* image = images + (size_t)img * wI * hI;
    for (int pi = 0; pi < np; pi++) { /* :467 */
      if (isnan(plane2imgX[pi])) continue;
      float x = plane2imgX[pi];
      float y = plane2imgY[pi];
      float colorImage = orc_interp(image, x, y, wI); /* :473-475 */
      float fac = orc_interp(vignetteFactor, x, y, wI);
      float colorPlane = planeColor[pi];
      if (isnan(colorPlane)) continue;
      if (isnan(colorImage)) continue;
      double residual = (double)((colorImage - colorPlane * fac) * (colorImage - colorPlane * fac)); /* :480 */
      if (fabs(residual) > oth2) { /* :481-486 */
        E += oth2;
        R++;
        continue;
      }
      int ix = (int)x; /* :489-493 */
      int iy = (int)y;
      float dx = x - ix;
      float dy = y - iy;
      float dxdy = dx * dy;
      vignetteFactorTT[ix + iy * wI + 0] += (1 - dx - dy + dxdy) * colorPlane * colorPlane; /* :495-498 */
      vignetteFactorTT[ix + iy * wI + 1] += (dx - dxdy) * colorPlane * colorPlane;
      vignetteFactorTT[ix + iy * wI + wI] += (dy - dxdy) * colorPlane * colorPlane;
      vignetteFactorTT[ix + iy * wI + 1 + wI] += dxdy * colorPlane * colorPlane;
      vignetteFactorCT[ix + iy * wI + 0] += (1 - dx - dy + dxdy) * colorImage * colorPlane; /* :500-503 */
      vignetteFactorCT[ix + iy * wI + 1] += (dx - dxdy) * colorImage * colorPlane;
      vignetteFactorCT[ix + iy * wI + wI] += (dy - dxdy) * colorImage * colorPlane;
      vignetteFactorCT[ix + iy * wI + 1 + wI] += dxdy * colorImage * colorPlane;
      if (isnan(fac)) continue; /* :505 */
      E += residual;
      R++;
    }
  }
  float maxFac = 0; /* :511-521 */
  for (int pi = 0; pi < hI * wI; pi++) {
    if (vignetteFactorTT[pi] < 1) vignetteFactor[pi] = NAN;
    else {
      vignetteFactor[pi] = vignetteFactorCT[pi] / vignetteFactorTT[pi];
      if (vignetteFactor[pi] > maxFac) maxFac = vignetteFactor[pi];
    }
  }
  for (int pi = 0; pi < hI * wI; pi++) vignetteFactor[pi] /= maxFac; /* :526-527 */
  *E_out = E;
  *R_out = R;
}

/* src/main_vignetteCalib.cpp:293-301: a pixel and a 5 x 5 neighbour that differ by more than maxAbsGrad (the reference's
 * int, :130) both become NaN -- in place and in raster order, so what an earlier pixel masked no longer takes part
 * (fabsf of a NaN difference is never > anything). */
void orc_vcal_gradient_mask(float* image, int wI, int hI, int maxAbsGrad) {
  for (int y = 2; y < hI - 2; y++)
    for (int x = 2; x < wI - 2; x++)
      for (int deltax = -2; deltax < 3; deltax++)
        for (int deltay = -2; deltay < 3; deltay++)
          if (fabsf(image[x + y * wI] - image[x + deltax + (y + deltay) * wI]) > maxAbsGrad) {
            image[x + y * wI] = NAN;
            image[x + deltax + (y + deltay) * wI] = NAN;
          }
}

/* src/main_vignetteCalib.cpp:345-357: a plane point whose image position, rounded by (int)(v + 0.5) (float + double 0.5,
 * truncated), is not strictly inside (1, w-2) x (1, h-2) loses both coordinates (NaN included: the conversion of NaN is
 * whatever cvttsd2si gives, INT_MIN, which fails the test). */
void orc_vcal_mask_coords(float* x, float* y, int n, int wI, int hI) {
  for (int i = 0; i < n; i++) {
    int u_d = (int)(x[i] + 0.5);
    int v_d = (int)(y[i] + 0.5);
    if (!(u_d > 1 && v_d > 1 && u_d < wI - 2 && v_d < hI - 2)) {
      x[i] = NAN;
      y[i] = NAN;
    }
  }
}

/* "dilate & smoothe vignette by 4 pixel for output", src/main_vignetteCalib.cpp:541-566: four passes of a NaN-aware
 * 3 x 3 mean (a pixel with no finite neighbour keeps its value); the nine conditional adds in the reference's order.
 * tt = result, ct = scratch (ends up holding the input of the last pass, as in the reference). */
void orc_vcal_smooth(const float* vignetteFactor, int wI, int hI, float* tt, float* ct) {
  memcpy(tt, vignetteFactor, sizeof(float) * hI * wI); /* :541 */
  for (int dilit = 0; dilit < 4; dilit++) {            /* :542 */
    memcpy(ct, tt, sizeof(float) * hI * wI);
    for (int y = 0; y < hI; y++)
      for (int x = 0; x < wI; x++) {
        int idx = x + y * wI;
        float sum = 0, num = 0;
        if (x < wI - 1 && y < hI - 1 && !isnan(ct[idx + 1 + wI])) { sum += ct[idx + 1 + wI]; num++; } /* :551-562 */
        if (x < wI - 1 && !isnan(ct[idx + 1])) { sum += ct[idx + 1]; num++; }
        if (x < wI - 1 && y > 0 && !isnan(ct[idx + 1 - wI])) { sum += ct[idx + 1 - wI]; num++; }
        if (y < hI - 1 && !isnan(ct[idx + wI])) { sum += ct[idx + wI]; num++; }
        if (!isnan(ct[idx])) { sum += ct[idx]; num++; }
        if (y > 0 && !isnan(ct[idx - wI])) { sum += ct[idx - wI]; num++; }
        if (y < hI - 1 && x > 0 && !isnan(ct[idx - 1 + wI])) { sum += ct[idx - 1 + wI]; num++; }
        if (x > 0 && !isnan(ct[idx - 1])) { sum += ct[idx - 1]; num++; }
        if (y > 0 && x > 0 && !isnan(ct[idx - 1 - wI])) { sum += ct[idx - 1 - wI]; num++; }
        if (num > 0) tt[idx] = sum / num; /* :563 */
      }
  }
}

/* DSO hand-off (NOT in the reference; own definition after DSO's FrameHessian::makeImages):
 * (I, dx, dy) triples and absSquaredGrad of one w x h level.  parity unpinned. */
void orc_gradients(const float* lvl, int w, int h, float* dI, float* abs2) {
  for (int i = 0; i < w * h; i++) {
    dI[3 * i + 0] = lvl[i];
    dI[3 * i + 1] = 0.f;
    dI[3 * i + 2] = 0.f;
    abs2[i] = 0.f;
  }
  for (int idx = w; idx < w * (h - 1); idx++) {
    float dx = 0.5f * (lvl[idx + 1] - lvl[idx - 1]);
    float dy = 0.5f * (lvl[idx + w] - lvl[idx - w]);
    if (!isfinite(dx)) dx = 0;
    if (!isfinite(dy)) dy = 0;
    dI[3 * idx + 1] = dx;
    dI[3 * idx + 2] = dy;
    abs2[idx] = dx * dx + dy * dy;
  }
}
