/*
 * mdc_host.h -- C facade over the drop-in C++ classes (libmdc_host.so) so that
 * non-C++ callers (the Python tests and bench.py via ctypes; a cgo/JNI binding
 * would look the same) can build calibration tables with the very code the C++
 * callers use, and bind them into a GPU context of include/mdc_hip.h.
 *
 * Each handle wraps one object of the class it is named after:
 *   mdch_fov_*   -> class UndistorterFOV          (reference src/FOVUndistorter.h:36-96)
 *   mdch_photo_* -> class PhotometricUndistorter  (reference src/PhotometricUndistorter.h:37-54)
 */
#ifndef MDC_HOST_H
#define MDC_HOST_H
#include "mdc_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mdch_fov mdch_fov;
typedef struct mdch_photo mdch_photo;

/* UndistorterFOV(const char* configFileName) */
mdch_fov* mdch_fov_create(const char* camera_txt);
void mdch_fov_destroy(mdch_fov*);
int mdch_fov_valid(const mdch_fov*);                 /* isValid() */
int mdch_fov_has_gpu(const mdch_fov*);               /* 1 if the remap was uploaded to a GPU context */
void mdch_fov_dims(const mdch_fov*, int d4[4]);      /* getInputDims(), getOutputDims(): in_w in_h out_w out_h */
/* getK_rect()[9] getK_org()[9] getOriginalCalibration()[5] getOmega()[1] normalised output calibration[5] */
void mdch_fov_intrinsics(const mdch_fov*, float out29[29]);
int mdch_fov_remap(const mdch_fov*, float* remap_x, float* remap_y); /* copies the tables; 0 if none */
void mdch_fov_distort(mdch_fov*, float* x, float* y, int n);         /* distortCoordinates() */
void mdch_fov_undistort_f32(const mdch_fov*, const float* in, float* out, int n_in, int n_out); /* undistort<float> */
void mdch_fov_undistort_u8(const mdch_fov*, const unsigned char* in, float* out, int n_in, int n_out);

void mdch_fov_model(const mdch_fov*, mdc_fov_model* model);           /* lens model for mdc_distort_points_* */

/* PhotometricUndistorter(std::string file, std::string vignetteImage, int w, int h) */
mdch_photo* mdch_photo_create(const char* pcalib_txt, const char* vignette_image, int w, int h);
void mdch_photo_destroy(mdch_photo*);
int mdch_photo_valid(const mdch_photo*);             /* bit0 validGamma, bit1 validVignette */
int mdch_photo_has_gpu(const mdch_photo*);
int mdch_photo_ginv(mdch_photo*, float out256[256]); /* getGInv(); 0 if invalid */
int mdch_photo_g(mdch_photo*, float out256[256]);    /* getG(); 0 if invalid */
int mdch_photo_vignette(const mdch_photo*, float* map, float* inv); /* copies w*h each (either may be NULL); 0 if invalid */
void mdch_photo_unmap(mdch_photo*, unsigned char* in, float* out, int n, int g, int v, int o); /* unMapImage() */

/* Uploads the tables of the two objects (either may be NULL) into one GPU
 * context so the fused mdc_process_* entry points can be used. */
int mdch_bind(mdc_ctx* ctx, const mdch_fov* fov, const mdch_photo* photo);

/* Serialises the tables of the two objects (either may be NULL) on the host into the
 * blob format of mdc_export_tables / mdc_import_tables -- what rank 0 broadcasts to the
 * other ranks of a multi-GPU job.  Needs no GPU.  mdch_pack_tables(f, p, NULL, 0, &n)
 * returns the size.  Returns MDC_OK or MDC_ERR_ARG (buffer too small). */
int mdch_pack_tables(const mdch_fov* fov, const mdch_photo* photo, void* blob, size_t cap, size_t* size);

#ifdef __cplusplus
}
#endif
#endif
