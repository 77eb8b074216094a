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
MDC_API mdch_fov* mdch_fov_create(const char* camera_txt);
MDC_API void mdch_fov_destroy(mdch_fov*);
MDC_API int mdch_fov_valid(const mdch_fov*);                 /* isValid() */
MDC_API int mdch_fov_has_gpu(const mdch_fov*);               /* 1 if the remap was uploaded to a GPU context */
MDC_API void mdch_fov_dims(const mdch_fov*, int d4[4]);      /* getInputDims(), getOutputDims(): in_w in_h out_w out_h */
/* getK_rect()[9] getK_org()[9] getOriginalCalibration()[5] getOmega()[1] normalised output calibration[5] */
MDC_API void mdch_fov_intrinsics(const mdch_fov*, float out29[29]);
MDC_API int mdch_fov_remap(const mdch_fov*, float* remap_x, float* remap_y); /* copies the tables; 0 if none */
MDC_API void mdch_fov_distort(mdch_fov*, float* x, float* y, int n);         /* distortCoordinates() */
MDC_API void mdch_fov_undistort_f32(const mdch_fov*, const float* in, float* out, int n_in, int n_out); /* undistort<float> */
MDC_API void mdch_fov_undistort_u8(const mdch_fov*, const unsigned char* in, float* out, int n_in, int n_out);

MDC_API void mdch_fov_model(const mdch_fov*, mdc_fov_model* model);           /* lens model for mdc_distort_points_* */

/* PhotometricUndistorter(std::string file, std::string vignetteImage, int w, int h) */
MDC_API mdch_photo* mdch_photo_create(const char* pcalib_txt, const char* vignette_image, int w, int h);
MDC_API void mdch_photo_destroy(mdch_photo*);
MDC_API int mdch_photo_valid(const mdch_photo*);             /* bit0 validGamma, bit1 validVignette */
MDC_API int mdch_photo_has_gpu(const mdch_photo*);
MDC_API int mdch_photo_ginv(mdch_photo*, float out256[256]); /* getGInv(); 0 if invalid */
MDC_API int mdch_photo_g(mdch_photo*, float out256[256]);    /* getG(); 0 if invalid */
MDC_API int mdch_photo_vignette(const mdch_photo*, float* map, float* inv); /* copies w*h each (either may be NULL); 0 if invalid */
MDC_API void mdch_photo_unmap(mdch_photo*, unsigned char* in, float* out, int n, int g, int v, int o); /* unMapImage() */

/* Uploads the tables of the two objects (either may be NULL) into one GPU
 * context so the fused mdc_process_* entry points can be used. */
MDC_API int mdch_bind(mdc_ctx* ctx, const mdch_fov* fov, const mdch_photo* photo);

/* Serialises the tables of the two objects (either may be NULL) on the host into the
 * blob format of mdc_export_tables / mdc_import_tables -- what rank 0 broadcasts to the
 * other ranks of a multi-GPU job.  Needs no GPU.  mdch_pack_tables(f, p, NULL, 0, &n)
 * returns the size.  Returns MDC_OK or MDC_ERR_ARG (buffer too small). */
MDC_API int mdch_pack_tables(const mdch_fov* fov, const mdch_photo* photo, void* blob, size_t cap, size_t* size);

/* ---- class DatasetReader (include/mono_dataset_code/BenchmarkDatasetReader.h; reference
 * src/BenchmarkDatasetReader.h:83-345) ------------------------------------------------------------ */
typedef struct mdch_reader mdch_reader;
MDC_API mdch_reader* mdch_reader_create(const char* folder);   /* DatasetReader(std::string folder) */
MDC_API void mdch_reader_destroy(mdch_reader*);
MDC_API int mdch_reader_num_images(mdch_reader*);                /* getNumImages() */
MDC_API double mdch_reader_timestamp(mdch_reader*, int id);      /* getTimestamp() */
MDC_API float mdch_reader_exposure(mdch_reader*, int id);        /* getExposure() */
MDC_API void mdch_reader_dims(mdch_reader*, int d4[4]);          /* in_w in_h out_w out_h of its UndistorterFOV */
/* getImage(id, rectify, removeGamma, removeVignette, nanOverexposed): copies ExposureImage::image into out
 * (cap floats) and the other public fields into meta = {w, h, id}, stamp, exposure; 1 on success, 0 if getImage
 * returned 0 or cap is too small. */
MDC_API int mdch_reader_get_image(mdch_reader*, int id, int rectify, int g, int v, int o, float* out, long cap, int meta3[3],
                          double* stamp, float* exposure);
/* getImages(first, count, ...): image i into out + i * frame_floats; ok[i] = 1 where an image was produced.
 * Returns the number produced. */
MDC_API int mdch_reader_get_images(mdch_reader*, int first, int count, int rectify, int g, int v, int o, float* out,
                           long frame_floats, unsigned char* ok);
/* getImagesDevice(first, count, ..., out, valid): results stay in the caller's DEVICE arrays (include/mdc_hip.h: mdc_device_outputs;
 * frame first + i at position i), on the device of mdch_reader_device(); mdch_reader_context() = getContext().  Returns the number produced. */
MDC_API int mdch_reader_get_images_device(mdch_reader*, int first, int count, int rectify, int g, int v, int o, const mdc_device_outputs* out,
                                  unsigned char* valid);
MDC_API mdc_ctx* mdch_reader_context(mdch_reader*);
MDC_API int mdch_reader_device(mdch_reader*);
MDC_API int mdch_reader_get_raw(mdch_reader*, int id, unsigned char* out, long cap, int wh[2]); /* getImageRaw(); 1 / 0 */
MDC_API void mdch_reader_set_threads(mdch_reader*, int n);       /* setDecodeThreads() */
MDC_API void mdch_reader_set_prefetch(mdch_reader*, int frames); /* setPrefetch() */
MDC_API void mdch_reader_set_lookahead(mdch_reader*, int frames); /* setResultLookahead(): getImage results made ahead on JPEG sequences read in order */
MDC_API void mdch_reader_set_gpu_jpeg(mdch_reader*, int stage);   /* setGpuJpegStage(): 0 host, 1 device inverse DCT, 2 (or any other) device Huffman too */
MDC_API const char* mdch_reader_last_error(mdch_reader*);
MDC_API void mdch_reader_prefetch_stats(mdch_reader*, long hits_misses[2]); /* getPrefetchStats() */
/* getDeviceStats(lane, ...): device ordinal + frames produced, seconds waiting for the decoders + seconds in GPU calls; 0 = no such lane */
MDC_API int mdch_reader_device_stats(mdch_reader*, int lane, int64_t device_frames[2], double wait_gpu_seconds[2]);

/* The reader's frame decoders on a byte string (8-bit gray PNG, PGM P5, baseline JPEG): 1 on success, else 0
 * with the reason in err (errcap bytes).  wh = decoded size (also set when only cap was too small). */
MDC_API int mdch_decode_gray8(const unsigned char* data, size_t n, unsigned char* out, size_t cap, int wh[2], char* err,
                      size_t errcap);

/* The host half of JPEG decoding for the GPU stage (mdc_process_jpeg_frames_host / mdc_jpeg_idct_batch_device, include/mdc_hip.h):
 * Huffman-decodes a baseline or progressive JPEG into a coefficient record -- 64 x uint16 luma quantisation table (natural
 * order), then rows of `pitch_blocks` blocks of 64 int16 quantised luma coefficients (natural order) -- WITHOUT the inverse
 * DCT.  dims = {w, h, pitch_blocks, block rows}.  mdch_jpeg_record_bytes(w, h, dims2) gives the record size and the
 * {pitch, rows} that fit every sampling layout of a w x h file.  1 on success, else 0 with the reason in err. */
MDC_API size_t mdch_jpeg_record_bytes(int w, int h, int pitch_rows[2]);
MDC_API int mdch_decode_jpeg_record(const unsigned char* data, size_t n, void* record, size_t record_bytes, int pitch_blocks, int dims[4], char* err,
                            size_t errcap);

/* The host part of JPEG decoding when the GPU does the Huffman decoding as well (mdc_process_jpeg_streams_host, include/mdc_hip.h):
 * parses the markers, builds the scan's two decode tables and copies the entropy-coded segment without its byte stuffing into
 * `stream` (4-byte aligned, cap bytes; page-locked memory for the upload): mdc_jpeg_stream_header + bytes + 16 zero bytes.
 * Returns the bytes written, 0 (reason in err) for what the device decoder does not take -- more than one component,
 * progressive files, restart markers, a stream that does not fit: decode those with mdch_decode_jpeg_record / _gray8. */
MDC_API long long mdch_jpeg_stream(const unsigned char* data, size_t n, void* stream, size_t cap, int wh[2], char* err, size_t errcap);

/* ExposureImage's pixel pool (include/mono_dataset_code/ExposureImage.h): page-locked blocks carved out of slabs of up to 64
 * images, lowest free address first (consecutive images lie back to back: a chunk of results leaves the GPU with one copy).  mdch_image_pool_trim releases every slab without a live image; mdch_image_pool_idle_bytes = the bytes it would release. */
MDC_API float* mdch_image_alloc(unsigned long nfloats);
MDC_API void mdch_image_free(float* block);
MDC_API void mdch_image_pool_trim(void);
MDC_API unsigned long mdch_image_pool_idle_bytes(void);

#ifdef __cplusplus
}
#endif
#endif
