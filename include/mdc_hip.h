/*
 * mdc_hip.h -- C ABI of libmdc_hip.so: the MI355X (gfx950) implementation of the
 * per-frame photometric + FOV undistortion hot path of tum-vision/mono_dataset_code.
 *
 * This is the drop-in boundary.  Everything above it (the C++ classes
 * PhotometricUndistorter / UndistorterFOV in include/mono_dataset_code/, any
 * ctypes / cgo / JNI binding) sees only plain pointers, sizes and status codes.
 * Each entry point names the reference interface it replaces; paths are
 * relative to the reference repository root.
 *
 * Conventions
 *   - every function returns MDC_OK (0) or a negative mdc_status; the message
 *     for the last failure on a context is mdc_last_error(ctx);
 *   - *_host functions take host pointers, are synchronous and blocking, and
 *     borrow the caller's buffers for the duration of the call only -- the
 *     semantics of the reference methods they stand in for;
 *   - *_device functions take device pointers valid on the context's GPU and
 *     enqueue on `stream` (a hipStream_t passed as void*; NULL = HIP's default
 *     stream, as everywhere in HIP) without synchronising;
 *   - a context is bound to one GPU and may be used from several host threads at
 *     once (the reference's unMapImage / undistort are re-entrant on shared objects,
 *     src/FOVUndistorter.cpp:322 is const): the per-frame entry points only read the
 *     context's tables and run concurrently -- every host-pointer call on its own
 *     internal stream and staging buffers (up to 8 in flight, further callers wait),
 *     every *_device call on the caller's stream; the setters (tables, options,
 *     mdc_tune_device) are exclusive and wait for the device.  mdc_last_error(ctx)
 *     returns the calling thread's own last failure on that context.  Different
 *     contexts are independent; no global state.
 */
#ifndef MDC_HIP_H
#define MDC_HIP_H

#include <stddef.h>
#include <stdint.h>

/* The libraries are built with -fvisibility=hidden: the entry points marked MDC_API are the whole dynamic symbol table
 * (tests/test_abi.py compares `nm -D` with this header, exactly). */
#ifndef MDC_API
#if defined(__GNUC__) || defined(__clang__)
#define MDC_API __attribute__((visibility("default")))
#else
#define MDC_API
#endif
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mdc_ctx mdc_ctx;

typedef enum mdc_status {
  MDC_OK = 0,
  MDC_ERR_ARG = -1,       /* NULL / out-of-range argument */
  MDC_ERR_STATE = -2,     /* a table the call needs was never set (cf. the reference's valid/validGamma/validVignette) */
  MDC_ERR_SIZE = -3,      /* pixel count does not match the tables (cf. FOVUndistorter.cpp:327-338) */
  MDC_ERR_HIP = -4,       /* a HIP runtime call failed */
  MDC_ERR_NO_DEVICE = -5, /* no gfx950 device visible: there is NO CPU fallback */
  MDC_ERR_NOMEM = -6      /* a host-side allocation failed (no exception crosses this interface) */
} mdc_status;

/* Flag word of the per-frame calls: the four bools of
 * DatasetReader::getImage(id, rectify, removeGamma, removeVignette, nanOverexposed)
 * (src/BenchmarkDatasetReader.h:188) resp. the three of
 * PhotometricUndistorter::unMapImage(..., undoGamma, undoVignette, killOverexposed)
 * (src/PhotometricUndistorter.h:43).  Normalised inside the library exactly as
 * src/PhotometricUndistorter.cpp:173-189 does. */
enum {
  MDC_GAMMA = 1u,    /* undoGamma / removeGamma          */
  MDC_VIGNETTE = 2u, /* undoVignette / removeVignette    */
  MDC_KILL_OVEREXPOSED = 4u, /* killOverexposed / nanOverexposed: raw 255 -> NaN */
  MDC_RECTIFY = 8u   /* rectify (mdc_process_* only)     */
};

/* Pipeline selector for mdc_set_option(MDC_OPT_KERNEL) -- test/bench hook. */
enum {
  MDC_KERNEL_AUTO = 0,   /* LDS-tiled kernel when the remap allows it, else gather */
  MDC_KERNEL_GATHER = 1, /* direct global gather (always legal)                    */
  MDC_KERNEL_TILED = 2   /* LDS-staged source windows (fails if not plannable)     */
};
enum { MDC_OPT_KERNEL = 1, MDC_OPT_FRAMES_PER_BLOCK = 2 /* frames a workgroup loops over; 0 = automatic */,
       MDC_OPT_TILE_ROWS = 5 /* tuning: output tile rows {16, 32, 60, 64} (64 columns: 256 / 512 / 960 / 1024 threads); 0 = automatic */,
       MDC_OPT_TILE_ORDER = 6 /* tuning: placement of output tiles on the 8 XCDs, MDC_ORDER_* */,
       MDC_OPT_WINDOW_BUFFERS = 7 /* tuning: LDS window buffers per workgroup, 2..4 (frames staged ahead + 1); 0 = automatic */,
       MDC_OPT_FRAME_INTERLEAVE = 8 /* tuning: a workgroup takes every G-th frame (1) or a run of consecutive frames (0) */,
       MDC_OPT_TILE_COLS = 9 /* tuning: output tile {64, 128} x rows (128 x {16, 32}: 512 / 1024 threads); 0 = automatic */,
       MDC_OPT_PIN_CALLER_BUFFERS = 10 /* 1: page-lock, in place, the W*H float buffer a caller passes repeatedly as
          image_out of mdc_unmap_host / input of mdc_undistort_host_f32 (hipHostRegister on the second consecutive
          sighting of the same pointer + size; released by mdc_destroy or when the option is cleared), so that the
          reference's two-call composition (unMapImage -> internalTempBuffer -> undistort<float>,
          src/BenchmarkDatasetReader.h:222-223) copies at PCIe rate.  OFF by default -- the caller promises that such a
          buffer stays allocated for as long as the context lives.  Also switched on by the environment variable
          MDC_PIN_CALLER_BUFFERS=1, for callers that cannot be recompiled. */,
       MDC_OPT_TWO_STAGE = 11 /* tuning: the wave-private strip kernel -- a wave owns a 128 x 8 output tile, stages its own source
          window, converts every staged source pixel to lut * vignette ONCE and samples floats, 16 outputs per lane, pyramid
          levels out of registers -- for remaps with about one output or more per source pixel (the scale-1 rectification of
          BASELINE.json configs[4], magnifying remaps): 0 = automatic (when fewer source pixels are staged than there are
          outputs), 1 = whenever it can be planned, 2 = never */,
       MDC_OPT_PREFETCH_CHUNK = 12 /* tuning: the strip path walks large batches in chunks and reads the next chunk's source
          rows linearly into the Infinity Cache before the launch that samples them: frames per chunk, 0 = automatic
          (~26 MiB of source rows, in whole frame groups), -1 = no prefetch (one launch over the whole batch) */,
       MDC_OPT_PREFETCH_STREAMS = 13 /* tuning: those chunks alternate between the caller's stream and a second, internal one
          (joined back into the caller's stream before the call returns: the caller sees one stream), so that a chunk's tail
          and the next chunk's prefetch run under the other chunk's launch: 0 = automatic (2), 1, 2 */,
       MDC_OPT_ZERO_COPY = 14 /* the host-pointer calls (mdc_process_host, mdc_unmap_host, mdc_undistort_host_*,
          mdc_process_frames_host, mdc_process_jpeg_frames_host) hand a caller's buffer to the kernels AS IT IS when it lies in
          page-locked memory mapped into the device's address space (mdc_host_alloc, hipHostMalloc, hipHostRegister; asked
          of the runtime per call): the kernel reads the frame / writes the result over PCIe itself, both directions at once,
          instead of copy in -> kernel -> copy out.  Pageable buffers go through the staging copies as before.
          0 = automatic (on), 1 = on, 2 = off */,
       MDC_OPT_DEVICE_PIPELINE_CHUNK = 16 /* tuning: frames per chunk of the device-output pipeline (mdc_process_*_host_to_device:
          upload || decode || fused pass), 16..256; 0 = automatic (64, or MDC_PIPE_DEV_CHUNK in the environment) */,
       MDC_OPT_DEVICE_PIPELINE_CHUNK_HINT = 17 /* the same as a HINT: what "automatic" means on this context where neither the option above
          nor MDC_PIPE_DEV_CHUNK is set (the reader gives 128 to the contexts it runs two lanes on); 0 = no hint */,
       MDC_OPT_TAIL_TAPER = 15 /* tuning: a large launch of the tiled kernel ends on frame groups of 1/2, 1/4 and 1/8 of the
          frames per workgroup, so the slots that free up when its last long workgroups finish do not idle for a long
          workgroup's time: 0 = automatic (on), 1 = on, 2 = off */ };
enum { MDC_ORDER_BANDS = 0 /* row-major runs of tiles per XCD */, MDC_ORDER_ROWS = 1 /* whole tile rows per XCD */,
       MDC_ORDER_IDENTITY = 2 /* block b = tile b (diagnosis) */,
       MDC_ORDER_BLOCKS2D = 3 /* the tile grid cut into 8 rectangles, one per XCD */ };

typedef struct mdc_info {
  int device;                /* HIP device ordinal                                  */
  int in_w, in_h;            /* raw frame size (0 if unknown)                       */
  int out_w, out_h;          /* rectified size (0 if no remap set)                  */
  int valid_gamma;           /* GInv set     (PhotometricUndistorter::validGamma)   */
  int valid_vignette;        /* vignette set (PhotometricUndistorter::validVignette)*/
  int valid_remap;           /* remap set    (UndistorterFOV::valid)                */
  int tiled;                 /* 1 if the LDS-tiled kernel is planned for the remap  */
  int tile_w, tile_h;        /* output tile of the tiled kernel                     */
  int n_tiles;
  int lds_bytes;             /* dynamic LDS per workgroup of the tiled kernel       */
  int window_buffers;        /* LDS window buffers of the tiled kernel (frames staged ahead + 1) */
  int f32_tiled;             /* 1 if undistort<float> runs on the LDS-tiled kernel; it has its own tile shape: */
  int f32_tile_w, f32_tile_h;
  int src_bbox[4];           /* x0,y0,x1,y1 (inclusive) of source pixels any valid output taps */
  int64_t src_bbox_bytes;    /* bbox area in bytes (u8 source)                      */
  int64_t src_staged_bytes;  /* bytes the tiled kernel stages per frame (sum of the exact per-row windows) */
  int64_t n_black;           /* outputs whose remap is the (-1,-1) sentinel         */
  int two_stage;             /* 1 if the fused pass runs on the wave-private strip kernel (MDC_OPT_TWO_STAGE) */
  int prefetch_chunk;        /* strip path: frames per chunk of a fused-pyramid batch of >= 2 chunks (each chunk: one linear
                                prefetch launch + one remap launch, MDC_OPT_PREFETCH_CHUNK); 0 = batches go in one launch */
  int prefetch_streams;      /* streams those chunks alternate over (MDC_OPT_PREFETCH_STREAMS); 0 if prefetch_chunk is 0 */
} mdc_info;

/* ---- lifetime -------------------------------------------------------------- */

/* Creates a context on HIP device `device` (-1 = the calling thread's current
 * device).  MDC_ERR_NO_DEVICE if no GPU is visible. */
MDC_API int mdc_create(int device, mdc_ctx** out);
MDC_API int mdc_device_count(void); /* visible HIP devices (0 without a GPU or a usable runtime) */
/* PCI address of the context's GPU, "0000:8b:00.0" (sysfs: /sys/bus/pci/devices/<that>/{local_cpulist, numa_node, pp_dpm_sclk}):
 * what a multi-GPU host side pins its per-device threads by. */
MDC_API int mdc_device_pci_bus_id(mdc_ctx* ctx, char* buf, size_t cap);
MDC_API void mdc_destroy(mdc_ctx* ctx);
MDC_API const char* mdc_last_error(const mdc_ctx* ctx); /* never NULL; "" if no error; ctx may be NULL (creation errors) */
MDC_API int mdc_get_info(mdc_ctx* ctx, mdc_info* info);
MDC_API int mdc_set_option(mdc_ctx* ctx, int option, int value);
/* Build-time switches of this library that are NOT at their shipped value, "NAME=value ..." (csrc/mdc_build_config.h):
 * "" for the product build.  Anything else is an experiment / debug / diagnosis build (mono_dataset_code_amd/variants/);
 * a diagnosis build computes wrong results on purpose and says MDC_DIAGNOSIS_BUILD here.  Never NULL, static storage. */
MDC_API const char* mdc_build_flags(void);
/* Identity of the kernel build: 16 hex digits of a SHA-256 over the library's sources, shared headers and compile flags,
 * generated by the build recipe (mono_dataset_code_amd/build.py) and linked in.  Measurements recorded per kernel NAME
 * (profiles/hbm_traffic.json) carry it, so that a changed kernel under an unchanged name cannot inherit stale figures. */
MDC_API const char* mdc_code_id(void);

/* ---- calibration tables (once per sequence) -------------------------------- */

/* Uploads what PhotometricUndistorter's constructor builds
 * (src/PhotometricUndistorter.cpp:42-157): ginv = GInv[256] (NULL = validGamma
 * false), vignette_inv = vignetteMapInv[w*h] (NULL = validVignette false). */
MDC_API int mdc_set_photometric(mdc_ctx* ctx, const float* ginv, const float* vignette_inv, int w, int h);

/* Uploads what UndistorterFOV's constructor builds (src/FOVUndistorter.cpp:223-251):
 * remap_x/remap_y[out_w*out_h] in source pixels, (-1,-1) = black.  Plans the
 * tiled kernel (source window per output tile).  NULL tables clear the remap. */
MDC_API int mdc_set_remap(mdc_ctx* ctx, const float* remap_x, const float* remap_y, int in_w, int in_h, int out_w, int out_h);

/* ---- host-pointer, single-frame: under the reference's class methods -------- */

/* PhotometricUndistorter::unMapImage(image_in, image_out, n, g, v, o)
 * (src/PhotometricUndistorter.cpp:165-212). */
MDC_API int mdc_unmap_host(mdc_ctx* ctx, const uint8_t* image_in, float* image_out, int n, unsigned flags);

/* UndistorterFOV::undistort<float> / <unsigned char>(input, output, nPixIn, nPixOut)
 * (src/FOVUndistorter.cpp:322-370).  MDC_ERR_STATE without a remap (the reference
 * returns silently, :325), MDC_ERR_SIZE on a pixel-count mismatch (:327-338);
 * `output` is untouched in both cases. */
MDC_API int mdc_undistort_host_f32(mdc_ctx* ctx, const float* input, float* output, int n_in, int n_out);
MDC_API int mdc_undistort_host_u8(mdc_ctx* ctx, const uint8_t* input, float* output, int n_in, int n_out);

/* The whole of DatasetReader::getImage after decode (src/BenchmarkDatasetReader.h:207-241)
 * in one fused pass -- no W*H float intermediate (internalTempBuffer, :145,:222).
 * `out` holds out_w*out_h floats with MDC_RECTIFY, else in_w*in_h. */
MDC_API int mdc_process_host(mdc_ctx* ctx, const uint8_t* raw, float* out, unsigned flags);

/* ---- host-pointer, many frames: a sequence through PCIe ---------------------- */

/* Page-locked host memory for frames and results (hipHostMalloc / hipHostFree): copies from and
 * to it run asynchronously at PCIe rate, so a reader that keeps its decoded frames and its
 * ExposureImage::image buffers (src/ExposureImage.h:45) in such memory overlaps transfers with
 * the kernels.  NULL on failure. */
MDC_API void* mdc_host_alloc(size_t bytes);
MDC_API void mdc_host_free(void* p);

/* DatasetReader::getImage (src/BenchmarkDatasetReader.h:207-241, after decode) for nframes frames
 * in one call: raw[i] -> out[i], results identical to nframes mdc_process_host calls.  Frames go
 * through the GPU in chunks on two streams, so the upload of one chunk, the kernel of the next and
 * the download of the previous one overlap.  Any host memory works; pageable buffers make the HIP
 * runtime stage every copy (a few GB/s), mdc_host_alloc'ed ones reach the PCIe rate.  Blocking. */
MDC_API int mdc_process_frames_host(mdc_ctx* ctx, const uint8_t* const* raw, float* const* out, int64_t nframes,
                            unsigned flags);

/* JPEG ingest with the inverse DCT on the GPU.  The host does the serial half of JPEG decoding
 * -- Huffman decoding, mdch_decode_jpeg_coefs in include/mdc_host.h -- into a coefficient RECORD per frame:
 *   [64 x uint16 luma quantisation table, natural order][blocks_rows x blocks_w blocks of 64 int16 quantised coefficients,
 *   natural order]        (block grid padded to whole MCUs; the frame covers the first ceil(h/8) rows, ceil(w/8) blocks a row)
 * mdc_process_jpeg_frames_host is mdc_process_frames_host with record i in place of raw frame i: records go up over PCIe
 * (page-locked memory: mdc_host_alloc), the device dequantises + runs libjpeg's islow integer inverse DCT into the frame
 * buffer the fused kernel reads, results come back as before.  Identical to decoding on the host and calling
 * mdc_process_frames_host, bit for bit (integer arithmetic).  mdc_jpeg_idct_batch_device is the device stage alone:
 * nframes records, record_bytes apart (multiple of 16), -> nframes * w * h bytes. */
MDC_API int mdc_process_jpeg_frames_host(mdc_ctx* ctx, const void* const* records, int64_t record_bytes, int blocks_w, int blocks_rows,
                                 float* const* out, int64_t nframes, unsigned flags);
MDC_API int mdc_jpeg_idct_batch_device(mdc_ctx* ctx, const void* d_records, int64_t record_bytes, uint8_t* d_frames, int w, int h, int blocks_w,
                               int blocks_rows, int64_t nframes, void* stream);

/* JPEG ingest with the Huffman decoding on the GPU too.  The host only parses the file's markers, builds the decode
 * tables of the scan and copies the entropy-coded segment with its FF00 byte stuffing and its restart markers removed
 * (mdch_jpeg_stream in include/mdc_host.h: ~0.1 ms per 1280 x 1024 frame against ~2.5 ms for the Huffman decoding) into a STREAM
 * per frame (layout: mdc_jpeg_stream_header below).
 * Baseline / extended-sequential Huffman files of one scan: grayscale (what the TUM mono dataset ships) or YCbCr with the three
 * components interleaved (any luma sampling up to 4 x 4, chroma 1 x 1 -- 4:4:4, 4:2:2, 4:2:0, ...; only the luma plane is decoded
 * to samples, which is what cv::imread(..., GRAYSCALE) returns for such a file), with or without restart markers.  Progressive,
 * multi-scan, arithmetic-coded and subsampled-luma files: mdch_jpeg_stream says no and the caller takes the record path above.
 * mdc_process_jpeg_streams_host: stream i in place of raw frame i.  The device decodes every stream with 1024 threads
 * (subsequences of the bit stream whose entry states are relaxed until they are the sequential decoder's; restart intervals
 * have exact entry states and are decoded one per thread, csrc/mdc_jpeg.hip),
 * then runs the inverse DCT and the fused kernel as for records: the results equal the host decoder's path bit for bit.
 * Results leave the device with one copy per run of out[] buffers that lie back to back in memory (the ExposureImage pool hands
 * out such runs): contiguous page-locked outputs go at PCIe rate, scattered ones at about half of it.
 * status[i] (optional): 0 = done, 1 = the stream holds a code no table knows or too few blocks, 2 = header does not describe
 * a frame of this context -- out[i] is then not a result and the caller decodes that file on the host.
 * mdc_jpeg_huffman_batch_device is the device stage alone: nframes streams, stream_stride bytes apart (multiple of 16) ->
 * nframes records (layout above), d_status[i] as status[i]. */
#define MDC_JPEG_STREAM_MAGIC 0x33534a4du /* "MJS3" */
#define MDC_JPEG_HUFF_SUBTABLES 32
typedef struct mdc_jpeg_huff {
  /* t1, indexed by the next 11 bits (MSB first): bits 0-4 code length L (1..16; 0 = no such code; 31 = longer than 11 bits:
   * bits 16-31 then hold the index of a subtable), bits 5-8 run, bits 9-12 size (DC table: run 0, size = category), bit 13: the
   * size magnitude bits lie inside the window too and bits 16-31 hold the (sign-extended) coefficient value.
   * t2[sub][next 5 bits]: the codes of 12..16 bits below that 11-bit prefix, same entry format. */
  uint32_t t1[2048];
  uint32_t t2[MDC_JPEG_HUFF_SUBTABLES][32];
} mdc_jpeg_huff; /* 12288 bytes */
typedef struct mdc_jpeg_stream_header {
  uint32_t magic, w, h, ecs_bytes;
  uint32_t restart_interval; /* MCUs per restart interval (DRI), 0 = the scan has no restart markers */
  uint32_t n_intervals;      /* restart intervals of the scan (1 without restart markers) */
  uint32_t comp_info;        /* components of the scan (1, or 3: Y Cb Cr interleaved) | luma sampling h << 8 | v << 12 (chroma 1 x 1) */
  uint32_t ecs_offset;       /* byte offset of the entropy-coded bytes from the start of the stream, a multiple of 16 */
  uint16_t quant[64];        /* luma quantisation table, natural order */
  mdc_jpeg_huff dc, ac;      /* luma tables */
} mdc_jpeg_stream_header; /* 24736 bytes; then, three components only: mdc_jpeg_huff dc_chroma, ac_chroma; then, with restart markers: uint32_t
                             start_byte[n_intervals] (offset of each interval inside the entropy-coded bytes, markers removed); then padding to
                             ecs_offset, the entropy-coded bytes, at least 16 zero bytes */
MDC_API int mdc_process_jpeg_streams_host(mdc_ctx* ctx, const void* const* streams, const int64_t* stream_bytes, float* const* out, int64_t nframes,
                                  unsigned flags, int* status);
MDC_API int mdc_jpeg_huffman_batch_device(mdc_ctx* ctx, const void* d_streams, int64_t stream_stride, void* d_records, int64_t record_bytes, int w,
                                  int h, int blocks_w, int blocks_rows, int64_t nframes, int* d_status, void* stream);

/* ---- host frames in, results LEFT ON THE DEVICE ------------------------------------
 * The three pipelined calls above with a device-resident end: frames (raw / JPEG coefficient records / JPEG streams) come from
 * host memory, go through the same upload -> (Huffman ->) (inverse DCT ->) fused-pass pipeline, and the results -- the processed
 * frame and, on request, its box-pyramid levels and DSO-style gradient images (mdc_process_pyramid_gradients_batch_device) -- stay in
 * HBM for a GPU consumer; nothing crosses PCIe on the way out (the host-output calls top out at ~31 k frames/s of 640x480 floats).
 * Reference call site: DatasetReader::getImage, src/BenchmarkDatasetReader.h:188-243, whose `new ExposureImage` + host float
 * block this replaces for callers that keep working on the device.
 * `out` describes device arrays for a whole sequence; frame i of the call lands at position frame_index[i] of every array
 * (frame_index == NULL: position i).  Same bytes as the host-output call followed by a copy up.  Blocking: on return the
 * results are complete in device memory.  The calls run on streams of the context's own: work of the caller's streams that still
 * touches the arrays (a fill, a consumer of the previous results) must have finished before the call.  status as for mdc_process_jpeg_streams_host (the arrays' entries of a frame with
 * status != 0 are not results).  mdc_device_alloc / _free / mdc_copy_to_host: for callers without a HIP toolchain of their own. */
typedef struct mdc_device_outputs {
  float* base;                /* positions x (w x h) floats: the processed frames (rectified size with MDC_RECTIFY); required */
  int levels;                 /* 1 = base only; 2..4 = + box levels 1..levels-1 (level l: (w >> l) x (h >> l)) */
  float* level[3];            /* levels 1..3, positions x level size each; NULL beyond `levels` */
  float* dI[4];               /* optional, all NULL = none: (I, dx, dy) triples of level l, positions x 3 x level size */
  float* abs_squared_grad[4]; /* with dI: dx^2 + dy^2 of level l */
} mdc_device_outputs;
MDC_API int mdc_process_frames_host_to_device(mdc_ctx* ctx, const uint8_t* const* raw, int64_t nframes, unsigned flags,
                                              const mdc_device_outputs* out, const int64_t* frame_index);
MDC_API int mdc_process_jpeg_frames_host_to_device(mdc_ctx* ctx, const void* const* records, int64_t record_bytes, int blocks_w, int blocks_rows,
                                                   int64_t nframes, unsigned flags, const mdc_device_outputs* out, const int64_t* frame_index);
MDC_API int mdc_process_jpeg_streams_host_to_device(mdc_ctx* ctx, const void* const* streams, const int64_t* stream_bytes, int64_t nframes,
                                                    unsigned flags, const mdc_device_outputs* out, const int64_t* frame_index, int* status);
MDC_API int mdc_device_alloc(mdc_ctx* ctx, size_t bytes, void** d_ptr); /* device memory on the context's GPU: hipMalloc; a GiB or more:
   striped over the device's memory classes like mdc_alloc_striped_set_device's buffers (MDC_PLACEMENT=first: never).  Give it back with
   mdc_device_free on the same context, never with hipFree; what is left when the context is destroyed goes with it. */
/* BUFFER PLACEMENT by measurement.  On MI355X the time of one and the same launch depends on the ALLOCATIONS it runs on -- on the
 * physical pages behind the caller's frame and result buffers, and on the PAIR of them: 1.48 to 1.63 ms for the headline launch (4096
 * frames) between pairs of hipMalloc'ed buffers of one process on one device, stable for the life of the buffers, not changed by offsets
 * inside an allocation and not predicted by a linear write or read pass over them (profiles/r05_experiments/05_*, 06_*, 08_*, 09_*).
 * A caller that allocates its frame / result buffers once (a sequence, a ring) can allocate a few candidates and let the context time the
 * real pass on them, as mdc_tune_device does for tile shapes: mdc_tune_placement_device runs the fused pass of `flags` over nframes
 * frames on every pair (d_in[i], d_out[j]) -- 2 untimed + 5 timed launches each, median -- and returns the fastest pair; ms (optional,
 * n_in * n_out floats, ms[i * n_out + j]) = the medians.  n_in * n_out <= 256.  Every input candidate must hold the same frames.  The
 * caller frees the losers.  Blocking, ~10 launches' time per pair.  Afterwards every output candidate holds the results of the pass. */
MDC_API int mdc_tune_placement_device(mdc_ctx* ctx, const uint8_t* const* d_in, int n_in, float* const* d_out, int n_out, int64_t nframes,
                                      unsigned flags, void* stream, int* best_in, int* best_out, float* ms);
/* The same as an ALLOCATOR: a frame buffer and a result buffer for `nframes` frames of the pass `flags` (sizes from the context's tables;
 * in_bytes / out_bytes = 0: exactly that, larger values are honoured), made so that the pass runs fast on the pair -- what bench.py,
 * libmdc_multi's callers (tests/native/multi_gpu_seq.cpp) and the reader's device-resident sequences use instead of two hipMalloc's.
 * The context's tables for `flags` must be set (the pass itself is the probe: min(nframes, 4096) frames, 2 untimed + 5 timed launches per
 * measurement, median).  Blocking; `stream` is used for the probes and is idle on return.  Strategies:
 *   MDC_PLACE_FIRST   two hipMalloc's, as they come (no measurement; what AUTO does for pairs below 1 GiB, which live in the Infinity Cache);
 *   MDC_PLACE_MALLOC  up to 6 x 6 hipMalloc'ed candidates spread over the device's memory by spacer allocations, every pair timed
 *                     (mdc_tune_placement_device), the fastest kept, the rest freed; needs room for the candidates (else fewer, down to 1);
 *   MDC_PLACE_VMM     both buffers assembled from 512-MiB physical pieces (hipMemCreate / hipMemMap), which are first sorted into the
 *                     device's three memory classes by a timed linear stream against reference pieces; every buffer is then striped over
 *                     all classes in equal shares, piece by piece.  Also fits pairs that leave no room for candidates (a 50,000-frame
 *                     sequence).  An address a kernel may know stays mapped until mdc_free_placed_device; the address ranges themselves
 *                     are never recycled within a process (ROCm 7.2 serves stale translations for a re-used range: DESIGN.md 6.1).
 *   MDC_PLACE_AUTO    the library's default (MDC_PLACEMENT=first|malloc|vmm in the environment overrides it).
 * ms_first = the probe on the first pair of plain allocations (what a caller gets who takes them as they come; 0 where not measured),
 * ms_chosen = on the pair handed out.  Release with mdc_free_placed_device (waits for the device; never hipFree the pointers). */
enum { MDC_PLACE_AUTO = 0, MDC_PLACE_FIRST = 1, MDC_PLACE_MALLOC = 2, MDC_PLACE_VMM = 3 };
typedef struct mdc_placed_buffers {
  uint8_t* d_in;            /* nframes frames (in_bytes) */
  float* d_out;             /* nframes results (out_bytes) */
  size_t in_bytes, out_bytes;
  int64_t nframes, probe_frames;
  int strategy;             /* the one that ran (never AUTO) */
  int candidates_in, candidates_out, picked_in, picked_out; /* MALLOC: candidates made, pair kept */
  float pair_ms[64];        /* MALLOC: probe time of frames candidate i on results candidate j at [i * candidates_out + j] */
  int pieces, piece_mib;    /* VMM: physical pieces created, their size */
  int class_count[3];       /* VMM: pieces per memory class (class of piece 0 / of the first piece fast with it / fast with both) */
  float ms_first, ms_chosen;
  char note[384];           /* one line for logs: what was done */
  void* handle;             /* the allocator's own */
} mdc_placed_buffers;
MDC_API int mdc_alloc_placed_device(mdc_ctx* ctx, size_t in_bytes, size_t out_bytes, int64_t nframes, unsigned flags, int strategy, void* stream,
                                    mdc_placed_buffers* out);
MDC_API int mdc_free_placed_device(mdc_ctx* ctx, mdc_placed_buffers* buffers);
/* The further buffers of a step that writes more than one result per frame -- the pyramid levels of mdc_process_pyramid_batch_device, the
 * gradient images of mdc_process_pyramid_gradients_batch_device -- made the way MDC_PLACE_VMM makes the pair: n buffers of bytes[k], every
 * one striped over the device's memory classes in equal shares (no probe pass, no tables needed).  Where the device has no virtual memory
 * management (or MDC_PLACEMENT=first / malloc): n plain allocations, `strategy` says which.  Release with mdc_free_striped_set_device. */
#define MDC_STRIPED_SET_MAX 16
typedef struct mdc_striped_set {
  int n;
  void* d_ptr[MDC_STRIPED_SET_MAX];
  size_t bytes[MDC_STRIPED_SET_MAX];
  int strategy;            /* MDC_PLACE_VMM or MDC_PLACE_FIRST */
  int pieces, piece_mib;
  int class_count[3];
  char note[384];
  void* handle;
} mdc_striped_set;
MDC_API int mdc_alloc_striped_set_device(mdc_ctx* ctx, int n, const size_t* bytes, void* stream, mdc_striped_set* out);
MDC_API int mdc_free_striped_set_device(mdc_ctx* ctx, mdc_striped_set* set);
MDC_API void mdc_device_free(mdc_ctx* ctx, void* d_ptr);
MDC_API int mdc_copy_to_host(mdc_ctx* ctx, void* dst, const void* d_src, size_t bytes); /* blocking device -> host copy */

/* ---- device-pointer, batched: the throughput path --------------------------- */

/* unMapImage over nframes back-to-back frames (in: nframes*w*h u8; out: same count f32). */
MDC_API int mdc_unmap_batch_device(mdc_ctx* ctx, const uint8_t* d_in, float* d_out, int64_t nframes, unsigned flags,
                           void* stream);

/* getImage over nframes frames, fused photometric + remap when MDC_RECTIFY is set
 * (out: nframes*out_w*out_h f32), else identical to mdc_unmap_batch_device. */
MDC_API int mdc_process_batch_device(mdc_ctx* ctx, const uint8_t* d_in, float* d_out, int64_t nframes, unsigned flags,
                             void* stream);

/* undistort<float> over nframes float frames (in: nframes*in_w*in_h f32). */
MDC_API int mdc_undistort_batch_device_f32(mdc_ctx* ctx, const float* d_in, float* d_out, int64_t nframes, void* stream);

/* 2x2 box pyramid (BASELINE.json config 5; NOT in the reference -- definition in
 * DESIGN.md): for each of nframes w*h f32 images writes levels 1..levels-1 into
 * d_levels[l-1] (each nframes*(w>>l)*(h>>l) f32).  levels counts level 0. */
MDC_API int mdc_pyramid_batch_device(mdc_ctx* ctx, const float* d_base, int w, int h, int levels, float* const* d_levels,
                             int64_t nframes, void* stream);

/* getImage + box pyramid in ONE pass over the raw frames (config 5, the DSO-style preprocessing path):
 * d_base as mdc_process_batch_device writes it, plus levels 1..levels-1 as mdc_pyramid_batch_device
 * would derive them from d_base -- bit-identical results.  With MDC_RECTIFY and an output made of whole
 * tiles (out_w % 64 == 0, out_h % tile rows == 0) levels 1..3 come out of the remap kernel's registers
 * (no re-read of the base); other geometries and levels >= 4 fall back to one pass per level. */
MDC_API int mdc_process_pyramid_batch_device(mdc_ctx* ctx, const uint8_t* d_in, float* d_base, int levels,
                                     float* const* d_levels, int64_t nframes, unsigned flags, void* stream);

/* DSO hand-off of one pyramid level (NOT in the reference: the per-level loop of DSO's
 * FrameHessian::makeImages, definition in DESIGN.md section 5.5): for nframes images of w x h floats (level 0 = the
 * output of mdc_process_batch_device, levels 1.. = mdc_process_pyramid_batch_device's) writes
 *   d_dI                : nframes*w*h triples (I, dx, dy), dx = 0.5f*(I[i+1]-I[i-1]), dy = 0.5f*(I[i+w]-I[i-w]) over
 *                         the linear index range [w, w*(h-1)), non-finite differences -> 0, first / last row -> 0;
 *   d_abs_squared_grad  : nframes*w*h floats dx*dx + dy*dy. */
MDC_API int mdc_gradients_batch_device(mdc_ctx* ctx, const float* d_level, int w, int h, float* d_dI, float* d_abs_squared_grad,
                               int64_t nframes, void* stream);

/* The DSO-style preprocessing of a batch in ONE call: base as mdc_process_batch_device, levels
 * 1..levels-1 as mdc_process_pyramid_batch_device, and for EVERY level l (0 = the base) the gradient images of
 * mdc_gradients_batch_device in d_dI[l] / d_abs_squared_grad[l] (nframes * (w>>l) * (h>>l) triples / floats) -- bit-identical
 * to the separate calls.  The batch is walked in chunks of frames (chunk_frames = 0: automatic, 96 at 1280 x 1024): per
 * chunk one launch writes base + levels and ONE launch turns all levels of the chunk into gradient images (chunks only bound
 * the launch sizes: small ones lose to their tails).  (Gradients do not come out of the pyramid launch itself: a level-l
 * pixel's neighbours lie up to 2^l base pixels outside the tile that produced it, in other workgroups' registers; what a
 * fused launch could win is measured in DESIGN.md section 5.5.) */
MDC_API int mdc_process_pyramid_gradients_batch_device(mdc_ctx* ctx, const uint8_t* d_in, float* d_base, int levels, float* const* d_levels,
                                               float* const* d_dI, float* const* d_abs_squared_grad, int64_t nframes, unsigned flags,
                                               int chunk_frames, void* stream);

/* ---- lens model on many points ------------------------------------------------ */

/* The FOV camera of one UndistorterFOV object: camera.txt line 1 (fx fy cx cy omega, relative to the
 * input size), the input size, and the NORMALISED output calibration its constructor leaves behind
 * (src/FOVUndistorter.cpp:214-218) with the output size.  mdc_fov_model_of (MdcBind.h) / mdch_fov_model
 * (mdc_host.h) fill it from an object. */
typedef struct mdc_fov_model {
  float in_calib[5];
  int in_w, in_h;
  float out_calib[5];
  int out_w, out_h;
} mdc_fov_model;

/* UndistorterFOV::distortCoordinates(in_x, in_y, n) (src/FOVUndistorter.cpp:280-319): rectified pixel
 * coordinates -> raw pixel coordinates, in place, n points -- for callers that warp many points per
 * frame (vignetteCalib: 10^6 per image, src/main_vignetteCalib.cpp:284).  Bit-identical to the
 * reference built against glibc's libm: the kernel restates that library's fdlibm atanf, it does not
 * call the GPU math library's (which differs in the last bit).  The class method itself keeps running
 * on the host (it builds the remap tables); this entry point is the opt-in. */
MDC_API int mdc_distort_points_device(mdc_ctx* ctx, const mdc_fov_model* model, float* d_x, float* d_y, int64_t n, void* stream);
MDC_API int mdc_distort_points_host(mdc_ctx* ctx, const mdc_fov_model* model, float* x, float* y, int64_t n);

/* ---- vignetteCalib solver (src/main_vignetteCalib.cpp:395-527) ----------------------------------- */

/* One "optimize planeColor" half-iteration (:400-448) over n_images images of w x h floats (stacked in d_images,
 * NaN = masked pixel, :294-300) seen through the plane -> image coordinates d_p2x / d_p2y (n_images x n_plane, NaN =
 * plane point outside that image, after distortCoordinates, :284): for every plane point the sums FF, FC over the
 * images (d_ff, d_fc: n_plane floats, overwritten) and the new colour FC / FF (NaN where FF < 1) in d_plane_color,
 * which is read first for the residual test against oth2 (the reference's int, :397-398).  d_er receives
 * {E, R} of the reference's printf (:449).  FF, FC and the colours are bit-identical to the reference.
 * Samples whose 2x2 footprint would leave the image -- the reference relies on its caller's mask (:345-357,
 * mdc_vcal_mask_coords_device) and would read out of bounds -- are skipped, in this step, in the atomic vignette step
 * and in the contribution index alike: all three always see the same sample set. */
MDC_API int mdc_vcal_plane_step_device(mdc_ctx* ctx, const float* d_images, const float* d_p2x, const float* d_p2y, int n_images, int w, int h,
                               int n_plane, float* d_plane_color, const float* d_vignette_factor, int oth2, float* d_ff,
                               float* d_fc, double* d_er, void* stream);
/* One "optimize vignette" half-iteration (:455-527): bilinear scatter of the plane colours into the image grid
 * (d_tt, d_ct: w*h floats, overwritten), new factor CT / TT (NaN where TT < 1) normalised to a maximum of 1 in
 * d_vignette_factor (read first for the residual test).  A scatter-add by concurrent float atomics: equal to the
 * reference to ~1e-6 relative, not bitwise (the reference sums sequentially). */
MDC_API int mdc_vcal_vignette_step_device(mdc_ctx* ctx, const float* d_images, const float* d_p2x, const float* d_p2y, int n_images, int w,
                                  int h, int n_plane, const float* d_plane_color, float* d_vignette_factor, int oth2, float* d_tt,
                                  float* d_ct, double* d_er, void* stream);

/* The same half-iteration WITHOUT atomics and bit-identical to the reference.  plane2img coordinates and image colours
 * are fixed over the solver's iterations, so the scatter :489-503 is inverted once: mdc_vcal_index_create lists, for
 * every image pixel, the (image, plane point, corner) contributions it receives in the order of the reference's
 * sequential loop (image-major, plane point ascending).  mdc_vcal_vignette_step_indexed_device then lets one lane walk
 * one pixel's list front to back: same terms, same f32 expressions, same order -> d_tt, d_ct and d_vignette_factor
 * equal the reference bit for bit.  The index holds 4 x 16 bytes per valid sample (12.8 GB for 200 images x 10^6
 * plane points -- sized for HBM) and belongs to the (d_images, d_p2x, d_p2y) it was built from; building it
 * synchronises `stream`.  Samples whose 2x2 footprint leaves the image (the reference's caller excludes them, :283-300)
 * are dropped instead of written out of bounds.  d_er = {E, R} as above (E in tree order). */
typedef struct mdc_vcal_index mdc_vcal_index;
MDC_API int mdc_vcal_index_create(mdc_ctx* ctx, const float* d_images, const float* d_p2x, const float* d_p2y, int n_images, int w, int h,
                          int n_plane, void* stream, mdc_vcal_index** out);
MDC_API void mdc_vcal_index_destroy(mdc_vcal_index* index);
MDC_API int64_t mdc_vcal_index_bytes(const mdc_vcal_index* index);   /* device bytes of the contribution lists */
MDC_API int64_t mdc_vcal_index_entries(const mdc_vcal_index* index); /* list entries = 4 x valid samples */
MDC_API int mdc_vcal_vignette_step_indexed_device(mdc_ctx* ctx, const mdc_vcal_index* index, const float* d_plane_color,
                                          float* d_vignette_factor, int oth2, float* d_tt, float* d_ct, double* d_er, void* stream);

/* The whole iteration loop :395-527 in one call: builds the contribution index, then max_iterations times the plane step
 * and the indexed vignette step with the reference's outlier schedule (oth2 = 10000^2 in the first half of the iterations,
 * outlier_th^2 after, :397-398; the reference's defaults are 20 iterations, outlierTh 15), everything on `stream`,
 * one synchronisation at the end.  d_plane_color (n_plane floats; the reference starts from an uninitialised array --
 * pass what you want it to start from, e.g. zeros) and d_vignette_factor (w*h floats; the reference starts from 1,
 * :391) are updated in place and end up bit-identical to the reference's arrays after the same iterations from the
 * same start.  er_out (host, may be NULL): max_iterations x {E, R of the plane step, E, R of the vignette step} -- what
 * the reference prints as "R residual terms => sqrtf(E/R)" (:449, :523). */
MDC_API int mdc_vcal_solve_device(mdc_ctx* ctx, const float* d_images, const float* d_p2x, const float* d_p2y, int n_images, int w, int h,
                          int n_plane, float* d_plane_color, float* d_vignette_factor, int max_iterations, int outlier_th,
                          double* er_out, void* stream);

/* The calibration images as the solver wants them (:286-291): image k = mean_exposure * image k / exposure_time k
 * (an exposure of 0 counts as 1), in place on n_images stacked images of npix floats -- e.g. the output of
 * mdc_unmap_batch_device with MDC_GAMMA only, which is getImage(i, false, true, false, false) of :265.
 * d_exposure_times: n_images floats on the device. */
MDC_API int mdc_vcal_scale_images_device(mdc_ctx* ctx, float* d_images, int n_images, int64_t npix, float mean_exposure,
                                 const float* d_exposure_times, void* stream);

/* The gradient mask of the calibration images (:293-301; max_abs_grad = the reference's int maxAbsGrad, :130, default
 * 255): a pixel and a 5 x 5 neighbour that differ by more than max_abs_grad both become NaN, in place, with the
 * reference's raster-order semantics (a masked pixel no longer takes part) -- replayed exactly as a skewed wavefront,
 * one workgroup per image, n_images stacked w x h float images side by side.  Bit-identical to the reference. */
MDC_API int mdc_vcal_gradient_mask_device(mdc_ctx* ctx, float* d_images, int n_images, int w, int h, int max_abs_grad, void* stream);

/* The step between distortCoordinates and the solver (:345-357): plane points whose image position, rounded as
 * (int)(v + 0.5), is not strictly inside (1, w-2) x (1, h-2) get NaN coordinates -- the "outside this image" marker the
 * two half-iterations test for (:409, :468).  In place on n coordinate pairs (n_images x n_plane in one call is fine: the
 * rule does not depend on the image).  With mdc_distort_points_device before it, the plane -> image coordinates never
 * leave the device. */
MDC_API int mdc_vcal_mask_coords_device(mdc_ctx* ctx, float* d_x, float* d_y, int64_t n, int w, int h, void* stream);

/* "dilate & smoothe vignette by 4 pixel for output" (:541-566): four passes of a NaN-aware 3 x 3 mean over the w x h
 * factor map (what the reference writes as vignetteSmoothed.png, i.e. the vignette image PhotometricUndistorter reads).
 * d_smoothed (result) and d_scratch are w*h floats each, distinct from each other; d_vignette_factor is not modified and
 * must not be d_scratch.  Bit-identical to the reference. */
MDC_API int mdc_vcal_smooth_device(mdc_ctx* ctx, const float* d_vignette_factor, int w, int h, float* d_smoothed, float* d_scratch,
                           void* stream);

/* Plan selection by measurement.  Which tile shape and workgroup length is fastest depends on the remap (window sizes)
 * and, by a few per cent, on the individual GPU (profiles/r02_experiments/04_*, 13_*).  mdc_tune_device runs the fused
 * pass (flags must contain MDC_RECTIFY) over the caller's device batch with each candidate -- tile 128x16 / 64x32 /
 * 128x32 x 32 / 64 frames per workgroup, 7 launches each, results in d_out are valid ones -- and keeps the fastest as
 * the context's plan for all later calls: it sets MDC_OPT_TILE_COLS / _ROWS (setting those to 0 returns to the built-in
 * choice) and remembers the winning frames-per-workgroup for fused launches of a comparable size (>= nframes / 4); small
 * launches, unMapImage and a caller's own MDC_OPT_FRAMES_PER_BLOCK are not affected.  Re-planning replaces device
 * tables, so the call waits for the WHOLE device first (like mdc_set_remap): do not call it while other streams of this
 * process have work in flight that must not be delayed. */
typedef struct mdc_tune_result {
  int tile_w, tile_h, frames_per_block;
  float ms;        /* median launch time of the winner over the given batch */
  int candidates;  /* configurations that could be planned and timed */
} mdc_tune_result;
MDC_API int mdc_tune_device(mdc_ctx* ctx, const uint8_t* d_in, float* d_out, int64_t nframes, unsigned flags, void* stream,
                    mdc_tune_result* result);

/* Diagnostics: the kernel instantiation mdc_process_batch_device (pyramid_levels 0 or 1),
 * mdc_process_pyramid_batch_device (pyramid_levels = its `levels`) or mdc_undistort_batch_device_f32 (pyramid_levels = -1, flags
 * ignored) launches for `flags` with the current tables and
 * options, spelt as rocprofv3 prints it without namespaces -- so a profile line can be matched to the kernel that ran.
 * (Measurement utilities -- the synthetic sequence generator, the linear-stream yardstick -- live in libmdc_bench.so,
 * include/mdc_bench.h: they are not part of this ABI.) */
MDC_API int mdc_describe_launch(mdc_ctx* ctx, unsigned flags, int pyramid_levels, char* buf, size_t cap);

/* ---- calibration hand-over between ranks (multi-GPU) ------------------------ */

/* Serialises every table of the context (header + GInv + vignetteInv + remapX/Y)
 * into one flat blob so that rank 0 can broadcast it (RCCL / gloo -- the caller's
 * collective) and the other ranks import it bit-identically.
 * mdc_export_tables(ctx, NULL, 0, &n) returns the size. */
MDC_API int mdc_export_tables(mdc_ctx* ctx, void* blob, size_t cap, size_t* size);
MDC_API int mdc_import_tables(mdc_ctx* ctx, const void* blob, size_t size);

/* Blocks until the context's own streams (used by the *_host calls) are idle. */
MDC_API int mdc_synchronize(mdc_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* MDC_HIP_H */
