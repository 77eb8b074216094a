/*
 * mdc_multi.h -- C ABI of libmdc_multi.so: ONE process driving every MI355X of a node.
 *
 * The path shards by frames (reference src/BenchmarkDatasetReader.h:188-243 reads only immutable
 * tables plus the frame itself): frame f of a sequence belongs to device f % N, there is no
 * exchange step on the data path, and the ONLY collective is the one-time hand-over of the
 * calibration tables -- here an RCCL broadcast over xGMI inside the process (ncclCommInitAll, one
 * communicator per device, ncclGroupStart / ncclBroadcast x N / ncclGroupEnd), so that C++ callers
 * (playDataset, DSO-style front ends) get all GPUs without torchrun.  SURVEY.md section 8(b)
 * "mdc_bcast_tables(ctx[], nranks)" and 8(e); BASELINE.json configs[3].
 *
 * One mdc_ctx (include/mdc_hip.h) per device; everything per-device keeps going through that ABI.
 */
#ifndef MDC_MULTI_H
#define MDC_MULTI_H

#include "mdc_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mdc_multi mdc_multi;

/* One context per listed HIP device (devices == NULL: devices 0 .. ndev-1; ndev <= 0: every visible
 * device) and one RCCL communicator per context.  MDC_ERR_NO_DEVICE without a GPU. */
MDC_API int mdc_multi_create(const int* devices, int ndev, mdc_multi** out);
MDC_API void mdc_multi_destroy(mdc_multi* m);
MDC_API int mdc_multi_size(const mdc_multi* m);                 /* N */
MDC_API mdc_ctx* mdc_multi_ctx(mdc_multi* m, int rank);         /* rank's context, for the per-device calls of mdc_hip.h */
MDC_API int mdc_multi_device(const mdc_multi* m, int rank);     /* rank's HIP device ordinal */
MDC_API const char* mdc_multi_last_error(const mdc_multi* m);   /* m may be NULL (creation errors) */

/* The tables of rank `root` (GInv, vignetteInv, remapX/Y as mdc_set_photometric / mdc_set_remap or
 * mdc_bind_objects put them there) to every other rank: one RCCL broadcast group on the devices'
 * streams, then every rank imports the blob it received -- bit-identical tables everywhere. */
MDC_API int mdc_multi_bcast_tables(mdc_multi* m, int root);

/* Round-robin sharding arithmetic: frames a rank owns out of nframes_total, and a frame's owner / local index. */
MDC_API int64_t mdc_multi_frames_of_rank(const mdc_multi* m, int64_t nframes_total, int rank);

/* The whole sequence through the fused pass, every device on its own stream, launched by one host
 * thread per device: d_in[r] holds rank r's frames (r, r+N, r+2N, ...) back to back, d_out[r] receives
 * its results in the same order.  Returns after the launches; mdc_multi_synchronize waits. */
MDC_API int mdc_multi_process_sequence_device(mdc_multi* m, const uint8_t* const* d_in, float* const* d_out,
                                      int64_t nframes_total, unsigned flags);
MDC_API int mdc_multi_synchronize(mdc_multi* m);

/* Ranks of rank's RCCL communicator (ncclCommCount): == mdc_multi_size() for a healthy object; -1 on error. */
MDC_API int mdc_multi_comm_count(const mdc_multi* m, int rank);
/* The stream (hipStream_t as void*) rank's launches go on -- for callers that enqueue their own work in order with them. */
MDC_API void* mdc_multi_stream(mdc_multi* m, int rank);

#ifdef __cplusplus
}
#endif
#endif /* MDC_MULTI_H */
