/*
 * mdc_bench.h -- C interface of libmdc_bench.so: measurement and test utilities.
 *
 * NOT part of the product (nothing of the reference corresponds to these; libmdc_hip.so / libmdc_host.so /
 * libmdc_multi.so neither link nor load this library).  Used by bench.py, tools/ and tests/ only.
 * The stream functions enqueue on `stream` (hipStream_t as void*, NULL = default stream) of HIP device `device`
 * (-1 = the calling thread's current device) without synchronising; 0 = ok, negative = error.
 */
#ifndef MDC_BENCH_H
#define MDC_BENCH_H
#include <stddef.h>
#include <stdint.h>
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

/* Synthetic sequence generator: byte i of frame f = fmix32(seed + (first_frame+f)*npix + i) >> 24. */
MDC_API int mdcb_synth_frames_device(int device, uint8_t* d_out, int64_t first_frame, int64_t nframes, int npix, uint32_t seed, void* stream);

/* A linear, arithmetic-free stream reading read_bytes from d_read (16-byte aligned) while writing write_bytes to d_write
 * with `blocks` workgroups of 256 (span = 0: grid-stride; 1: each workgroup walks its own contiguous span) -- the rate the
 * memory system of THIS box gives to a kernel's traffic mix, to normalise the kernel's own rate against; bench.py takes
 * the fastest of several (blocks, span) settings. */
MDC_API int mdcb_ceiling_mix_device(int device, const void* d_read, int64_t read_bytes, float* d_write, int64_t write_bytes, int blocks, int span,
                            void* stream);

/* A no-op kernel (`mdcb_marker_kernel`) on `stream`: bench.py --markers brackets its timed region with two of them, so that a rocprofv3
 * kernel trace / counter collection of the run can be cut to the timed launches (tools/profile_round.py). */
MDC_API int mdcb_marker_device(int device, int id, void* stream);

/* Diagnosis (tools/mall_bracket.py): a device range of repeats * chunk_bytes virtual addresses that all map ONE physical
 * allocation of chunk_bytes (HIP virtual memory management; chunk_bytes must be a multiple of *out_granularity, which is
 * also returned on the -3 "not a whole number of pages" error).  Synchronous. */
MDC_API int mdcb_alias_alloc(int device, int64_t chunk_bytes, int repeats, void** out_ptr, int64_t* out_granularity);
MDC_API int mdcb_alias_free(int device, void* ptr, int64_t chunk_bytes, int repeats);
/* Experiment (tools/alloc_probe.py): n physical chunks of chunk_bytes created one after the other and mapped into one range, chunk i of
 * the range = the (i * stride mod n)-th created (stride 1: in order).  Freed with mdcb_alias_free(device, ptr, chunk_bytes, n). */
MDC_API int mdcb_chunked_alloc(int device, int64_t chunk_bytes, int n, int stride, void** out_ptr);

#ifdef __cplusplus
}
#endif
#endif /* MDC_BENCH_H */
