// Internal to libmdc_hip.so: the context behind the opaque mdc_ctx of include/mdc_hip.h and what the translation units
// of the C ABI share.  Not installed.
//   mdc_capi.hip        lifetime, options, tables, the *_device entry points, tuning, table blobs; enqueue_process
//   mdc_plan.hip        tile / strip plans of the remap kernels (host side)
//   mdc_host_calls.hip  the synchronous host-pointer entry points (unMapImage / undistort<T> / getImage semantics)
//   mdc_pipeline.hip    the pipelined many-frame host calls (raw frames, JPEG coefficient records, JPEG streams)
#pragma once
#include "../../include/mdc_hip.h"
#include "mdc_internal.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <new>
#include <shared_mutex>
#include <string>
#include <vector>

struct mdc_ctx {
  int device = 0;
  // Locking.  `mu` guards the calibration tables, the plans and the options: the entry points that only READ them -- every
  // per-frame call, host- or device-pointer -- take it shared and run concurrently; the setters (tables, options, tuning)
  // take it exclusively (and then wait for the whole device, kernels on caller streams may still read the tables).
  // What the readers do mutate has its own small lock: the last-error string, the list of page-locked caller buffers, the
  // slots of the host-pointer calls, the pipeline of mdc_process_frames_host.
  std::shared_timed_mutex mu;
  mutable std::mutex err_mu;
  std::string err;
  std::mutex pin_mu, pipe_mu;

  // Host-pointer calls (mdc_unmap_host, mdc_undistort_host_*, mdc_process_host, mdc_distort_points_host): each call leases a
  // slot -- its own stream and staging buffers -- so that calls from several host threads overlap their copies and
  // kernels instead of queueing on one stream.  Slots are created on demand, at most kMaxSlots; a caller beyond that waits.
  struct HostSlot {
    hipStream_t stream = nullptr;
    void* d_in = nullptr;
    size_t in_cap = 0;
    float* d_out = nullptr;
    size_t out_cap = 0;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;  // the chunked strip path borrows a slot's stream as its second one
    bool busy = false;
  };
  static constexpr int kMaxSlots = 8;
  std::mutex slot_mu;
  std::condition_variable slot_cv;
  std::vector<HostSlot*> slots;

  // photometric tables
  int in_w = 0, in_h = 0;
  bool valid_gamma = false, valid_vignette = false;
  std::vector<float> h_ginv;  // 256
  std::vector<float> h_vinv;  // in_w*in_h
  float* d_luts = nullptr;    // 4 x 256: [gamma | kill<<1]
  float* d_vinv = nullptr;

  // geometric tables
  bool valid_remap = false;
  int rm_in_w = 0, rm_in_h = 0, out_w = 0, out_h = 0;
  std::vector<float> h_rx, h_ry;
  float *d_rx = nullptr, *d_ry = nullptr;

  // tile plans, see TilePlan (mdc_internal.h): [0] raw u8 frames (fused path), [1] float frames
  // (undistort<float>).  Each source type has its own tile shape and XCD placement table.
  struct SrcPlan {
    uint32_t* d_chunks = nullptr;
    int* d_nch = nullptr;
    uint32_t* d_taps = nullptr;
    int* d_order = nullptr;  // block -> tile placement table (XCD bands)
    int chunk_cap = 0, win_bytes = 0, nbuf = 2;
    int tile_w = 0, tile_h = 0, n_tiles = 0, tiles_x = 0, n_blocks = 0;
    bool tiled = false;
    int64_t staged_bytes = 0;
  } plan[2];
  // wave-private strip kernel (StripPlan, mdc_internal.h): u8 frames, remaps with about one output or more per source pixel
  struct Strip {
    uint32_t* d_chunks = nullptr;
    int* d_nch = nullptr;
    uint32_t* d_taps = nullptr;
    int* d_order = nullptr;
    int n_blocks = 0, n_tiles = 0, tiles_x = 0, win_bytes = 0, passes = 0, nbuf = 2;
    bool planned = false;
    int64_t staged_bytes = 0;
  } strip;
  int bbox[4] = {0, 0, -1, -1};
  int64_t n_black = 0;

  // options
  int opt_kernel = MDC_KERNEL_AUTO;
  int opt_fpb = 0;
  int tuned_fpb = 0;        // mdc_tune_device's pick for the u8 tiled plan; applies to launches of >= tuned_min_frames only
  int64_t tuned_min_frames = 0;
  int opt_tile_h = 0;  // 0 = automatic: the first shape of the candidate list whose windows fit
  int opt_tile_w = 0;  // 0 = automatic
  int opt_order = MDC_ORDER_BANDS;
  int opt_nbuf = 0;  // 0 = automatic
  int opt_interleave = 0;
  int opt_pin_caller = 0;  // MDC_OPT_PIN_CALLER_BUFFERS
  int opt_taper = 0;             // MDC_OPT_TAIL_TAPER: 0 = automatic (on), 1 = on, 2 = off
  int opt_zero_copy = 0;         // MDC_OPT_ZERO_COPY: 0 = automatic (on), 1 = on, 2 = off
  int opt_prefetch_streams = 0;  // MDC_OPT_PREFETCH_STREAMS: 0 = automatic (2), 1, 2
  int opt_prefetch_chunk = 0;  // MDC_OPT_PREFETCH_CHUNK: frames per prefetched chunk of the strip path; 0 = automatic, -1 = no prefetch
  int opt_dev_chunk = 0;   // MDC_OPT_DEVICE_PIPELINE_CHUNK: 0 = automatic
  int opt_dev_chunk_hint = 0;  // MDC_OPT_DEVICE_PIPELINE_CHUNK_HINT: what automatic means here (below the option and the environment)
  int opt_two_stage = 0;   // MDC_OPT_TWO_STAGE: 0 = automatic (strip kernel by source pixels per output), 1 = strip kernel whenever
                           // plannable, 2 = never

  // Caller buffers page-locked in place (opt-in): the W*H float image that the reference's two-call composition
  // moves host -> device -> host -> device (DatasetReader::internalTempBuffer, src/BenchmarkDatasetReader.h:145,222).
  // An entry is made when the same (pointer, size) shows up on two consecutive calls of one role.
  struct Pinned {
    const void* p = nullptr;
    size_t bytes = 0;
    bool ok = false;  // false = registration was refused (e.g. already page-locked): do not try again
    uint64_t used = 0;
  };
  std::vector<Pinned> pinned;
  const void* pin_candidate[2] = {nullptr, nullptr};
  size_t pin_candidate_bytes[2] = {0, 0};
  uint64_t pin_clock = 0;

  // pipelined host-frame path (mdc_process_frames_host): two chunk slots, each with its own stream
  hipStream_t pipe_stream[2] = {nullptr, nullptr};
  hipEvent_t pipe_done[2] = {nullptr, nullptr};
  hipEvent_t pipe_dec[2] = {nullptr, nullptr};  // streams: "chunk decoded" (decode stream -> output stream)
  hipStream_t pipe_up_stream = nullptr;          // streams: uploads run ahead of the decode stream on their own
  hipEvent_t pipe_up[2] = {nullptr, nullptr};   // "chunk uploaded" (upload stream -> decode stream)
  hipEvent_t pipe_huff[2] = {nullptr, nullptr}; // "stream buffer read" (decode stream -> upload stream)
  uint8_t* d_pipe_in[2] = {nullptr, nullptr};
  float* d_pipe_out[2] = {nullptr, nullptr};
  void* d_pipe_rec[2] = {nullptr, nullptr};  // JPEG coefficient records of a chunk (mdc_process_jpeg_frames_host)
  void* d_pipe_strm[2] = {nullptr, nullptr};  // JPEG streams of a chunk (mdc_process_jpeg_streams_host)
  int* d_pipe_status[2] = {nullptr, nullptr}; // their decode status words (one chunk each)
  void* d_pipe_seg[2] = {nullptr, nullptr};   // ... and the segment states of the Huffman decoder's several workgroups per frame
  int* h_pipe_status = nullptr;               // page-locked landing buffer for them (a whole call)
  size_t pipe_status_cap = 0;
  int pipe_chunk_cap = 0;  // frames d_pipe_status / d_pipe_seg are sized for
  size_t pipe_in_cap = 0, pipe_out_cap = 0, pipe_rec_cap = 0, pipe_strm_cap = 0;

  // mdc_device_alloc: buffers of a GiB or more are striped over the device's memory classes (mdc_placement.hip); what mdc_device_free
  // must undo for them, by address
  std::mutex striped_mu;
  std::map<void*, void*> striped;  // device pointer -> the allocator's arena

  // vignetteCalib: bit pattern of the largest new vignette factor of ONE vignette step.  A ring of words, one per call:
  // steps that different threads put on different streams of one context never share a word.
  static constexpr int kVcalMaxWords = 256;
  unsigned* d_vcal_max = nullptr;
  std::atomic<unsigned> vcal_max_next{0};

};
using ReadLock = std::shared_lock<std::shared_timed_mutex>;
using WriteLock = std::unique_lock<std::shared_timed_mutex>;

namespace mdc {

// status + message: sets the context's (and the calling thread's) last error, returns `code`   (mdc_capi.hip)
int fail(mdc_ctx* c, int code, const char* fmt, ...);

#define MDC_HIP(c, call)                                                                      \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) return fail((c), MDC_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
  } while (0)

struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) (void)hipSetDevice(dev);
    else prev = -1;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

// No exception leaves the C ABI: allocation failures of the host-side containers (tables, plans, pointer lists) and anything
// else unexpected become a status + message.
#define MDC_CATCH(c_)                                                                    \
  catch (const std::bad_alloc&) {                                                        \
    return fail((c_), MDC_ERR_NOMEM, "out of host memory");                              \
  }                                                                                      \
  catch (const std::exception& e_) {                                                     \
    return fail((c_), MDC_ERR_HIP, "unexpected exception: %s", e_.what());               \
  }                                                                                      \
  catch (...) {                                                                          \
    return fail((c_), MDC_ERR_HIP, "unexpected exception");                              \
  }

// ---- mdc_capi.hip ------------------------------------------------------------------------------------------------
void normalise(const mdc_ctx* c, unsigned flags, bool& g, bool& v, bool& o);  // src/PhotometricUndistorter.cpp:173-189
const float* lut_for(const mdc_ctx* c, bool g, bool o);
int frames_per_block(const mdc_ctx* c, int64_t nframes, int blocks_per_group, int target_wgs = 4800);
TilePlan tile_plan(const mdc_ctx* c, int which);
RemapArgs remap_args(const mdc_ctx* c, const float* lut, const float* vinv);
// the fused pass (or unMapImage alone) over device frames, on stream s; pyr: levels 1..3 wanted (-> *pyr_done: written by this launch)
int enqueue_process(mdc_ctx* c, const uint8_t* d_in, float* d_out, int64_t nframes, unsigned flags, hipStream_t s, float* const* pyr = nullptr,
                    bool* pyr_done = nullptr);
int enqueue_pyramid_gradients(mdc_ctx* c, const uint8_t* d_in, float* d_base, int levels, float* const* d_levels, float* const* d_dI,
                              float* const* d_abs_squared_grad, int64_t nframes, unsigned flags, int chunk_frames, hipStream_t s);
int enqueue_undistort_f32(mdc_ctx* c, const float* d_in, float* d_out, int64_t nframes, hipStream_t s);
DistortModel distort_model(const mdc_fov_model* f);

// ---- mdc_plan.hip ------------------------------------------------------------------------------------------------
int plan_tiles(mdc_ctx* c);  // (re)plans both source types and the strip kernel from the context's remap + options
void free_plan(mdc_ctx* c);

// ---- mdc_host_calls.hip ------------------------------------------------------------------------------------------
void maybe_pin(mdc_ctx* c, int role, const void* p, size_t bytes);
void unpin_all(mdc_ctx* c);
const void* device_view_raw(const mdc_ctx* c, const void* p, size_t bytes);
template <class T>
T* device_view(const mdc_ctx* c, T* p, size_t bytes) {
  return (T*)device_view_raw(c, (const void*)p, bytes);
}
bool one_host_allocation(const void* p, size_t bytes);
int ensure_stage(mdc_ctx* c, mdc_ctx::HostSlot* h, size_t in_bytes, size_t out_bytes);

// A slot of the host-pointer calls for the duration of one call (RAII).  s == nullptr: no slot could be made (error set).
struct SlotLease {
  mdc_ctx* c;
  mdc_ctx::HostSlot* s = nullptr;
  // wait == false: take a free slot (or make one) or come back empty-handed, without an error
  explicit SlotLease(mdc_ctx* ctx, bool wait = true);
  // Every host call borrows the caller's buffers for its duration only and hands the slot (its stream, its staging
  // buffers) to the next caller: whatever way the call ends -- an early return after an asynchronous copy was enqueued
  // included --, nothing of it may still be in flight.  A call that has synchronised successfully says drained().
  bool in_flight = true;
  void drained() { in_flight = false; }
  ~SlotLease();
  SlotLease(const SlotLease&) = delete;
  SlotLease& operator=(const SlotLease&) = delete;
};

}  // namespace mdc
