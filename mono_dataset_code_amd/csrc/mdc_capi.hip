// libmdc_hip.so -- implementation of the C ABI in include/mdc_hip.h.
//
// Host-side responsibilities only: device copies of the calibration tables, the
// tile plan of the LDS-staged remap kernel, flag normalisation exactly as
// src/PhotometricUndistorter.cpp:173-189 (reference repo), staging for the
// host-pointer calls.  There is NO CPU implementation of the per-frame maths in
// this library: without a HIP device every entry point fails with
// MDC_ERR_NO_DEVICE.
#include "../../include/mdc_hip.h"
#include "mdc_internal.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <vector>

using namespace mdc;

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
  int* h_pipe_status = nullptr;               // page-locked landing buffer for them (a whole call)
  size_t pipe_status_cap = 0;
  size_t pipe_in_cap = 0, pipe_out_cap = 0, pipe_rec_cap = 0, pipe_strm_cap = 0;

  // vignetteCalib: bit pattern of the largest new vignette factor of ONE vignette step.  A ring of words, one per call:
  // steps that different threads put on different streams of one context never share a word.
  static constexpr int kVcalMaxWords = 256;
  unsigned* d_vcal_max = nullptr;
  std::atomic<unsigned> vcal_max_next{0};

};
using ReadLock = std::shared_lock<std::shared_timed_mutex>;
using WriteLock = std::unique_lock<std::shared_timed_mutex>;

namespace {

thread_local std::string g_create_err;
// mdc_last_error(ctx) returns the calling thread's own last failure on that context if it had one (several threads may
// use one context), else the context's most recent one
thread_local std::string t_err, t_err_other;
thread_local const mdc_ctx* t_err_ctx = nullptr;

int fail(mdc_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) {
    {
      std::lock_guard<std::mutex> lk(c->err_mu);
      c->err = buf;
    }
    t_err = buf;
    t_err_ctx = c;
  } else {
    g_create_err = buf;
  }
  return code;
}

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

// The reference's flag degradation, src/PhotometricUndistorter.cpp:173-189.
void normalise(const mdc_ctx* c, unsigned flags, bool& g, bool& v, bool& o) {
  g = (flags & MDC_GAMMA) != 0;
  v = (flags & MDC_VIGNETTE) != 0;
  o = (flags & MDC_KILL_OVEREXPOSED) != 0;
  if (!c->valid_gamma && g) g = false;
  if (!c->valid_vignette && v) v = false;
  if (!g && v) {
    v = false;
    g = false;
  }
}

const float* lut_for(const mdc_ctx* c, bool g, bool o) { return c->d_luts + 256 * ((g ? 1 : 0) | (o ? 2 : 0)); }

int upload_luts(mdc_ctx* c) {
  std::vector<float> l(4 * 256);
  for (int var = 0; var < 4; var++)
    for (int b = 0; b < 256; b++) {
      float x = (var & 1) ? (c->valid_gamma ? c->h_ginv[b] : (float)b) : (float)b;
      if ((var & 2) && b == 255) x = std::numeric_limits<float>::quiet_NaN();
      l[var * 256 + b] = x;
    }
  if (!c->d_luts) MDC_HIP(c, hipMalloc(&c->d_luts, l.size() * sizeof(float)));
  MDC_HIP(c, hipMemcpy(c->d_luts, l.data(), l.size() * sizeof(float), hipMemcpyHostToDevice));
  return MDC_OK;
}

int frames_per_block(const mdc_ctx* c, int64_t nframes, int blocks_per_group, int target_wgs = 4800) {
  if (c->opt_fpb > 0) return (int)std::min<int64_t>(c->opt_fpb, std::max<int64_t>(nframes, 1));
  // enough workgroups to fill 256 CUs several times over (tail), yet >= 8 frames per
  // workgroup so the per-workgroup table reads stay amortised
  int64_t groups = std::max<int64_t>(1, (target_wgs + blocks_per_group - 1) / std::max(1, blocks_per_group));
  groups = std::min<int64_t>(groups, std::max<int64_t>(1, nframes / 8));
  int64_t fpb = (nframes + groups - 1) / groups;
  // few, large tiles (128 x 32: 80 blocks per frame group): not below 32 frames per workgroup as long as
  // that still leaves >= 2048 workgroups -- the per-workgroup prologue costs about two frames' time
  if (fpb < 32 && (int64_t)blocks_per_group * ((nframes + 31) / 32) >= 2048) fpb = 32;
  // ... and not above 64: workgroups that live for hundreds of frames drift apart and fall into lockstep phases
  // (a 50,000-frame launch with 1563 frames per workgroup ran 20 % slower per frame than with 32..64)
  if (target_wgs == 4800 && fpb > 64) fpb = 64;
  return (int)fpb;
}

// Placement table of the tiled kernel: entry b = tile run by block b of a frame group, -1 = none.
// The dispatcher deals blocks round-robin over the 8 XCDs (block b -> XCD b % 8, slot b / 8), so
// XCD k runs the tiles of entries k, k+8, k+16, ...  Neighbouring tiles share source lines (halo
// rows, 128-byte lines straddling a tile border); they should meet in ONE XCD's L2.
//   MDC_ORDER_BANDS     row-major runs of ceil(n/8) tiles per XCD
//   MDC_ORDER_ROWS      whole tile rows per XCD, as even as the row count allows (no horizontal
//                       neighbours split; XCDs with a row less idle at the end of a frame group)
//   MDC_ORDER_IDENTITY  block b = tile b: neighbours land on different XCDs (diagnosis: worst case)
//   MDC_ORDER_BLOCKS2D  the tile grid cut into 8 rectangles by recursive bisection of the longer side
//                       (least shared halo perimeter between XCDs; the rectangles differ in size by up to
//                       one row / column, XCDs with fewer tiles get padding slots)
static void bisect(int x0, int y0, int x1, int y1, int parts, int tx, std::vector<std::vector<int>>& out) {
  if (parts == 1) {
    std::vector<int> v;
    for (int y = y0; y < y1; y++)
      for (int x = x0; x < x1; x++) v.push_back(y * tx + x);
    out.push_back(v);
    return;
  }
  if (x1 - x0 > y1 - y0) {
    const int xm = x0 + (x1 - x0 + 1) / 2;
    bisect(x0, y0, xm, y1, parts / 2, tx, out);
    bisect(xm, y0, x1, y1, parts / 2, tx, out);
  } else {
    const int ym = y0 + (y1 - y0 + 1) / 2;
    bisect(x0, y0, x1, ym, parts / 2, tx, out);
    bisect(x0, ym, x1, y1, parts / 2, tx, out);
  }
}

std::vector<int> tile_order(int tx, int ty, int mode) {
  const int n = tx * ty;
  std::vector<std::vector<int>> per_xcd(8);
  if (mode == MDC_ORDER_BLOCKS2D && tx * ty >= 8) {
    per_xcd.clear();
    bisect(0, 0, tx, ty, 8, tx, per_xcd);
  } else if (mode == MDC_ORDER_IDENTITY) {
    for (int t = 0; t < n; t++) per_xcd[t % 8].push_back(t);
  } else if (mode == MDC_ORDER_ROWS && ty >= 8) {
    int r = 0;
    for (int k = 0; k < 8; k++) {
      const int rows = ty / 8 + (k < ty % 8 ? 1 : 0);
      for (int y = r; y < r + rows; y++)
        for (int x = 0; x < tx; x++) per_xcd[k].push_back(y * tx + x);
      r += rows;
    }
  } else {
    const int per = (n + 7) / 8;
    for (int t = 0; t < n; t++) per_xcd[t / per].push_back(t);
  }
  size_t slots = 0;
  for (const auto& v : per_xcd) slots = std::max(slots, v.size());
  std::vector<int> order(slots * 8, -1);
  for (int k = 0; k < 8; k++)
    for (size_t j = 0; j < per_xcd[k].size(); j++) order[j * 8 + k] = per_xcd[k][j];
  return order;
}

// Plan of the tiled kernel (see TilePlan): per tile the exact source window as a list of
// 16-byte chunks, per output the LDS offsets of its two tap rows.  Fails (tiled = false)
// when rows of the frame are not whole 16-byte chunks or a window is too large for LDS.
void free_src_plan(mdc_ctx::SrcPlan& pl) {
  for (void** p : {(void**)&pl.d_chunks, (void**)&pl.d_nch, (void**)&pl.d_taps, (void**)&pl.d_order})
    if (*p) {
      (void)hipFree(*p);
      *p = nullptr;
    }
  pl.tiled = false;
  pl.staged_bytes = 0;
  pl.n_tiles = pl.tiles_x = pl.n_blocks = 0;
}
void free_strip_plan(mdc_ctx::Strip& st) {
  for (void** p : {(void**)&st.d_chunks, (void**)&st.d_nch, (void**)&st.d_taps, (void**)&st.d_order})
    if (*p) {
      (void)hipFree(*p);
      *p = nullptr;
    }
  st.planned = false;
  st.staged_bytes = 0;
  st.n_blocks = st.n_tiles = st.tiles_x = 0;
}
void free_plan(mdc_ctx* c) {
  for (auto& pl : c->plan) {
    free_src_plan(pl);
  }
  free_strip_plan(c->strip);
}

template <typename T>
int upload(mdc_ctx* c, T** dst, const std::vector<T>& v) {
  MDC_HIP(c, hipMalloc(dst, std::max<size_t>(v.size(), 1) * sizeof(T)));
  if (!v.empty()) MDC_HIP(c, hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return MDC_OK;
}

// The plan for source pixels of `es` bytes (1 = raw u8 frames with the LUT replicas in LDS,
// 4 = float frames, no LUT): a 16-byte chunk holds 16 / es pixels.  Leaves pl.tiled = false when
// frame rows are not whole chunks or a window is too large.
int plan_source(mdc_ctx* c, int es, int kTileW, int kTileH, mdc_ctx::SrcPlan& pl) {
  const int ow = c->out_w, oh = c->out_h, iw = c->rm_in_w;
  const int kTileThreads = tile_threads(kTileW, kTileH);
  pl.staged_bytes = 0;
  const int tx = (ow + kTileW - 1) / kTileW, ty = (oh + kTileH - 1) / kTileH;
  const int n_tiles = tx * ty;
  const int ppc = 16 / es;  // pixels per chunk
  const bool lut = es == 1;
  // whole 16-byte chunks per frame row; one frame within the 32-bit lane offsets of the buffer descriptors
  const char* why = "frame rows are not whole chunks / frame too large";
  bool ok = (iw % ppc == 0) && (int64_t)iw * c->rm_in_h * es < (int64_t)kOutside && (int64_t)ow * oh * 4 < (int64_t)kOutside;
  if (tile_rpt(kTileW, kTileH) != 4 && es != 1) ok = false;  // the 8-rows-per-thread tiles exist for raw u8 frames only
  // the 960-/1024-thread tiles derive the output offsets of rows 1..3 from row 0 (kOutsideLean, mdc_kernels.hip)
  if ((kTileThreads >= 960 || tile_rpt(kTileW, kTileH) > 4) && (int64_t)ow * (oh + kTileH) * 4 >= 0xc0000000ll) ok = false;
  std::vector<std::vector<uint32_t>> chunks(n_tiles);
  std::vector<int> nch(n_tiles, 0);
  std::vector<uint32_t> taps((size_t)ow * oh, 0u);
  for (int t = 0; t < n_tiles && ok; t++) {
    const int bx = (t % tx) * kTileW, by = (t / tx) * kTileH;
    const int x1 = std::min(bx + kTileW, ow), y1 = std::min(by + kTileH, oh);
    int y_lo = std::numeric_limits<int>::max(), y_hi = -1;
    for (int y = by; y < y1; y++)
      for (int x = bx; x < x1; x++) {
        const float xx = c->h_rx[(size_t)y * ow + x], yy = c->h_ry[(size_t)y * ow + x];
        if (xx < 0) continue;
        y_lo = std::min(y_lo, (int)yy);
        y_hi = std::max(y_hi, (int)yy + 1);
      }
    if (y_hi < 0) continue;  // every output black: no window
    // Exact chunk SET per source row (not one run from the leftmost to the rightmost tap: the source footprint of a wide,
    // flat tile is a bowed band that touches a row in two separate places).  pos[row][chunk] = index of the chunk in the
    // tile's list, -1 = not staged.  A tap pair (xi, xi+1) marks both bytes' chunks, so chunks that are neighbours in a
    // frame row and both used are neighbours in the list too: the pair stays contiguous in LDS.
    const int cpr = iw / ppc;  // chunks per frame row
    const int nrows = y_hi - y_lo + 1;
    std::vector<int> pos((size_t)nrows * cpr, -1);
    for (int y = by; y < y1; y++)
      for (int x = bx; x < x1; x++) {
        const float xx = c->h_rx[(size_t)y * ow + x], yy = c->h_ry[(size_t)y * ow + x];
        if (xx < 0) continue;
        const int xi = (int)xx, yi = (int)yy;
        for (int dy = 0; dy < 2; dy++) {
          int* pr = &pos[(size_t)(yi + dy - y_lo) * cpr];
          pr[xi / ppc] = 0;
          pr[(xi + 1) / ppc] = 0;
        }
      }
    for (int k = 0; k < nrows; k++)
      for (int ch = 0; ch < cpr; ch++) {
        int& q = pos[(size_t)k * cpr + ch];
        if (q < 0) continue;
        q = (int)chunks[t].size();
        chunks[t].push_back((uint32_t)(((y_lo + k) * iw + ch * ppc) * es));
      }
    nch[t] = (int)chunks[t].size();
    if (nch[t] > (lut ? kTileMaxChunks : kTileMaxChunksF32) * kTileThreads || nch[t] * 16 > 65535) {
      ok = false;
      why = "a window has too many chunks";
    }
    pl.staged_bytes += (int64_t)nch[t] * 16;
    for (int y = by; y < y1 && ok; y++)
      for (int x = bx; x < x1; x++) {
        const float xx = c->h_rx[(size_t)y * ow + x], yy = c->h_ry[(size_t)y * ow + x];
        if (xx < 0) continue;
        const int xi = (int)xx, yi = (int)yy;
        const int p0 = pos[(size_t)(yi - y_lo) * cpr + xi / ppc], p1 = pos[(size_t)(yi + 1 - y_lo) * cpr + xi / ppc];
        taps[(size_t)y * ow + x] = (uint32_t)(p0 * 16 + (xi % ppc) * es) | ((uint32_t)(p1 * 16 + (xi % ppc) * es) << 16);
      }
  }
  // every tile's chunk list is padded (kOutside) to the kernel's maximum of staging rounds: the kernel loads
  // all of them unconditionally, first thing, before it knows the tile's chunk count
  const int cap = (lut ? kTileMaxChunks : kTileMaxChunksF32) * kTileThreads;
  int nch_max = 1;
  for (int t = 0; t < n_tiles; t++) nch_max = std::max(nch_max, nch[t]);
  const int win_bytes = (nch_max * 16 + 1023) & ~1023;  // a wave's DMA destination is 1 KiB aligned
  // Window buffers: as many frames staged ahead as LDS allows WITHOUT lowering the number of
  // workgroups per CU that two buffers permit (occupancy first, then depth), at most 4.
  const int wg_per_cu =
      std::max<int>(1, std::min<size_t>(kLdsPerCU / tiled_lds_bytes(win_bytes, 2, lut), 2048 / kTileThreads));
  const int nbuf_max = kTileThreads > 512 ? 3 : 4;
  int nbuf = 2;
  while (nbuf < nbuf_max && tiled_lds_bytes(win_bytes, nbuf + 1, lut) * wg_per_cu <= kLdsPerCU) nbuf++;
  if (c->opt_nbuf >= 2) nbuf = std::min(c->opt_nbuf, nbuf_max);
  if (tile_rpt(kTileW, kTileH) > 4) nbuf = 3;  // the only instantiation of the 8-rows-per-thread tiles (mdc_kernels.hip: launch_tiled_buf)
  if (tiled_lds_bytes(win_bytes, nbuf, lut) > kLdsPerCU) {
    ok = false;
    why = "windows do not fit LDS";
  }
  if (!ok) {
    if (getenv("MDC_DEBUG_PLAN")) fprintf(stderr, "mdc plan %dx%d (element size %d): not plannable: %s\n", kTileW, kTileH, es, why);
    return MDC_OK;
  }
  std::vector<uint32_t> flat((size_t)n_tiles * cap, kOutside);
  for (int t = 0; t < n_tiles; t++) std::copy(chunks[t].begin(), chunks[t].end(), flat.begin() + (size_t)t * cap);
  int rc;
  if ((rc = upload(c, &pl.d_chunks, flat)) != MDC_OK || (rc = upload(c, &pl.d_nch, nch)) != MDC_OK ||
      (rc = upload(c, &pl.d_taps, taps)) != MDC_OK)
    return rc;
  const std::vector<int> order = tile_order(tx, ty, c->opt_order);
  if ((rc = upload(c, &pl.d_order, order)) != MDC_OK) return rc;
  pl.n_blocks = (int)order.size();
  pl.n_tiles = n_tiles;
  pl.tiles_x = tx;
  pl.tile_w = kTileW;
  pl.tile_h = kTileH;
  pl.chunk_cap = cap;
  pl.win_bytes = win_bytes;
  pl.nbuf = nbuf;
  pl.tiled = true;
  return MDC_OK;
}

// Plan of the wave-private strip kernel (StripPlan): per 128 x 8 output tile the exact source window as a dense list of
// 16-byte chunks (<= kStripChunkCap), per output the byte offsets of its two tap rows inside the wave's FLOAT window.
// Planned when the remap stages fewer source pixels than it has outputs (config 5's scale-1 rectification, magnifying
// remaps) or on request (MDC_OPT_TWO_STAGE = 1); leaves st.planned = false when a window is too large, frame rows are not
// whole chunks, or the output height is not a multiple of 8 (rows are addressed through the store's scalar offset,
// which the hardware's range check does not cover).
int plan_strip(mdc_ctx* c) {
  mdc_ctx::Strip& st = c->strip;
  free_strip_plan(st);
  if (c->opt_two_stage == 2) return MDC_OK;
  const int ow = c->out_w, oh = c->out_h, iw = c->rm_in_w;
  constexpr int TW = kStripTileW, TH = kStripTileH;
  if (iw % 16 != 0 || oh % TH != 0 || (int64_t)iw * c->rm_in_h >= (int64_t)kOutside || (int64_t)ow * (oh + TH) * 4 >= 0xc0000000ll) return MDC_OK;
  const int tx = (ow + TW - 1) / TW, ty = oh / TH, n_tiles = tx * ty;
  std::vector<uint32_t> flat((size_t)n_tiles * kStripChunkCap, kOutside);
  std::vector<int> nch(n_tiles, 0);
  std::vector<uint32_t> taps((size_t)ow * oh, 0u);
  struct Row {
    int lo = std::numeric_limits<int>::max(), hi = -1, x0 = 0, lds = 0;
  };
  int nch_max = 1;
  int64_t staged = 0;
  for (int t = 0; t < n_tiles; t++) {
    const int bx = (t % tx) * TW, by = (t / tx) * TH;
    const int x1 = std::min(bx + TW, ow), y1 = by + TH;
    int y_lo = std::numeric_limits<int>::max(), y_hi = -1;
    for (int y = by; y < y1; y++)
      for (int x = bx; x < x1; x++) {
        const float xx = c->h_rx[(size_t)y * ow + x], yy = c->h_ry[(size_t)y * ow + x];
        if (xx < 0) continue;
        y_lo = std::min(y_lo, (int)yy);
        y_hi = std::max(y_hi, (int)yy + 1);
      }
    if (y_hi < 0) continue;  // every output black
    std::vector<Row> rows(y_hi - y_lo + 1);
    for (int y = by; y < y1; y++)
      for (int x = bx; x < x1; x++) {
        const float xx = c->h_rx[(size_t)y * ow + x], yy = c->h_ry[(size_t)y * ow + x];
        if (xx < 0) continue;
        const int xi = (int)xx, yi = (int)yy;
        for (int dy = 0; dy < 2; dy++) {
          Row& r = rows[yi + dy - y_lo];
          r.lo = std::min(r.lo, xi);
          r.hi = std::max(r.hi, xi + 1);
        }
      }
    int n = 0;
    for (size_t k = 0; k < rows.size(); k++) {
      Row& r = rows[k];
      if (r.hi < 0) continue;
      r.x0 = r.lo - r.lo % 16;
      r.lds = n * 16;
      const int cnt = (r.hi - r.x0) / 16 + 1;
      if (r.x0 + cnt * 16 > iw || n + cnt > kStripChunkCap) return MDC_OK;  // not plannable: the workgroup kernels keep the job
      for (int j = 0; j < cnt; j++) flat[(size_t)t * kStripChunkCap + n + j] = (uint32_t)((y_lo + (int)k) * iw + r.x0 + j * 16);
      n += cnt;
    }
    nch[t] = n;
    nch_max = std::max(nch_max, n);
    staged += (int64_t)n * 16;
    for (int y = by; y < y1; y++)
      for (int x = bx; x < x1; x++) {
        const float xx = c->h_rx[(size_t)y * ow + x], yy = c->h_ry[(size_t)y * ow + x];
        if (xx < 0) continue;
        const int xi = (int)xx, yi = (int)yy;
        const Row &r0 = rows[yi - y_lo], &r1 = rows[yi + 1 - y_lo];
        taps[(size_t)y * ow + x] = (uint32_t)(4 * (r0.lds + xi - r0.x0)) | ((uint32_t)(4 * (r1.lds + xi - r1.x0)) << 16);
      }
  }
  const double src_per_out = (double)staged / std::max<double>(1.0, (double)ow * oh);
  if (c->opt_two_stage != 1 && src_per_out >= 1.0) return MDC_OK;
  const int win = (nch_max * 16 + 63) & ~63;
  const int need = (4 * nch_max + 63) / 64;  // convert passes
  const int passes = need <= 2 ? 2 : need <= 3 ? 3 : need <= 4 ? 4 : need <= 5 ? 5 : 8;
  const int nbuf = c->opt_nbuf >= 1 && c->opt_nbuf <= 4 ? c->opt_nbuf : 2;
  if (strip_lds_bytes(win, nbuf, kStripWaves) > kLdsPerCU) return MDC_OK;
  int rc;
  if ((rc = upload(c, &st.d_chunks, flat)) != MDC_OK || (rc = upload(c, &st.d_nch, nch)) != MDC_OK || (rc = upload(c, &st.d_taps, taps)) != MDC_OK)
    return rc;
  // groups of kStripWaves consecutive tiles (row-major: neighbours along a tile row); XCD placement as for the workgroup tiles
  const int n_groups = (n_tiles + kStripWaves - 1) / kStripWaves;
  const int gx = std::max(1, tx / kStripWaves);
  const std::vector<int> order = (tx % kStripWaves == 0) ? tile_order(gx, n_groups / gx, c->opt_order) : tile_order(n_groups, 1, MDC_ORDER_BANDS);
  if ((rc = upload(c, &st.d_order, order)) != MDC_OK) return rc;
  st.n_blocks = (int)order.size();
  st.n_tiles = n_tiles;
  st.tiles_x = tx;
  st.win_bytes = win;
  st.passes = passes;
  st.nbuf = nbuf;
  st.staged_bytes = staged;
  st.planned = true;
  return MDC_OK;
}

// Plans of the tiled kernels for the current remap: tile grid, XCD placement, source bounding
// box, one SrcPlan per source pixel type.
int plan_tiles(mdc_ctx* c) {
  c->n_black = 0;
  c->bbox[0] = c->bbox[1] = std::numeric_limits<int>::max();
  c->bbox[2] = c->bbox[3] = -1;
  free_plan(c);
  const int ow = c->out_w, oh = c->out_h;
  for (size_t i = 0; i < (size_t)ow * oh; i++) {
    const float xx = c->h_rx[i], yy = c->h_ry[i];
    if (xx < 0) {
      c->n_black++;
      continue;
    }
    c->bbox[0] = std::min(c->bbox[0], (int)xx);
    c->bbox[2] = std::max(c->bbox[2], (int)xx + 1);
    c->bbox[1] = std::min(c->bbox[1], (int)yy);
    c->bbox[3] = std::max(c->bbox[3], (int)yy + 1);
  }
  if (c->bbox[2] < 0) c->bbox[0] = c->bbox[1] = 0;
  // Tile shape per source type: the requested one, or the first candidate whose windows fit (strongly
  // distorting cameras need the taller tiles: their windows are too wide for the staging rounds of the
  // smaller workgroups).  Both lists are in order of measured speed on the bench camera (tools/sweep.py,
  // tools/rate_undistort_f32.py).
  // (128 x 16 first: measured 5-7 % faster than 64 x 32 on the bench camera -- a 64-wide tile spans ~86 source
  // bytes, less than one 128-byte line, so nearly every line is fetched by two workgroups; at 128 columns far
  // fewer are.  profiles/r02_experiments/)
  static const TileShape cand_u8[] = {{128, 16}, {64, 32}, {128, 32}, {64, 64}, {64, 60}, {64, 16}};
  static const TileShape cand_f32[] = {{128, 16}, {64, 32}, {64, 16}, {128, 32}, {64, 64}, {64, 60}};  // 0.66 / 0.63 / 0.60 / 0.60 / 0.54 of 8 TB/s
  for (int which = 0; which < 2; which++) {
    const TileShape* cand = which == 0 ? cand_u8 : cand_f32;
    const bool forced = c->opt_tile_h != 0 || c->opt_tile_w != 0;
    for (int k = 0; k < 6; k++) {
      const int tw = c->opt_tile_w ? c->opt_tile_w : cand[k].w, th = c->opt_tile_h ? c->opt_tile_h : cand[k].h;
      if (forced && (tw != cand[k].w || th != cand[k].h)) continue;  // a forced dimension filters the list
      free_src_plan(c->plan[which]);
      const int rc = plan_source(c, which == 0 ? 1 : 4, tw, th, c->plan[which]);
      if (rc != MDC_OK) return rc;
      if (c->plan[which].tiled) break;
    }
  }
  return plan_strip(c);
}

// role 0: image_out of unMapImage, role 1: input of undistort<float>
void maybe_pin(mdc_ctx* c, int role, const void* p, size_t bytes) {
  if (!c->opt_pin_caller || bytes < (256u << 10)) return;
  std::lock_guard<std::mutex> plk(c->pin_mu);
  for (auto& e : c->pinned)
    if (e.p == p && e.bytes == bytes) {
      e.used = ++c->pin_clock;
      return;
    }
  if (c->pin_candidate[role] != p || c->pin_candidate_bytes[role] != bytes) {  // first sighting: remember only
    c->pin_candidate[role] = p;
    c->pin_candidate_bytes[role] = bytes;
    return;
  }
  constexpr size_t kMaxEntries = 8;
  if (c->pinned.size() >= kMaxEntries) {  // least recently used entry goes
    size_t lru = 0;
    for (size_t i = 1; i < c->pinned.size(); i++)
      if (c->pinned[i].used < c->pinned[lru].used) lru = i;
    if (c->pinned[lru].ok) (void)hipHostUnregister(const_cast<void*>(c->pinned[lru].p));
    c->pinned.erase(c->pinned.begin() + (long)lru);
  }
  for (auto& e : c->pinned)  // an overlapping older registration (the caller re-used part of the range)
    if (e.ok && (const char*)p < (const char*)e.p + e.bytes && (const char*)e.p < (const char*)p + bytes) {
      (void)hipHostUnregister(const_cast<void*>(e.p));
      e.ok = false;
    }
  mdc_ctx::Pinned e;
  e.p = p;
  e.bytes = bytes;
  e.ok = hipHostRegister(const_cast<void*>(p), bytes, hipHostRegisterDefault) == hipSuccess;
  if (!e.ok) (void)hipGetLastError();  // refused (already page-locked, ...): plain copies keep working
  e.used = ++c->pin_clock;
  c->pinned.push_back(e);
}
void unpin_all(mdc_ctx* c) {
  std::lock_guard<std::mutex> plk(c->pin_mu);
  for (auto& e : c->pinned)
    if (e.ok) (void)hipHostUnregister(const_cast<void*>(e.p));
  c->pinned.clear();
  c->pin_candidate[0] = c->pin_candidate[1] = nullptr;
}

// Zero copy (MDC_OPT_ZERO_COPY): a host buffer that is page-locked and mapped into the device's address space (hipHostMalloc
// -- mdc_host_alloc, the reader's rings and image pool --, hipHostRegister) is handed to the kernels as it is: they read the
// frame / write the result over PCIe themselves, both directions at once, instead of copy in -> kernel -> copy out.  Returns
// the device's view of [p, p + bytes) or nullptr (pageable memory, a range that leaves its allocation, zero copy off).
// Asked of the runtime on every call -- nothing is remembered about a caller's memory.
template <class T>
T* device_view(const mdc_ctx* c, T* p, size_t bytes) {
  if (c->opt_zero_copy == 2 || !p || bytes == 0) return nullptr;
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, (const void*)p) != hipSuccess) {
    (void)hipGetLastError();  // pageable memory: not an error of ours
    return nullptr;
  }
  if (a.type != hipMemoryTypeHost || !a.devicePointer) return nullptr;
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)a.devicePointer) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  const uintptr_t lo = (uintptr_t)a.devicePointer, b0 = (uintptr_t)base;
  if (lo < b0 || lo + bytes > b0 + size) return nullptr;
  return (T*)a.devicePointer;
}

// true when [p, p + bytes) lies inside ONE page-locked allocation the runtime knows (whatever MDC_OPT_ZERO_COPY says): the
// bytes between two buffers of such a range are readable
bool one_host_allocation(const void* p, size_t bytes) {
  if (!p || bytes == 0) return false;
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  if (a.type != hipMemoryTypeHost || !a.devicePointer) return false;
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)a.devicePointer) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  const uintptr_t lo = (uintptr_t)a.devicePointer, b0 = (uintptr_t)base;
  return lo >= b0 && lo + bytes <= b0 + size;
}

// A slot of the host-pointer calls for the duration of one call (RAII).  s == nullptr: no slot could be made (error set).
struct SlotLease {
  mdc_ctx* c;
  mdc_ctx::HostSlot* s = nullptr;
  // wait == false: take a free slot (or make one) or come back empty-handed, without an error
  explicit SlotLease(mdc_ctx* ctx, bool wait = true) : c(ctx) {
    std::unique_lock<std::mutex> lk(c->slot_mu);
    for (;;) {
      for (mdc_ctx::HostSlot* h : c->slots)
        if (!h->busy) {
          h->busy = true;
          s = h;
          return;
        }
      if ((int)c->slots.size() < mdc_ctx::kMaxSlots) {
        mdc_ctx::HostSlot* h = new mdc_ctx::HostSlot();
        const hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
          delete h;
          if (wait) fail(c, MDC_ERR_HIP, "hipStreamCreateWithFlags: %s", hipGetErrorString(e));
          return;
        }
        h->busy = true;
        c->slots.push_back(h);
        s = h;
        return;
      }
      if (!wait) return;
      c->slot_cv.wait(lk);
    }
  }
  // Every host call borrows the caller's buffers for its duration only and hands the slot (its stream, its staging
  // buffers) to the next caller: whatever way the call ends -- an early return after an asynchronous copy was enqueued
  // included --, nothing of it may still be in flight.  A call that has synchronised successfully says drained().
  bool in_flight = true;
  void drained() { in_flight = false; }
  ~SlotLease() {
    if (!s) return;
    if (in_flight) (void)hipStreamSynchronize(s->stream);
    {
      std::lock_guard<std::mutex> lk(c->slot_mu);
      s->busy = false;
    }
    c->slot_cv.notify_one();
  }
  SlotLease(const SlotLease&) = delete;
  SlotLease& operator=(const SlotLease&) = delete;
};

int ensure_stage(mdc_ctx* c, mdc_ctx::HostSlot* h, size_t in_bytes, size_t out_bytes) {
  if (in_bytes > h->in_cap) {
    if (h->d_in) (void)hipFree(h->d_in);
    h->d_in = nullptr;
    h->in_cap = 0;
    MDC_HIP(c, hipMalloc(&h->d_in, in_bytes));
    h->in_cap = in_bytes;
  }
  if (out_bytes > h->out_cap) {
    if (h->d_out) (void)hipFree(h->d_out);
    h->d_out = nullptr;
    h->out_cap = 0;
    MDC_HIP(c, hipMalloc(&h->d_out, out_bytes));
    h->out_cap = out_bytes;
  }
  return MDC_OK;
}

TilePlan tile_plan(const mdc_ctx* c, int which) {
  const mdc_ctx::SrcPlan& pl = c->plan[which];
  return TilePlan{pl.d_chunks, pl.d_nch, pl.d_taps, pl.d_order, pl.n_blocks, pl.n_tiles, pl.tiles_x, pl.tile_w, pl.tile_h,
                  pl.chunk_cap, pl.win_bytes, pl.nbuf, c->n_black > 0, c->opt_interleave != 0, c->opt_taper != 2};
}

RemapArgs remap_args(const mdc_ctx* c, const float* lut, const float* vinv) {
  RemapArgs a;
  a.lut = lut;
  a.vinv = vinv;
  a.rx = c->d_rx;
  a.ry = c->d_ry;
  a.in_w = c->rm_in_w;
  a.in_h = c->rm_in_h;
  a.out_w = c->out_w;
  a.out_h = c->out_h;
  return a;
}

// Strip path, chunked launches with prefetch (enqueue_process): bytes of the source bounding box widened to whole 128-byte
// lines, and the frames per chunk.  48 frames of the bench camera (0.65 MB of box each) measured best -- beyond ~64 frames
// the prefetched lines no longer survive until they are used (the launch's own output passes through the same cache).
int64_t prefetch_box_bytes(const mdc_ctx* c) {
  const int x0 = c->bbox[0], x1 = c->bbox[2], y0 = c->bbox[1], y1 = c->bbox[3];
  return y1 >= y0 ? (int64_t)(y1 - y0 + 1) * std::min(c->rm_in_w, ((x1 + 128) & ~127) - (x0 & ~127)) : 0;
}
// Two streams (MDC_OPT_PREFETCH_STREAMS, the default): the chunks alternate between the caller's stream and a second
// one, so a chunk's tail and the next prefetch run under the other chunk's launch; a launch then holds ~5/8 of the
// workgroups the chip has room for (two of them overlap, staggered), in whole frame groups.  Measured on config 5
// (profiles/r03_experiments/10_*): 40 frames per chunk, 2 groups of 20 -- 1.76 -> 1.52 ms per 1024 frames.
int prefetch_streams(const mdc_ctx* c) { return c->opt_prefetch_streams == 1 ? 1 : 2; }
int64_t strip_resident(bool pyr) { return 256 * (pyr ? 4 : 5); }  // workgroups of kStripWaves waves the chip holds at once
int64_t chunk_groups(const mdc_ctx* c, bool pyr) {
  const int64_t nb = std::max(1, c->strip.n_blocks);
  return std::max<int64_t>(1, (strip_resident(pyr) * 5 / 8 + nb / 2) / nb);
}
int64_t prefetch_chunk_frames(const mdc_ctx* c, bool pyr = true) {
  if (c->opt_prefetch_chunk > 0) return c->opt_prefetch_chunk;
  const int64_t box = std::max<int64_t>(1, prefetch_box_bytes(c));
  if (prefetch_streams(c) == 1) return std::max<int64_t>(16, (31ll << 20) / box);
  const int64_t g = chunk_groups(c, pyr);
  const int64_t per_group = std::max<int64_t>(8, ((26ll << 20) / box + g / 2) / g);  // >= 8 frames per workgroup
  return g * per_group;
}

// UndistorterFOV::undistort<float> over float frames: LDS-tiled kernel when planned, else the gather kernel.
int enqueue_undistort_f32(mdc_ctx* c, const float* d_in, float* d_out, int64_t nframes, hipStream_t s) {
  RemapArgs a = remap_args(c, nullptr, nullptr);
  const bool aligned = (reinterpret_cast<uintptr_t>(d_in) & 15) == 0;
  if (c->plan[1].tiled && aligned && c->opt_kernel != MDC_KERNEL_GATHER) {
    const int fpb = frames_per_block(c, nframes, c->plan[1].n_blocks);
    MDC_HIP(c, launch_remap_tiled_f32(d_in, d_out, a, tile_plan(c, 1), nframes, fpb, s));
  } else {
    const int fpb = frames_per_block(c, nframes, (c->out_w * c->out_h + 255) / 256);
    MDC_HIP(c, launch_remap_gather_f32(d_in, d_out, a, nframes, fpb, s));
  }
  return MDC_OK;
}

// Enqueue the fused / photometric-only pipeline on `s`.  Lock held by caller.
// pyr (optional): levels 1..3 of the box pyramid; *pyr_done tells whether the launch wrote them.
int enqueue_process(mdc_ctx* c, const uint8_t* d_in, float* d_out, int64_t nframes, unsigned flags, hipStream_t s,
                    float* const* pyr = nullptr, bool* pyr_done = nullptr) {
  if (pyr_done) *pyr_done = false;
  bool g, v, o;
  normalise(c, flags, g, v, o);
  const float* lut = lut_for(c, g, o);
  const float* vinv = v ? c->d_vinv : nullptr;
  if (!(flags & MDC_RECTIFY)) {
    const int fw = c->in_w > 0 ? c->in_w : c->rm_in_w, fh = c->in_h > 0 ? c->in_h : c->rm_in_h;
    if (fw <= 0 || fh <= 0) return fail(c, MDC_ERR_STATE, "frame size unknown: set the photometric tables or a remap first");
    const int64_t npix = (int64_t)fw * fh;
    int fpb = frames_per_block(c, nframes, (int)((npix + 4095) / 4096), 20000);  // measured best at 8 frames per workgroup,
    if (!c->opt_fpb) fpb = std::min(fpb, 8);                                       // also for launches of thousands of frames
    MDC_HIP(c, launch_unmap(d_in, d_out, lut, vinv, npix, nframes, fpb, s));
    return MDC_OK;
  }
  if (!c->valid_remap) return fail(c, MDC_ERR_STATE, "no remap set (UndistorterFOV invalid)");
  if (vinv && (c->in_w != c->rm_in_w || c->in_h != c->rm_in_h))
    return fail(c, MDC_ERR_SIZE, "vignette is %dx%d but the remap expects %dx%d input", c->in_w, c->in_h, c->rm_in_w,
                c->rm_in_h);
  RemapArgs a = remap_args(c, lut, vinv);
  const bool aligned = (reinterpret_cast<uintptr_t>(d_in) & 15) == 0;
  if (c->strip.planned && aligned && c->opt_kernel != MDC_KERNEL_GATHER) {  // wave-private strips (scale >= ~1 remaps)
    const mdc_ctx::Strip& st = c->strip;
    const StripPlan sp{st.d_chunks, st.d_nch, st.d_taps, st.d_order, st.n_blocks, st.n_tiles, st.tiles_x, st.win_bytes, st.passes, st.nbuf,
                       c->opt_interleave != 0};
    const bool fuse_pyr = pyr && c->out_w % kStripTileW == 0;
    // Large batches go in chunks, each preceded by a linear prefetch of the NEXT chunk's source rows into the Infinity
    // Cache (launch_prefetch_rows): the strips' small window reads then hit the cache instead of interrupting the output
    // stream in HBM.  Chunk = as many frames as keep the prefetched rows (bounding box rows x frame width) within ~96 MiB.
    const int iw = c->rm_in_w;
    const int x0 = c->bbox[0], x1 = c->bbox[2], y0 = c->bbox[1], y1 = c->bbox[3];
    const int64_t frame_in = (int64_t)iw * c->rm_in_h;
    const int64_t box_bytes = prefetch_box_bytes(c);
    int64_t chunk = prefetch_chunk_frames(c, fuse_pyr);
    // Measured (profiles/r03_experiments/08_*, 10_*), ms per 1024 frames of config 5: with the fused pyramid one launch 1.86-1.94,
    // chunks on one stream 1.64-1.70, over two streams 1.47-1.52; without the levels 1.62 / 1.55-1.65 / 1.29.
    const bool prefetch = c->opt_prefetch_chunk >= 0 && (fuse_pyr || c->opt_prefetch_chunk > 0 || prefetch_streams(c) == 2) && box_bytes >= 4096 &&
                          nframes >= 2 * chunk;
    if (!prefetch) chunk = nframes;
    const size_t no = (size_t)c->out_w * c->out_h;
    const int64_t resident = strip_resident(fuse_pyr);
    auto strip = [&](int64_t f0, int64_t n, int fpb, hipStream_t t) {
      return launch_remap_strip_u8(d_in + (size_t)f0 * frame_in, d_out + (size_t)f0 * no, a, sp, n, fpb, t,
                                   fuse_pyr ? pyr[0] + (size_t)f0 * (no / 4) : nullptr, fuse_pyr ? pyr[1] + (size_t)f0 * (no / 16) : nullptr,
                                   fuse_pyr ? pyr[2] + (size_t)f0 * (no / 64) : nullptr);
    };
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &capturing) != hipSuccess) {
      (void)hipGetLastError();
      capturing = hipStreamCaptureStatusNone;
    }
    if (prefetch && prefetch_streams(c) == 2 && capturing == hipStreamCaptureStatusNone) {
      // The second stream is a slot's (with its two events), borrowed while the launches are enqueued: what is queued on it
      // stays ordered after the lease ends.  No free slot (every one held by a host call): one stream.  A caller's stream
      // that is being captured into a graph keeps everything on itself (a shared stream must not be drawn into a capture).
      SlotLease side(c, false);
      side.drained();  // a *_device call stays asynchronous: the lease only covers the enqueueing
      mdc_ctx::HostSlot* h = side.s;
      if (h && !h->ev_fork) {
        if (hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess) h->ev_fork = nullptr;
        if (h->ev_fork && hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) {
          (void)hipEventDestroy(h->ev_fork);
          h->ev_fork = nullptr;
        }
      }
      if (h && h->ev_fork) {
        const int64_t groups = chunk_groups(c, fuse_pyr);
        MDC_HIP(c, hipEventRecord(h->ev_fork, s));
        MDC_HIP(c, hipStreamWaitEvent(h->stream, h->ev_fork, 0));
        // whatever happens below, the caller's stream waits for what reached the second one
        hipError_t e = hipSuccess;
        int k = 0;
        for (int64_t f0 = 0; f0 < nframes && e == hipSuccess; f0 += chunk, k++) {
          const int64_t n = std::min<int64_t>(chunk, nframes - f0);
          hipStream_t t = (k & 1) ? h->stream : s;
          e = launch_prefetch_rows(d_in + (size_t)f0 * frame_in, frame_in, iw, x0, x1, y0, y1, n, c->d_vcal_max, t);
          const int fpb = c->opt_fpb > 0 ? (int)std::min<int64_t>(c->opt_fpb, n) : (int)((n + groups - 1) / groups);
          if (e == hipSuccess) e = strip(f0, n, fpb, t);
        }
        const hipError_t ej = hipEventRecord(h->ev_join, h->stream);
        const hipError_t ew = ej == hipSuccess ? hipStreamWaitEvent(s, h->ev_join, 0) : ej;
        MDC_HIP(c, e);
        MDC_HIP(c, ew);
        if (pyr_done) *pyr_done = fuse_pyr;
        return MDC_OK;
      }
    }
    if (prefetch) MDC_HIP(c, launch_prefetch_rows(d_in, frame_in, iw, x0, x1, y0, y1, std::min<int64_t>(chunk, nframes), c->d_vcal_max, s));
    for (int64_t f0 = 0; f0 < nframes; f0 += chunk) {
      const int64_t n = std::min<int64_t>(chunk, nframes - f0);
      if (prefetch && f0 + chunk < nframes)
        MDC_HIP(c, launch_prefetch_rows(d_in + (size_t)(f0 + chunk) * frame_in, frame_in, iw, x0, x1, y0, y1, std::min<int64_t>(chunk, nframes - f0 - chunk),
                                        c->d_vcal_max, s));
      // one round of workgroups per chunk where that is possible: a chunk is a launch of its own, its tail is not hidden
      int fpb = frames_per_block(c, n, st.n_blocks);
      if (prefetch && !c->opt_fpb) fpb = (int)std::max<int64_t>(8, (n * st.n_blocks + resident - 1) / resident);
      MDC_HIP(c, strip(f0, n, fpb, s));
    }
    if (pyr_done) *pyr_done = fuse_pyr;
    return MDC_OK;
  }
  bool use_tiled = c->plan[0].tiled && aligned;
  if (c->opt_kernel == MDC_KERNEL_GATHER) use_tiled = false;
  if (c->opt_kernel == MDC_KERNEL_TILED && !use_tiled)
    return fail(c, MDC_ERR_STATE, "tiled kernel requested but not plannable for this remap / alignment");
  if (use_tiled) {
    const TilePlan p = tile_plan(c, 0);
    int fpb = frames_per_block(c, nframes, p.n_blocks);
    // a measured pick (mdc_tune_device) belongs to this plan and to batches of the size it was measured on: small
    // launches (undistort<uchar>, the 16-frame chunks of mdc_process_frames_host) and the unmap path keep their own rule
    if (!c->opt_fpb && c->tuned_fpb > 0 && nframes >= c->tuned_min_frames) fpb = (int)std::min<int64_t>(c->tuned_fpb, nframes);
    // the fused pyramid adds level-2 hand-over rows to the workgroup's LDS: without room for them the
    // per-level passes run instead
    const bool fuse_pyr = pyr && p.tile_w * p.tile_h <= 2048 && p.tile_h % 8 == 0 && c->out_w % p.tile_w == 0 && c->out_h % p.tile_h == 0 &&
                          tiled_lds_bytes(p.win_bytes, p.nbuf, true) + tiled_pyramid_lds_bytes(p.tile_w, p.tile_h) <= kLdsPerCU;
    MDC_HIP(c, launch_remap_tiled_u8(d_in, d_out, a, p, nframes, fpb, s, fuse_pyr ? pyr[0] : nullptr,
                                     fuse_pyr ? pyr[1] : nullptr, fuse_pyr ? pyr[2] : nullptr));
    if (pyr_done) *pyr_done = fuse_pyr;
  } else {
    const int fpb = frames_per_block(c, nframes, (c->out_w * c->out_h + 255) / 256);
    MDC_HIP(c, launch_remap_gather_u8(d_in, d_out, a, nframes, fpb, s));
  }
  return MDC_OK;
}

struct BlobHeader {
  uint32_t magic, version;
  int32_t in_w, in_h, rm_in_w, rm_in_h, out_w, out_h;
  int32_t valid_gamma, valid_vignette, valid_remap, pad;
};
constexpr uint32_t kMagic = 0x4d444331u;  // "MDC1"

}  // namespace

static int set_photometric_locked(mdc_ctx* c, const float* ginv, const float* vignette_inv, int w, int h);
static int set_remap_locked(mdc_ctx* c, const float* rx, const float* ry, int in_w, int in_h, int out_w, int out_h);

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

extern "C" {

int mdc_create(int device, mdc_ctx** out) try {
  if (!out) return fail(nullptr, MDC_ERR_ARG, "mdc_create: out is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(nullptr, MDC_ERR_NO_DEVICE, "no HIP device visible (%s); this library has no CPU fallback",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (device == -1 && hipGetDevice(&device) != hipSuccess) device = 0;
  if (device < 0 || device >= n) return fail(nullptr, MDC_ERR_ARG, "device %d out of range [0,%d)", device, n);
  mdc_ctx* c = new mdc_ctx();
  c->device = device;
  DeviceGuard dg(device);
  c->h_ginv.assign(256, 0.f);
  if (const char* e = getenv("MDC_PIN_CALLER_BUFFERS")) c->opt_pin_caller = atoi(e) != 0;
  if (const char* e = getenv("MDC_ZERO_COPY")) c->opt_zero_copy = std::max(0, std::min(2, atoi(e)));  // as MDC_OPT_ZERO_COPY, for callers that cannot be recompiled
  int rc = upload_luts(c);
  if (rc == MDC_OK && hipMalloc(&c->d_vcal_max, mdc_ctx::kVcalMaxWords * sizeof(unsigned)) != hipSuccess)
    rc = fail(c, MDC_ERR_HIP, "hipMalloc of the context's scratch words failed");
  if (rc != MDC_OK) {
    g_create_err = t_err;
    if (c->d_luts) (void)hipFree(c->d_luts);
    t_err_ctx = nullptr;
    delete c;
    return rc;
  }
  *out = c;
  return MDC_OK;
} MDC_CATCH(nullptr)

void mdc_destroy(mdc_ctx* c) {
  if (!c) return;
  {
    DeviceGuard dg(c->device);
    for (mdc_ctx::HostSlot* h : c->slots) {
      if (h->stream) (void)hipStreamSynchronize(h->stream);
      if (h->d_in) (void)hipFree(h->d_in);
      if (h->d_out) (void)hipFree(h->d_out);
      if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
      if (h->ev_join) (void)hipEventDestroy(h->ev_join);
      if (h->stream) (void)hipStreamDestroy(h->stream);
      delete h;
    }
    c->slots.clear();
    for (int k = 0; k < 2; k++)
      if (c->pipe_stream[k]) (void)hipStreamSynchronize(c->pipe_stream[k]);
    if (c->pipe_up_stream) (void)hipStreamSynchronize(c->pipe_up_stream);
    unpin_all(c);
    free_plan(c);
    void* ptrs[] = {c->d_luts, c->d_vinv, c->d_rx, c->d_ry, c->d_vcal_max, c->d_pipe_in[0], c->d_pipe_in[1], c->d_pipe_out[0], c->d_pipe_out[1],
                    c->d_pipe_rec[0], c->d_pipe_rec[1], c->d_pipe_strm[0], c->d_pipe_strm[1], c->d_pipe_status[0], c->d_pipe_status[1]};
    for (void* p : ptrs)
      if (p) (void)hipFree(p);
    if (c->h_pipe_status) (void)hipHostFree(c->h_pipe_status);
    for (int k = 0; k < 2; k++) {
      if (c->pipe_done[k]) (void)hipEventDestroy(c->pipe_done[k]);
      if (c->pipe_dec[k]) (void)hipEventDestroy(c->pipe_dec[k]);
      if (c->pipe_up[k]) (void)hipEventDestroy(c->pipe_up[k]);
      if (c->pipe_huff[k]) (void)hipEventDestroy(c->pipe_huff[k]);
      if (c->pipe_stream[k]) (void)hipStreamDestroy(c->pipe_stream[k]);
    }
    if (c->pipe_up_stream) (void)hipStreamDestroy(c->pipe_up_stream);
  }
  if (t_err_ctx == c) t_err_ctx = nullptr;
  delete c;
}

const char* mdc_build_flags(void) { return mdc::build_flags_string(); }

const char* mdc_last_error(const mdc_ctx* c) {
  if (!c) return g_create_err.c_str();
  if (t_err_ctx == c) return t_err.c_str();
  std::lock_guard<std::mutex> lk(c->err_mu);
  t_err_other = c->err;
  return t_err_other.c_str();
}

int mdc_set_option(mdc_ctx* c, int option, int value) try {
  if (!c) return MDC_ERR_ARG;
  WriteLock lk(c->mu);
  switch (option) {
    case MDC_OPT_KERNEL:
      if (value < MDC_KERNEL_AUTO || value > MDC_KERNEL_TILED) return fail(c, MDC_ERR_ARG, "bad kernel selector %d", value);
      c->opt_kernel = value;
      return MDC_OK;
    case MDC_OPT_FRAMES_PER_BLOCK:
      if (value < 0) return fail(c, MDC_ERR_ARG, "bad frames-per-block %d", value);
      c->opt_fpb = value;
      return MDC_OK;
    case MDC_OPT_TILE_ROWS: {
      if (value != 0 && value != 16 && value != 32 && value != 60 && value != 64)
        return fail(c, MDC_ERR_ARG, "tile rows must be 0 (automatic), 16, 32, 60 or 64");
      if (value == c->opt_tile_h) return MDC_OK;
      c->opt_tile_h = value;
      c->tuned_fpb = 0;  // a measured frames-per-workgroup belongs to the plan it was measured on
      if (!c->valid_remap) return MDC_OK;
      DeviceGuard dg(c->device);
      MDC_HIP(c, hipDeviceSynchronize());
      return plan_tiles(c);
    }
    case MDC_OPT_TILE_COLS: {
      if (value != 0 && value != 64 && value != 128) return fail(c, MDC_ERR_ARG, "tile columns must be 0 (automatic), 64 or 128");
      if (value == c->opt_tile_w) return MDC_OK;
      c->opt_tile_w = value;
      c->tuned_fpb = 0;
      if (!c->valid_remap) return MDC_OK;
      DeviceGuard dg(c->device);
      MDC_HIP(c, hipDeviceSynchronize());
      return plan_tiles(c);
    }
    case MDC_OPT_FRAME_INTERLEAVE:
      c->opt_interleave = value != 0;
      return MDC_OK;
    case MDC_OPT_PIN_CALLER_BUFFERS: {
      c->opt_pin_caller = value != 0;
      if (!c->opt_pin_caller) {
        DeviceGuard dg(c->device);
        MDC_HIP(c, hipDeviceSynchronize());
        unpin_all(c);
      }
      return MDC_OK;
    }
    case MDC_OPT_WINDOW_BUFFERS: {
      if (value < 0 || value > 4) return fail(c, MDC_ERR_ARG, "window buffers must be 0 (auto) or 1..4 (1: strip kernel only)");
      if (value == c->opt_nbuf) return MDC_OK;
      c->opt_nbuf = value;
      if (!c->valid_remap) return MDC_OK;
      DeviceGuard dg(c->device);
      MDC_HIP(c, hipDeviceSynchronize());
      return plan_tiles(c);
    }
    case MDC_OPT_TAIL_TAPER:
      if (value < 0 || value > 2) return fail(c, MDC_ERR_ARG, "tail taper selector must be 0 (automatic), 1 (on) or 2 (off)");
      c->opt_taper = value;
      return MDC_OK;
    case MDC_OPT_ZERO_COPY:
      if (value < 0 || value > 2) return fail(c, MDC_ERR_ARG, "zero-copy selector must be 0 (automatic), 1 (on) or 2 (off)");
      c->opt_zero_copy = value;
      return MDC_OK;
    case MDC_OPT_PREFETCH_STREAMS:
      if (value < 0 || value > 2) return fail(c, MDC_ERR_ARG, "prefetch streams must be 0 (automatic), 1 or 2");
      c->opt_prefetch_streams = value;
      return MDC_OK;
    case MDC_OPT_PREFETCH_CHUNK:
      if (value < -1) return fail(c, MDC_ERR_ARG, "prefetch chunk must be -1 (off), 0 (automatic) or a frame count");
      c->opt_prefetch_chunk = value;
      return MDC_OK;
    case MDC_OPT_TWO_STAGE: {
      if (value < 0 || value > 2) return fail(c, MDC_ERR_ARG, "two-stage selector must be 0 (automatic), 1 (on) or 2 (off)");
      if (value == c->opt_two_stage) return MDC_OK;
      c->opt_two_stage = value;
      c->tuned_fpb = 0;
      if (!c->valid_remap) return MDC_OK;
      DeviceGuard dg(c->device);
      MDC_HIP(c, hipDeviceSynchronize());
      return plan_tiles(c);
    }
    case MDC_OPT_TILE_ORDER: {
      if (value < MDC_ORDER_BANDS || value > MDC_ORDER_BLOCKS2D) return fail(c, MDC_ERR_ARG, "bad tile order %d", value);
      if (value == c->opt_order) return MDC_OK;
      c->opt_order = value;
      if (!c->valid_remap) return MDC_OK;
      DeviceGuard dg(c->device);
      MDC_HIP(c, hipDeviceSynchronize());
      return plan_tiles(c);
    }
  }
  return fail(c, MDC_ERR_ARG, "unknown option %d", option);
} MDC_CATCH(c)

int mdc_get_info(mdc_ctx* c, mdc_info* i) try {
  if (!c || !i) return MDC_ERR_ARG;
  ReadLock lk(c->mu);
  memset(i, 0, sizeof *i);
  i->device = c->device;
  i->in_w = c->in_w ? c->in_w : c->rm_in_w;
  i->in_h = c->in_h ? c->in_h : c->rm_in_h;
  i->out_w = c->valid_remap ? c->out_w : 0;
  i->out_h = c->valid_remap ? c->out_h : 0;
  i->valid_gamma = c->valid_gamma;
  i->valid_vignette = c->valid_vignette;
  i->valid_remap = c->valid_remap;
  const mdc_ctx::SrcPlan& p0 = c->plan[0];
  i->tiled = c->valid_remap && p0.tiled;
  i->tile_w = p0.tiled ? p0.tile_w : c->opt_tile_w;
  i->tile_h = p0.tiled ? p0.tile_h : c->opt_tile_h;
  i->n_tiles = p0.n_tiles;
  i->lds_bytes = p0.tiled ? (int)tiled_lds_bytes(p0.win_bytes, p0.nbuf, true) : 0;
  i->two_stage = c->strip.planned ? 1 : 0;
  i->prefetch_chunk = (c->strip.planned && c->opt_prefetch_chunk >= 0 && prefetch_box_bytes(c) >= 4096) ? (int)prefetch_chunk_frames(c) : 0;
  i->prefetch_streams = i->prefetch_chunk ? prefetch_streams(c) : 0;
  if (c->strip.planned) {
    i->tiled = c->valid_remap;
    i->tile_w = kStripTileW;
    i->tile_h = kStripTileH;
    i->n_tiles = c->strip.n_tiles;
    i->lds_bytes = (int)strip_lds_bytes(c->strip.win_bytes, c->strip.nbuf, kStripWaves);
    i->window_buffers = c->strip.nbuf;
  }
  i->window_buffers = p0.tiled ? p0.nbuf : 0;
  i->f32_tiled = c->valid_remap && c->plan[1].tiled;
  i->f32_tile_w = c->plan[1].tiled ? c->plan[1].tile_w : 0;
  i->f32_tile_h = c->plan[1].tiled ? c->plan[1].tile_h : 0;
  for (int k = 0; k < 4; k++) i->src_bbox[k] = c->bbox[k];
  i->src_bbox_bytes = c->bbox[2] >= 0 ? (int64_t)(c->bbox[2] - c->bbox[0] + 1) * (c->bbox[3] - c->bbox[1] + 1) : 0;
  i->src_staged_bytes = c->strip.planned ? c->strip.staged_bytes : p0.staged_bytes;
  i->n_black = c->n_black;
  return MDC_OK;
} MDC_CATCH(c)

int mdc_set_photometric(mdc_ctx* c, const float* ginv, const float* vignette_inv, int w, int h) try {
  if (!c) return MDC_ERR_ARG;
  if (w <= 0 || h <= 0) return fail(c, MDC_ERR_ARG, "bad frame size %dx%d", w, h);
  WriteLock lk(c->mu);
  return set_photometric_locked(c, ginv, vignette_inv, w, h);
} MDC_CATCH(c)

// (lock held) The tables are replaced in place: every kernel that may still read them -- also those the
// *_device entry points put on caller streams -- has to be done first, hence the device-wide wait.
static int set_photometric_locked(mdc_ctx* c, const float* ginv, const float* vignette_inv, int w, int h) {
  DeviceGuard dg(c->device);
  MDC_HIP(c, hipDeviceSynchronize());
  c->in_w = w;
  c->in_h = h;
  c->valid_gamma = ginv != nullptr;
  if (ginv) c->h_ginv.assign(ginv, ginv + 256);
  int rc = upload_luts(c);
  if (rc != MDC_OK) return rc;
  c->valid_vignette = vignette_inv != nullptr;
  if (c->d_vinv) {
    (void)hipFree(c->d_vinv);
    c->d_vinv = nullptr;
  }
  c->h_vinv.clear();
  if (vignette_inv) {
    const size_t n = (size_t)w * h;
    c->h_vinv.assign(vignette_inv, vignette_inv + n);
    MDC_HIP(c, hipMalloc(&c->d_vinv, n * sizeof(float)));
    MDC_HIP(c, hipMemcpy(c->d_vinv, vignette_inv, n * sizeof(float), hipMemcpyHostToDevice));
  }
  return MDC_OK;
}

int mdc_set_remap(mdc_ctx* c, const float* rx, const float* ry, int in_w, int in_h, int out_w, int out_h) try {
  if (!c) return MDC_ERR_ARG;
  WriteLock lk(c->mu);
  return set_remap_locked(c, rx, ry, in_w, in_h, out_w, out_h);
} MDC_CATCH(c)

static int set_remap_locked(mdc_ctx* c, const float* rx, const float* ry, int in_w, int in_h, int out_w, int out_h) {
  DeviceGuard dg(c->device);
  MDC_HIP(c, hipDeviceSynchronize());
  c->valid_remap = false;
  c->tuned_fpb = 0;
  free_plan(c);
  for (float** p : {&c->d_rx, &c->d_ry})
    if (*p) {
      (void)hipFree(*p);
      *p = nullptr;
    }
  c->h_rx.clear();
  c->h_ry.clear();
  if (!rx || !ry) return MDC_OK;
  if (in_w <= 1 || in_h <= 1 || out_w <= 0 || out_h <= 0)
    return fail(c, MDC_ERR_ARG, "bad remap geometry %dx%d -> %dx%d", in_w, in_h, out_w, out_h);
  const size_t n = (size_t)out_w * out_h;
  // Defensive validation of what UndistorterFOV's constructor guarantees
  // (src/FOVUndistorter.cpp:243): every non-black tap lies strictly inside the frame.
  for (size_t i = 0; i < n; i++) {
    if (rx[i] < 0) continue;
    if (!(rx[i] > 0 && ry[i] > 0 && rx[i] < in_w - 1 && ry[i] < in_h - 1))
      return fail(c, MDC_ERR_ARG, "remap entry %zu = (%g,%g) is outside (0,%d)x(0,%d)", i, rx[i], ry[i], in_w - 1,
                  in_h - 1);
  }
  c->h_rx.assign(rx, rx + n);
  c->h_ry.assign(ry, ry + n);
  c->rm_in_w = in_w;
  c->rm_in_h = in_h;
  c->out_w = out_w;
  c->out_h = out_h;
  MDC_HIP(c, hipMalloc(&c->d_rx, n * sizeof(float)));
  MDC_HIP(c, hipMalloc(&c->d_ry, n * sizeof(float)));
  MDC_HIP(c, hipMemcpy(c->d_rx, rx, n * sizeof(float), hipMemcpyHostToDevice));
  MDC_HIP(c, hipMemcpy(c->d_ry, ry, n * sizeof(float), hipMemcpyHostToDevice));
  int rc = plan_tiles(c);
  if (rc != MDC_OK) return rc;
  c->valid_remap = true;
  return MDC_OK;
}

int mdc_unmap_batch_device(mdc_ctx* c, const uint8_t* d_in, float* d_out, int64_t nframes, unsigned flags, void* stream) try {
  if (!c) return MDC_ERR_ARG;
  if (!d_in || !d_out || nframes < 0) return fail(c, MDC_ERR_ARG, "mdc_unmap_batch_device: bad argument");
  ReadLock lk(c->mu);
  DeviceGuard dg(c->device);
  return enqueue_process(c, d_in, d_out, nframes, flags & ~MDC_RECTIFY, (hipStream_t)stream);
} MDC_CATCH(c)

int mdc_process_batch_device(mdc_ctx* c, const uint8_t* d_in, float* d_out, int64_t nframes, unsigned flags, void* stream) try {
  if (!c) return MDC_ERR_ARG;
  if (!d_in || !d_out || nframes < 0) return fail(c, MDC_ERR_ARG, "mdc_process_batch_device: bad argument");
  ReadLock lk(c->mu);
  DeviceGuard dg(c->device);
  return enqueue_process(c, d_in, d_out, nframes, flags, (hipStream_t)stream);
} MDC_CATCH(c)

int mdc_undistort_batch_device_f32(mdc_ctx* c, const float* d_in, float* d_out, int64_t nframes, void* stream) try {
  if (!c) return MDC_ERR_ARG;
  if (!d_in || !d_out || nframes < 0) return fail(c, MDC_ERR_ARG, "mdc_undistort_batch_device_f32: bad argument");
  ReadLock lk(c->mu);
  DeviceGuard dg(c->device);
  if (!c->valid_remap) return fail(c, MDC_ERR_STATE, "no remap set (UndistorterFOV invalid)");
  return enqueue_undistort_f32(c, d_in, d_out, nframes, (hipStream_t)stream);
} MDC_CATCH(c)

int mdc_pyramid_batch_device(mdc_ctx* c, const float* d_base, int w, int h, int levels, float* const* d_levels,
                             int64_t nframes, void* stream) try {
  if (!c) return MDC_ERR_ARG;
  if (!d_base || w <= 0 || h <= 0 || levels < 1 || nframes < 0 || (levels > 1 && !d_levels))
    return fail(c, MDC_ERR_ARG, "mdc_pyramid_batch_device: bad argument");
  ReadLock lk(c->mu);
  DeviceGuard dg(c->device);
  hipStream_t s = (hipStream_t)stream;
  const float* src = d_base;
  for (int l = 1; l < levels; l++) {
    if (!d_levels[l - 1]) return fail(c, MDC_ERR_ARG, "level %d buffer is NULL", l);
    MDC_HIP(c, launch_pyramid_level(src, d_levels[l - 1], w >> (l - 1), h >> (l - 1), nframes, s));
    src = d_levels[l - 1];
  }
  return MDC_OK;
} MDC_CATCH(c)

int mdc_process_pyramid_batch_device(mdc_ctx* c, const uint8_t* d_in, float* d_base, int levels, float* const* d_levels,
                                     int64_t nframes, unsigned flags, void* stream) try {
  if (!c) return MDC_ERR_ARG;
  if (!d_in || !d_base || nframes < 0 || levels < 1 || (levels > 1 && !d_levels))
    return fail(c, MDC_ERR_ARG, "mdc_process_pyramid_batch_device: bad argument");
  for (int l = 1; l < levels; l++)
    if (!d_levels[l - 1]) return fail(c, MDC_ERR_ARG, "level %d buffer is NULL", l);
  ReadLock lk(c->mu);
  DeviceGuard dg(c->device);
  hipStream_t s = (hipStream_t)stream;
  const bool rect = (flags & MDC_RECTIFY) != 0;
  const int w = rect ? c->out_w : (c->in_w > 0 ? c->in_w : c->rm_in_w), h = rect ? c->out_h : (c->in_h > 0 ? c->in_h : c->rm_in_h);
  float* pyr[3] = {levels > 1 ? d_levels[0] : nullptr, levels > 2 ? d_levels[1] : nullptr, levels > 3 ? d_levels[2] : nullptr};
  bool fused = false;
  int rc = enqueue_process(c, d_in, d_base, nframes, flags, s, levels > 1 ? pyr : nullptr, &fused);
  if (rc != MDC_OK) return rc;
  // levels the launch did not write (no fused path for this geometry, or more than 4 levels)
  const int first = fused ? std::min(levels, 4) : 1;
  const float* src = first == 1 ? d_base : d_levels[first - 2];
  for (int l = first; l < levels; l++) {
    MDC_HIP(c, launch_pyramid_level(src, d_levels[l - 1], w >> (l - 1), h >> (l - 1), nframes, s));
    src = d_levels[l - 1];
  }
  return MDC_OK;
} MDC_CATCH(c)

// mdc_fov_model -> pixel-unit lens model, operation for operation as src/FOVUndistorter.cpp:289-301
// (float products, `- 0.5` in double for the input camera, `- 0.5f`-equivalent narrowing for the output one,
// double tan narrowed to float -- see DESIGN.md section 2).
static DistortModel distort_model(const mdc_fov_model* f) {
  return make_distort_model(f->in_calib, f->in_w, f->in_h, f->out_calib, f->out_w, f->out_h);
}

int mdc_distort_points_device(mdc_ctx* c, const mdc_fov_model* model, float* d_x, float* d_y, int64_t n, void* stream) try {
  if (!c) return MDC_ERR_ARG;
  if (!model || n < 0 || (n > 0 && (!d_x || !d_y))) return fail(c, MDC_ERR_ARG, "mdc_distort_points_device: bad argument");
  ReadLock lk(c->mu);
  DeviceGuard dg(c->device);
  MDC_HIP(c, launch_distort_points(d_x, d_y, n, distort_model(model), (hipStream_t)stream));
  return MDC_OK;
} MDC_CATCH(c)

int mdc_distort_points_host(mdc_ctx* c, const mdc_fov_model* model, float* x, float* y, int64_t n) try {
  if (!c) return MDC_ERR_ARG;
  if (!model || n < 0 || (n > 0 && (!x || !y))) return fail(c, MDC_ERR_ARG, "mdc_distort_points_host: bad argument");
  if (n == 0) return MDC_OK;
  ReadLock lk(c->mu);
  DeviceGuard dg(c->device);
  const size_t bytes = (size_t)n * sizeof(float);
  SlotLease slot(c);
  if (!slot.s) return MDC_ERR_HIP;
  int rc = ensure_stage(c, slot.s, bytes, bytes);
  if (rc != MDC_OK) return rc;
  hipStream_t st = slot.s->stream;
  float* dx = (float*)slot.s->d_in;
  float* dy = slot.s->d_out;
  MDC_HIP(c, hipMemcpyAsync(dx, x, bytes, hipMemcpyHostToDevice, st));
  MDC_HIP(c, hipMemcpyAsync(dy, y, bytes, hipMemcpyHostToDevice, st));
  MDC_HIP(c, launch_distort_points(dx, dy, n, distort_model(model), st));
  MDC_HIP(c, hipMemcpyAsync(x, dx, bytes, hipMemcpyDeviceToHost, st));
  MDC_HIP(c, hipMemcpyAsync(y, dy, bytes, hipMemcpyDeviceToHost, st));
  MDC_HIP(c, hipStreamSynchronize(st));
  slot.drained();
  return MDC_OK;
} MDC_CATCH(c)

int mdc_gradients_batch_device(mdc_ctx* c, const float* d_level, int w, int h, float* d_dI, float* d_abs_squared_grad,
                               int64_t nframes, void* stream) try {
  if (!c) return MDC_ERR_ARG;
  if (!d_level || !d_dI || !d_abs_squared_grad || w < 1 || h < 1 || nframes < 0 || (int64_t)w * h >= (1ll << 31))
    return fail(c, MDC_ERR_ARG, "mdc_gradients_batch_device: bad argument");
  ReadLock lk(c->mu);
  DeviceGuard dg(c->device);
  MDC_HIP(c, launch_gradients(d_level, d_dI, d_abs_squared_grad, w, h, nframes, (hipStream_t)stream));
  return MDC_OK;
} MDC_CATCH(c)

int mdc_process_pyramid_gradients_batch_device(mdc_ctx* c, const uint8_t* d_in, float* d_base, int levels, float* const* d_levels,
                                               float* const* d_dI, float* const* d_abs_squared_grad, int64_t nframes, unsigned flags,
                                               int chunk_frames, void* stream) try {
  if (!c) return MDC_ERR_ARG;
  if (!d_in || !d_base || nframes < 0 || levels < 1 || levels > 8 || (levels > 1 && !d_levels) || !d_dI || !d_abs_squared_grad || chunk_frames < 0)
    return fail(c, MDC_ERR_ARG, "mdc_process_pyramid_gradients_batch_device: bad argument");
  for (int l = 0; l < levels; l++)
    if ((l && !d_levels[l - 1]) || !d_dI[l] || !d_abs_squared_grad[l]) return fail(c, MDC_ERR_ARG, "level %d has a NULL buffer", l);
  ReadLock lk(c->mu);
  DeviceGuard dg(c->device);
  hipStream_t s = (hipStream_t)stream;
  const bool rect = (flags & MDC_RECTIFY) != 0;
  if (rect && !c->valid_remap) return fail(c, MDC_ERR_STATE, "no remap set (UndistorterFOV invalid)");
  const int w0 = rect ? c->out_w : (c->in_w > 0 ? c->in_w : c->rm_in_w), h0 = rect ? c->out_h : (c->in_h > 0 ? c->in_h : c->rm_in_h);
  const int iw = (rect || c->in_w <= 0) ? c->rm_in_w : c->in_w, ih = (rect || c->in_h <= 0) ? c->rm_in_h : c->in_h;
  if (w0 <= 0 || h0 <= 0 || iw <= 0 || ih <= 0) return fail(c, MDC_ERR_STATE, "frame size unknown");
  int lw[8], lh[8];
  size_t level_bytes = 0;
  for (int l = 0; l < levels; l++) {
    lw[l] = w0 >> l;
    lh[l] = h0 >> l;
    if (lw[l] < 1 || lh[l] < 1) return fail(c, MDC_ERR_ARG, "level %d of a %dx%d image is empty", l, w0, h0);
    level_bytes += (size_t)lw[l] * lh[l] * sizeof(float);
  }
  // Frames per chunk.  Measured (tools/dso_rate.py, 1280 x 1024, 512 frames, profiles/r03_dso_rate.txt): 8 frames per chunk
  // 5.0 ms, 24: 4.2, 96: 3.85-3.89, separate launches over the whole batch: 3.87-3.95 -- the levels are NOT read back from the
  // Infinity Cache (the remap's stores are nontemporal: plain ones evict its prefetched source rows and measured slower in
  // total, experiment 11), the whole path runs at what the memory system gives 34.8 MB of writes + 7.6 MB of reads per
  // frame; chunks exist to bound the launch sizes, and below ~100 frames they lose to their tails.
  int64_t chunk = chunk_frames > 0 ? chunk_frames : std::max<int64_t>(1, (int64_t)((640u << 20) / level_bytes));
  {  // a gradient launch holds a chunk's workgroups of up to four levels: fewer than 2^31 (128 x 8 pixels each)
    int64_t wgs = 0;
    for (int l = 0; l < std::min(levels, 4); l++) wgs += (int64_t)((lw[l] + 127) / 128) * ((lh[l] + 7) / 8);
    chunk = std::max<int64_t>(1, std::min<int64_t>(chunk, ((1ll << 31) - 1) / std::max<int64_t>(1, wgs)));
  }
  const size_t npi = (size_t)iw * ih;
  for (int64_t f0 = 0; f0 < nframes; f0 += chunk) {
    const int64_t n = std::min<int64_t>(chunk, nframes - f0);
    float* lv[8];
    for (int l = 1; l < levels; l++) lv[l - 1] = d_levels[l - 1] + (size_t)f0 * lw[l] * lh[l];
    float* base = d_base + (size_t)f0 * w0 * h0;
    float* pyr[3] = {levels > 1 ? lv[0] : nullptr, levels > 2 ? lv[1] : nullptr, levels > 3 ? lv[2] : nullptr};
    bool fused = false;
    int rc = enqueue_process(c, d_in + (size_t)f0 * npi, base, n, flags, s, levels > 1 ? pyr : nullptr, &fused);
    if (rc != MDC_OK) return rc;
    const int first = fused ? std::min(levels, 4) : 1;
    const float* src = first == 1 ? base : lv[first - 2];
    for (int l = first; l < levels; l++) {
      MDC_HIP(c, launch_pyramid_level(src, lv[l - 1], lw[l - 1], lh[l - 1], n, s));
      src = lv[l - 1];
    }
    for (int l0 = 0; l0 < levels; l0 += 4) {  // gradients: four levels per launch
      const int nl = std::min(4, levels - l0);
      const float* gs[4];
      float *gd[4], *ga[4];
      for (int k = 0; k < nl; k++) {
        const int l = l0 + k;
        gs[k] = l == 0 ? base : lv[l - 1];
        gd[k] = d_dI[l] + (size_t)f0 * lw[l] * lh[l] * 3;
        ga[k] = d_abs_squared_grad[l] + (size_t)f0 * lw[l] * lh[l];
      }
      MDC_HIP(c, launch_gradients_levels(nl, gs, gd, ga, lw + l0, lh + l0, n, s));
    }
  }
  return MDC_OK;
} MDC_CATCH(c)

int mdc_vcal_plane_step_device(mdc_ctx* c, const float* d_images, const float* d_p2x, const float* d_p2y, int n_images, int w, int h,
                               int n_plane, float* d_plane_color, const float* d_vignette_factor, int oth2, float* d_ff,
                               float* d_fc, double* d_er, void* stream) try {
  if (!c) return MDC_ERR_ARG;
  if (!d_images || !d_p2x || !d_p2y || !d_plane_color || !d_vignette_factor || !d_ff || !d_fc || !d_er || n_images < 0 || w < 2 ||
      h < 2 || n_plane < 0)
    return fail(c, MDC_ERR_ARG, "mdc_vcal_plane_step_device: bad argument");
  ReadLock lk(c->mu);
  DeviceGuard dg(c->device);
  MDC_HIP(c, launch_vcal_plane_step(d_images, d_p2x, d_p2y, n_images, w, h, n_plane, d_plane_color, d_vignette_factor, oth2, d_ff,
                                    d_fc, d_er, (hipStream_t)stream));
  return MDC_OK;
} MDC_CATCH(c)

int mdc_vcal_vignette_step_device(mdc_ctx* c, const float* d_images, const float* d_p2x, const float* d_p2y, int n_images, int w,
                                  int h, int n_plane, const float* d_plane_color, float* d_vignette_factor, int oth2, float* d_tt,
                                  float* d_ct, double* d_er, void* stream) try {
  if (!c) return MDC_ERR_ARG;
  if (!d_images || !d_p2x || !d_p2y || !d_plane_color || !d_vignette_factor || !d_tt || !d_ct || !d_er || n_images < 0 || w < 2 ||
      h < 2 || n_plane < 0)
    return fail(c, MDC_ERR_ARG, "mdc_vcal_vignette_step_device: bad argument");
  ReadLock lk(c->mu);
  DeviceGuard dg(c->device);
  MDC_HIP(c, launch_vcal_vignette_step(d_images, d_p2x, d_p2y, n_images, w, h, n_plane, d_plane_color, d_vignette_factor, oth2,
                                       d_tt, d_ct, d_er, c->d_vcal_max + (c->vcal_max_next++ % mdc_ctx::kVcalMaxWords), (hipStream_t)stream));
  return MDC_OK;
} MDC_CATCH(c)

struct mdc_vcal_index {
  mdc::VcalIndex* ix;
  int device;
};

int mdc_vcal_index_create(mdc_ctx* c, const float* d_images, const float* d_p2x, const float* d_p2y, int n_images, int w, int h,
                          int n_plane, void* stream, mdc_vcal_index** out) try {
  if (!c) return MDC_ERR_ARG;
  if (!out) return fail(c, MDC_ERR_ARG, "mdc_vcal_index_create: out is NULL");
  *out = nullptr;
  if (!d_images || !d_p2x || !d_p2y || n_images < 0 || n_images > 65535 || w < 2 || h < 2 || n_plane < 0 || n_plane >= (1 << 30) ||
      (long long)w * h >= (1ll << 31))
    return fail(c, MDC_ERR_ARG, "mdc_vcal_index_create: bad argument");
  ReadLock lk(c->mu);
  DeviceGuard dg(c->device);
  mdc::VcalIndex* ix = nullptr;
  MDC_HIP(c, mdc::vcal_index_build(d_images, d_p2x, d_p2y, n_images, w, h, n_plane, (hipStream_t)stream, &ix));
  *out = new mdc_vcal_index{ix, c->device};
  return MDC_OK;
} MDC_CATCH(c)

void mdc_vcal_index_destroy(mdc_vcal_index* index) {
  if (!index) return;
  DeviceGuard dg(index->device);
  mdc::vcal_index_free(index->ix);
  delete index;
}

int64_t mdc_vcal_index_bytes(const mdc_vcal_index* index) { return index ? mdc::vcal_index_bytes(index->ix) : 0; }
int64_t mdc_vcal_index_entries(const mdc_vcal_index* index) { return index ? mdc::vcal_index_entries(index->ix) : 0; }

int mdc_vcal_vignette_step_indexed_device(mdc_ctx* c, const mdc_vcal_index* index, const float* d_plane_color,
                                          float* d_vignette_factor, int oth2, float* d_tt, float* d_ct, double* d_er, void* stream) try {
  if (!c) return MDC_ERR_ARG;
  if (!index || !d_plane_color || !d_vignette_factor || !d_tt || !d_ct || !d_er)
    return fail(c, MDC_ERR_ARG, "mdc_vcal_vignette_step_indexed_device: bad argument");
  if (index->device != c->device) return fail(c, MDC_ERR_ARG, "mdc_vcal_vignette_step_indexed_device: index built on another device");
  ReadLock lk(c->mu);
  DeviceGuard dg(c->device);
  MDC_HIP(c, mdc::launch_vcal_vignette_step_indexed(index->ix, d_plane_color, d_vignette_factor, oth2, d_tt, d_ct, d_er,
                                                    c->d_vcal_max + (c->vcal_max_next++ % mdc_ctx::kVcalMaxWords), (hipStream_t)stream));
  return MDC_OK;
} MDC_CATCH(c)

int mdc_vcal_scale_images_device(mdc_ctx* c, float* d_images, int n_images, int64_t npix, float mean_exposure,
                                 const float* d_exposure_times, void* stream) try {
  if (!c) return MDC_ERR_ARG;
  if (n_images < 0 || n_images > 65535 || npix < 0 || (n_images > 0 && npix > 0 && (!d_images || !d_exposure_times)))
    return fail(c, MDC_ERR_ARG, "mdc_vcal_scale_images_device: bad argument");
  ReadLock lk(c->mu);
  DeviceGuard dg(c->device);
  MDC_HIP(c, launch_vcal_scale_images(d_images, n_images, npix, mean_exposure, d_exposure_times, (hipStream_t)stream));
  return MDC_OK;
} MDC_CATCH(c)

int mdc_vcal_gradient_mask_device(mdc_ctx* c, float* d_images, int n_images, int w, int h, int max_abs_grad, void* stream) try {
  if (!c) return MDC_ERR_ARG;
  if (n_images < 0 || (n_images > 0 && !d_images) || w < 1 || h < 1 || (long long)w * h >= (1ll << 31) || max_abs_grad < 0)
    return fail(c, MDC_ERR_ARG, "mdc_vcal_gradient_mask_device: bad argument");
  ReadLock lk(c->mu);
  DeviceGuard dg(c->device);
  MDC_HIP(c, launch_vcal_gradient_mask(d_images, n_images, w, h, max_abs_grad, (hipStream_t)stream));
  return MDC_OK;
} MDC_CATCH(c)

int mdc_vcal_mask_coords_device(mdc_ctx* c, float* d_x, float* d_y, int64_t n, int w, int h, void* stream) try {
  if (!c) return MDC_ERR_ARG;
  if (n < 0 || (n > 0 && (!d_x || !d_y)) || w < 1 || h < 1) return fail(c, MDC_ERR_ARG, "mdc_vcal_mask_coords_device: bad argument");
  ReadLock lk(c->mu);
  DeviceGuard dg(c->device);
  MDC_HIP(c, launch_vcal_mask_coords(d_x, d_y, n, w, h, (hipStream_t)stream));
  return MDC_OK;
} MDC_CATCH(c)

int mdc_vcal_smooth_device(mdc_ctx* c, const float* d_vignette_factor, int w, int h, float* d_smoothed, float* d_scratch,
                           void* stream) try {
  if (!c) return MDC_ERR_ARG;
  if (!d_vignette_factor || !d_smoothed || !d_scratch || w < 1 || h < 1 || (long long)w * h >= (1ll << 31) ||
      d_smoothed == d_scratch || d_vignette_factor == d_scratch)
    return fail(c, MDC_ERR_ARG, "mdc_vcal_smooth_device: bad argument");
  ReadLock lk(c->mu);
  DeviceGuard dg(c->device);
  MDC_HIP(c, launch_vcal_smooth(d_vignette_factor, w, h, d_smoothed, d_scratch, (hipStream_t)stream));
  return MDC_OK;
} MDC_CATCH(c)

int mdc_vcal_solve_device(mdc_ctx* c, const float* d_images, const float* d_p2x, const float* d_p2y, int n_images, int w, int h,
                          int n_plane, float* d_plane_color, float* d_vignette_factor, int max_iterations, int outlier_th,
                          double* er_out, void* stream) try {
  if (!c) return MDC_ERR_ARG;
  if (!d_plane_color || !d_vignette_factor || max_iterations < 0 || outlier_th < 0 || outlier_th > 46340)
    return fail(c, MDC_ERR_ARG, "mdc_vcal_solve_device: bad argument");
  if (max_iterations == 0) return MDC_OK;
  mdc_vcal_index* index = nullptr;
  int rc = mdc_vcal_index_create(c, d_images, d_p2x, d_p2y, n_images, w, h, n_plane, stream, &index);
  if (rc != MDC_OK) return rc;
  ReadLock lk(c->mu);
  DeviceGuard dg(c->device);
  hipStream_t s = (hipStream_t)stream;
  float *d_ff = nullptr, *d_fc = nullptr, *d_tt = nullptr, *d_ct = nullptr;
  double* d_er = nullptr;
  const size_t plane_bytes = (size_t)std::max(n_plane, 1) * sizeof(float), img_bytes = (size_t)w * h * sizeof(float);
  const size_t er_bytes = (size_t)max_iterations * 4 * sizeof(double);
  auto done = [&](int code) {
    (void)hipStreamSynchronize(s);
    for (void* p : {(void*)d_ff, (void*)d_fc, (void*)d_tt, (void*)d_ct, (void*)d_er}) (void)hipFree(p);
    mdc_vcal_index_destroy(index);
    return code;
  };
#define MDC_SOLVE(call_)                                                  \
  do {                                                                    \
    hipError_t e_ = (call_);                                              \
    if (e_ != hipSuccess) return done(fail(c, MDC_ERR_HIP, "%s: %s", #call_, hipGetErrorString(e_))); \
  } while (0)
  MDC_SOLVE(hipMalloc(&d_ff, plane_bytes));
  MDC_SOLVE(hipMalloc(&d_fc, plane_bytes));
  MDC_SOLVE(hipMalloc(&d_tt, img_bytes));
  MDC_SOLVE(hipMalloc(&d_ct, img_bytes));
  MDC_SOLVE(hipMalloc(&d_er, er_bytes));
  for (int it = 0; it < max_iterations; it++) {
    const int oth2 = it < max_iterations / 2 ? 10000 * 10000 : outlier_th * outlier_th;  // :397-398
    MDC_SOLVE(launch_vcal_plane_step(d_images, d_p2x, d_p2y, n_images, w, h, n_plane, d_plane_color, d_vignette_factor, oth2, d_ff,
                                     d_fc, d_er + 4 * it, s));
    MDC_SOLVE(mdc::launch_vcal_vignette_step_indexed(index->ix, d_plane_color, d_vignette_factor, oth2, d_tt, d_ct, d_er + 4 * it + 2,
                                                     c->d_vcal_max + (c->vcal_max_next++ % mdc_ctx::kVcalMaxWords), s));
  }
  if (er_out) MDC_SOLVE(hipMemcpyAsync(er_out, d_er, er_bytes, hipMemcpyDeviceToHost, s));
#undef MDC_SOLVE
  return done(MDC_OK);
} MDC_CATCH(c)

// Plan selection by measurement (as FFT / BLAS libraries do): which tile shape and workgroup length is fastest depends on
// the remap (window sizes) and, by a few per cent, on the individual GPU.  Runs the fused pass over the caller's
// batch with every candidate, keeps the fastest as the context's plan.
int mdc_tune_device(mdc_ctx* c, const uint8_t* d_in, float* d_out, int64_t nframes, unsigned flags, void* stream,
                    mdc_tune_result* result) try {
  if (!c) return MDC_ERR_ARG;
  if (!d_in || !d_out || nframes <= 0) return fail(c, MDC_ERR_ARG, "mdc_tune_device: bad argument");
  WriteLock lk(c->mu);
  DeviceGuard dg(c->device);
  if (!(flags & MDC_RECTIFY) || !c->valid_remap) return fail(c, MDC_ERR_STATE, "mdc_tune_device: needs a remap and MDC_RECTIFY");
  hipStream_t s = (hipStream_t)stream;
  static const TileShape shapes[] = {{128, 16}, {64, 32}, {128, 32}};
  static const int fpbs[] = {32, 64, 96, 128};  // (with the tapered tail the longer workgroups pay: 96 measured 1.8 % ahead of 64)
  hipEvent_t e0, e1;
  MDC_HIP(c, hipEventCreate(&e0));
  MDC_HIP(c, hipEventCreate(&e1));
  float best = 1e30f;
  int bw = 0, bh = 0, bf = 0, tried = 0;
  int rc = MDC_OK;
  const int caller_fpb = c->opt_fpb;  // the caller's own override is restored afterwards: the pick goes into tuned_fpb
  for (const TileShape& sh : shapes) {
    c->opt_tile_w = sh.w;
    c->opt_tile_h = sh.h;
    // re-planning frees the plan tables: like every setter that re-plans, wait for the WHOLE device -- kernels that
    // other threads put on other streams through the *_device entry points may still be reading them
    if (hipDeviceSynchronize() != hipSuccess) {
      rc = fail(c, MDC_ERR_HIP, "mdc_tune_device: hipDeviceSynchronize failed");
      break;
    }
    if ((rc = plan_tiles(c)) != MDC_OK) break;
    if (!c->plan[0].tiled) continue;
    for (int fpb : fpbs) {
      c->opt_fpb = fpb;
      float ms[5];
      bool ok = true;
      for (int k = 0; k < 7 && ok; k++) {  // 2 warm-up launches, 5 timed
        if (k >= 2) ok = hipEventRecord(e0, s) == hipSuccess;
        ok = ok && enqueue_process(c, d_in, d_out, nframes, flags, s) == MDC_OK;
        if (k >= 2) ok = ok && hipEventRecord(e1, s) == hipSuccess && hipEventSynchronize(e1) == hipSuccess &&
                         hipEventElapsedTime(&ms[k - 2], e0, e1) == hipSuccess;
      }
      if (!ok) continue;
      std::sort(ms, ms + 5);
      tried++;
      if (ms[2] < best) {
        best = ms[2];
        bw = sh.w;
        bh = sh.h;
        bf = fpb;
      }
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipDeviceSynchronize();
  // the winner (or, if nothing could be timed, the automatic choice) becomes the plan
  c->opt_tile_w = bw;
  c->opt_tile_h = bh;
  c->opt_fpb = caller_fpb;
  c->tuned_fpb = bf;
  c->tuned_min_frames = bf ? std::max<int64_t>(nframes / 4, (int64_t)bf * 8) : 0;
  const int rc2 = plan_tiles(c);
  if (rc == MDC_OK) rc = rc2;
  if (result) {
    result->tile_w = bw;
    result->tile_h = bh;
    result->frames_per_block = bf;
    result->ms = tried ? best : 0.f;
    result->candidates = tried;
  }
  return rc;
} MDC_CATCH(c)

int mdc_describe_launch(mdc_ctx* c, unsigned flags, int pyramid_levels, char* buf, size_t cap) try {
  if (!c || !buf || cap == 0) return MDC_ERR_ARG;
  ReadLock lk(c->mu);
  bool g, v, o;
  normalise(c, flags, g, v, o);
  char tmp[160];
  if (!(flags & MDC_RECTIFY)) {
    const int fw = c->in_w > 0 ? c->in_w : c->rm_in_w, fh = c->in_h > 0 ? c->in_h : c->rm_in_h;
    snprintf(tmp, sizeof tmp, "%s<%s>", ((int64_t)fw * fh) % 4 == 0 ? "unmap_xpose_kernel" : "unmap_scalar_kernel", v ? "true" : "false");
  } else if (!c->valid_remap) {
    return fail(c, MDC_ERR_STATE, "no remap set (UndistorterFOV invalid)");
  } else if (c->strip.planned && c->opt_kernel != MDC_KERNEL_GATHER) {
    const bool pyr = pyramid_levels > 1 && c->out_w % kStripTileW == 0;
    snprintf(tmp, sizeof tmp, "remap_strip_kernel<%s, %s, %d, %d, %d>", v ? "true" : "false", pyr ? "true" : "false", c->strip.nbuf,
             c->strip.passes, kStripWaves);
  } else if (c->plan[0].tiled && c->opt_kernel != MDC_KERNEL_GATHER) {
    const mdc_ctx::SrcPlan& p = c->plan[0];
    const bool pyr = pyramid_levels > 1 && p.tile_w * p.tile_h <= 2048 && p.tile_h % 8 == 0 && c->out_w % p.tile_w == 0 && c->out_h % p.tile_h == 0 &&
                     tiled_lds_bytes(p.win_bytes, p.nbuf, true) + tiled_pyramid_lds_bytes(p.tile_w, p.tile_h) <= kLdsPerCU;
    snprintf(tmp, sizeof tmp, "remap_tiled_kernel<%s, %s, %s, false, %d, %d, %d>", v ? "true" : "false",
             c->n_black > 0 ? "true" : "false", pyr ? "true" : "false", p.tile_w, tile_threads(p.tile_w, p.tile_h), p.nbuf);
  } else {
    snprintf(tmp, sizeof tmp, "remap_gather_u8_kernel<%s>", v ? "true" : "false");
  }
  if (strlen(tmp) + 1 > cap) return fail(c, MDC_ERR_ARG, "mdc_describe_launch: buffer too small");
  memcpy(buf, tmp, strlen(tmp) + 1);
  return MDC_OK;
} MDC_CATCH(c)

int mdc_synchronize(mdc_ctx* c) try {
  if (!c) return MDC_ERR_ARG;
  ReadLock lk(c->mu);
  DeviceGuard dg(c->device);
  std::vector<hipStream_t> streams;
  {
    std::lock_guard<std::mutex> sl(c->slot_mu);
    for (mdc_ctx::HostSlot* h : c->slots) streams.push_back(h->stream);
  }
  for (int k = 0; k < 2; k++)
    if (c->pipe_stream[k]) streams.push_back(c->pipe_stream[k]);
  if (c->pipe_up_stream) streams.push_back(c->pipe_up_stream);
  for (hipStream_t st : streams) MDC_HIP(c, hipStreamSynchronize(st));
  return MDC_OK;
} MDC_CATCH(c)

// ---- host-pointer single-frame calls ---------------------------------------------

int mdc_unmap_host(mdc_ctx* c, const uint8_t* in, float* out, int n, unsigned flags) try {
  if (!c) return MDC_ERR_ARG;
  if (!in || !out || n < 0) return fail(c, MDC_ERR_ARG, "mdc_unmap_host: bad argument");
  ReadLock lk(c->mu);
  DeviceGuard dg(c->device);
  if (n == 0) return MDC_OK;
  bool g, v, o;
  normalise(c, flags, g, v, o);
  // The reference asserts n == w*h (compiled out under NDEBUG, :191) and would read
  // vignetteMapInv[i] for i < n; with the vignette on we refuse a mismatching n.
  if (v && (int64_t)n != (int64_t)c->in_w * c->in_h)
    return fail(c, MDC_ERR_SIZE, "unMapImage: n = %d but the vignette holds %d pixels", n, c->in_w * c->in_h);
  SlotLease slot(c);
  if (!slot.s) return MDC_ERR_HIP;
  maybe_pin(c, 0, out, (size_t)n * sizeof(float));
  const uint8_t* z_in = device_view(c, in, (size_t)n);
  float* z_out = device_view(c, out, (size_t)n * sizeof(float));
  int rc = ensure_stage(c, slot.s, z_in ? 0 : (size_t)n, z_out ? 0 : (size_t)n * sizeof(float));
  if (rc != MDC_OK) return rc;
  hipStream_t st = slot.s->stream;
  if (!z_in) MDC_HIP(c, hipMemcpyAsync(slot.s->d_in, in, (size_t)n, hipMemcpyHostToDevice, st));
  MDC_HIP(c, launch_unmap(z_in ? z_in : (const uint8_t*)slot.s->d_in, z_out ? z_out : slot.s->d_out, lut_for(c, g, o), v ? c->d_vinv : nullptr, n,
                          1, 1, st));
  if (!z_out) MDC_HIP(c, hipMemcpyAsync(out, slot.s->d_out, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, st));
  MDC_HIP(c, hipStreamSynchronize(st));
  slot.drained();
  return MDC_OK;
} MDC_CATCH(c)

static int undistort_host(mdc_ctx* c, const void* in, bool is_f32, float* out, int n_in, int n_out) {
  if (!c) return MDC_ERR_ARG;
  if (!in || !out) return fail(c, MDC_ERR_ARG, "undistort: NULL buffer");
  ReadLock lk(c->mu);
  DeviceGuard dg(c->device);
  if (!c->valid_remap) return fail(c, MDC_ERR_STATE, "no remap set (UndistorterFOV invalid)");
  if (n_in != c->rm_in_w * c->rm_in_h)
    return fail(c, MDC_ERR_SIZE, "undistort called with wrong input image dimensions (expected %d pixel, got %d pixel)",
                c->rm_in_w * c->rm_in_h, n_in);
  if (n_out != c->out_w * c->out_h)
    return fail(c, MDC_ERR_SIZE, "undistort called with wrong output image dimensions (expected %d pixel, got %d pixel)",
                c->out_w * c->out_h, n_out);
  const size_t in_bytes = (size_t)n_in * (is_f32 ? 4 : 1);
  SlotLease slot(c);
  if (!slot.s) return MDC_ERR_HIP;
  if (is_f32) maybe_pin(c, 1, in, in_bytes);
  const void* z_in = device_view(c, in, in_bytes);
  float* z_out = device_view(c, out, (size_t)n_out * sizeof(float));
  int rc = ensure_stage(c, slot.s, z_in ? 0 : in_bytes, z_out ? 0 : (size_t)n_out * sizeof(float));
  if (rc != MDC_OK) return rc;
  hipStream_t st = slot.s->stream;
  if (!z_in) MDC_HIP(c, hipMemcpyAsync(slot.s->d_in, in, in_bytes, hipMemcpyHostToDevice, st));
  const void* src = z_in ? z_in : slot.s->d_in;
  float* dst = z_out ? z_out : slot.s->d_out;
  if (is_f32) rc = enqueue_undistort_f32(c, (const float*)src, dst, 1, st);
  else rc = enqueue_process(c, (const uint8_t*)src, dst, 1, MDC_RECTIFY, st);
  if (rc != MDC_OK) {
    (void)hipStreamSynchronize(st);  // the upload borrows the caller's buffer: not in flight after the call
    return rc;
  }
  if (!z_out) MDC_HIP(c, hipMemcpyAsync(out, slot.s->d_out, (size_t)n_out * sizeof(float), hipMemcpyDeviceToHost, st));
  MDC_HIP(c, hipStreamSynchronize(st));
  slot.drained();
  return MDC_OK;
}

int mdc_undistort_host_f32(mdc_ctx* c, const float* in, float* out, int n_in, int n_out) try {
  return undistort_host(c, in, true, out, n_in, n_out);
} MDC_CATCH(c)
int mdc_undistort_host_u8(mdc_ctx* c, const uint8_t* in, float* out, int n_in, int n_out) try {
  return undistort_host(c, in, false, out, n_in, n_out);
} MDC_CATCH(c)

int mdc_process_host(mdc_ctx* c, const uint8_t* raw, float* out, unsigned flags) try {
  if (!c) return MDC_ERR_ARG;
  if (!raw || !out) return fail(c, MDC_ERR_ARG, "mdc_process_host: NULL buffer");
  ReadLock lk(c->mu);
  DeviceGuard dg(c->device);
  const bool rect = (flags & MDC_RECTIFY) != 0;
  if (rect && !c->valid_remap) return fail(c, MDC_ERR_STATE, "no remap set (UndistorterFOV invalid)");
  const int iw = (rect || c->in_w <= 0) ? c->rm_in_w : c->in_w, ih = (rect || c->in_h <= 0) ? c->rm_in_h : c->in_h;
  if (iw <= 0 || ih <= 0) return fail(c, MDC_ERR_STATE, "frame size unknown");
  const size_t n_in = (size_t)iw * ih;
  const size_t n_out = rect ? (size_t)c->out_w * c->out_h : n_in;
  SlotLease slot(c);
  if (!slot.s) return MDC_ERR_HIP;
  const uint8_t* z_in = device_view(c, raw, n_in);
  float* z_out = device_view(c, out, n_out * sizeof(float));
  int rc = ensure_stage(c, slot.s, z_in ? 0 : n_in, z_out ? 0 : n_out * sizeof(float));
  if (rc != MDC_OK) return rc;
  hipStream_t st = slot.s->stream;
  if (!z_in) MDC_HIP(c, hipMemcpyAsync(slot.s->d_in, raw, n_in, hipMemcpyHostToDevice, st));
  rc = enqueue_process(c, z_in ? z_in : (const uint8_t*)slot.s->d_in, z_out ? z_out : slot.s->d_out, 1, flags, st);
  if (rc != MDC_OK) {
    (void)hipStreamSynchronize(st);
    return rc;
  }
  if (!z_out) MDC_HIP(c, hipMemcpyAsync(out, slot.s->d_out, n_out * sizeof(float), hipMemcpyDeviceToHost, st));
  MDC_HIP(c, hipStreamSynchronize(st));
  slot.drained();
  return MDC_OK;
} MDC_CATCH(c)

// ---- host-pointer, many frames: copies and kernels overlapped ------------------------------

void* mdc_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}
void mdc_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

// Common body of the pipelined host calls: frame i comes from raw[i] (bytes), or, with `rec`, from the JPEG coefficient
// record rec[i] through the device-side inverse DCT, or, with `strm`, from the JPEG stream strm[i] (strm_bytes[i] bytes)
// through the device-side Huffman decoder and the inverse DCT (status: per-frame decode status, may be NULL).
static int process_frames_pipeline(mdc_ctx* c, const uint8_t* const* raw, const void* const* rec, int64_t record_bytes, int blocks_w,
                                   int blocks_rows, float* const* out, int64_t nframes, unsigned flags, const char* who,
                                   const void* const* strm = nullptr, const int64_t* strm_bytes = nullptr, int* status = nullptr) {
  if (!c) return MDC_ERR_ARG;
  if (nframes < 0 || (nframes > 0 && ((!raw && !rec && !strm) || !out || (strm && !strm_bytes)))) return fail(c, MDC_ERR_ARG, "%s: bad argument", who);
  ReadLock lk(c->mu);
  std::lock_guard<std::mutex> pipe_lk(c->pipe_mu);  // one pipelined call at a time per context (it overlaps internally)
  DeviceGuard dg(c->device);
  const bool rect = (flags & MDC_RECTIFY) != 0;
  if (rect && !c->valid_remap) return fail(c, MDC_ERR_STATE, "no remap set (UndistorterFOV invalid)");
  const int iw = (rect || c->in_w <= 0) ? c->rm_in_w : c->in_w, ih = (rect || c->in_h <= 0) ? c->rm_in_h : c->in_h;
  if (iw <= 0 || ih <= 0) return fail(c, MDC_ERR_STATE, "frame size unknown");
  const size_t n_in = (size_t)iw * ih;
  const size_t n_out = rect ? (size_t)c->out_w * c->out_h : n_in;
  size_t strm_stride = 0;
  if (strm) {  // streams are decoded into records of the reader's geometry (block grid rounded up to multiples of 4)
    blocks_w = ((iw + 7) / 8 + 3) & ~3;
    blocks_rows = ((ih + 7) / 8 + 3) & ~3;
    record_bytes = 128 + (int64_t)blocks_w * blocks_rows * 128;
    for (int64_t i = 0; i < nframes; i++) {
      if (strm_bytes[i] < (int64_t)sizeof(mdc_jpeg_stream_header) + 17 || strm_bytes[i] > (1ll << 28))
        return fail(c, MDC_ERR_ARG, "%s: stream %lld has an impossible size", who, (long long)i);
      strm_stride = std::max(strm_stride, (size_t)strm_bytes[i]);
    }
    strm_stride = (strm_stride + 15) & ~(size_t)15;
  }
  if ((rec || strm) && (blocks_w < (iw + 7) / 8 || blocks_rows < (ih + 7) / 8 || record_bytes % 16 != 0 ||
                        record_bytes < 128 + (int64_t)blocks_w * blocks_rows * 128))
    return fail(c, MDC_ERR_ARG, "%s: coefficient records do not describe a %dx%d frame", who, iw, ih);
  for (int64_t i = 0; i < nframes; i++)
    if (!(strm ? strm[i] : rec ? rec[i] : (const void*)raw[i]) || !out[i])
      return fail(c, MDC_ERR_ARG, "%s: frame %lld has a NULL buffer", who, (long long)i);
  if (status)
    for (int64_t i = 0; i < nframes; i++) status[i] = 0;
  if (strm && status && c->pipe_status_cap < (size_t)nframes) {  // status words come back asynchronously: page-locked landing buffer
    if (c->h_pipe_status) (void)hipHostFree(c->h_pipe_status);
    c->h_pipe_status = nullptr;
    c->pipe_status_cap = 0;
    const size_t cap = std::max<size_t>(256, (size_t)nframes * 2);
    MDC_HIP(c, hipHostMalloc((void**)&c->h_pipe_status, cap * sizeof(int), hipHostMallocDefault));
    c->pipe_status_cap = cap;
  }
  int* d_host_status = nullptr;  // the device's view of h_pipe_status
  if (strm && status && c->h_pipe_status && hipHostGetDevicePointer((void**)&d_host_status, c->h_pipe_status, 0) != hipSuccess) {
    (void)hipGetLastError();
    d_host_status = nullptr;
  }
  constexpr int kChunk = 16;  // frames per slot: one kernel launch (two with the inverse DCT), 2 x 16 async copies
  // Zero copy (device_view): results go straight into the caller's images when every one of them is mapped page-locked
  // memory, frames are read straight from the caller's buffers when every one of them is (coefficient records are always
  // copied: the inverse DCT reads a record 16 bytes at a time per thread, uncached that would cross PCIe several times).
  static const bool trace = getenv("MDC_PIPE_TRACE") != nullptr;  // where a pipelined call spends its host time (stderr)
  const auto t_begin = std::chrono::steady_clock::now();
  auto since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
  std::vector<float*> z_out((size_t)nframes);
  std::vector<const uint8_t*> z_in((rec || strm) ? 0 : (size_t)nframes);
  // Streams: results are staged on the device and leave with ONE copy per run of images that lie back to back (the reader's pool
  // hands out slabs): while a kernel writes results across PCIe itself, the kernels of other streams make no progress -- the
  // Huffman launch of the next chunk finished 0.8 ms (its own time) after the output of the current one, however few
  // workgroups the output launch had --, the copy engine moves the same bytes at the same 52 GB/s and leaves the CUs alone
  // (256 frames: 10.4 -> 7.5 ms, profiles/r03_experiments/20_*).
  bool zc_out = nframes > 0 && !strm, zc_in = !rec && !strm && nframes > 0;
  for (int64_t i = 0; i < nframes && zc_out; i++) zc_out = (z_out[(size_t)i] = device_view(c, out[i], n_out * sizeof(float))) != nullptr;
  for (int64_t i = 0; i < nframes && zc_in; i++) zc_in = (z_in[(size_t)i] = device_view(c, raw[i], n_in)) != nullptr;
  if (!zc_out) zc_in = false;  // frames alone: the copy pipeline (one launch per chunk) stays
  if (zc_out) {
    // Zero copy pays here when frames lie back to back (rows of one block: one launch per chunk reads / writes them in place).
    // Scattered images -- the reader's pool -- would mean one single-frame launch each: measured next to the decode stream
    // those run at 47 us per frame where one batched launch + the DMA engines' copies out take 25 (experiment 14).
    int64_t runs = 1;
    for (int64_t i = 1; i < nframes; i++)
      if (z_out[(size_t)i] != z_out[(size_t)i - 1] + n_out || (zc_in && z_in[(size_t)i] != z_in[(size_t)i - 1] + n_in)) runs++;
    if (runs * 8 > nframes && nframes >= 8) zc_out = zc_in = false;
  }
  const double t_views = since();
  // frames per slot.  Streams: 64 -- the Huffman kernel's time does not depend on the frame count up to ~64 (one workgroup per
  // frame, 1.3 ms), so small chunks would only repeat that latency; nothing staged: the chunk only alternates the streams
  const int chunk = (zc_in && zc_out) ? 64 : (strm ? 64 : kChunk);
  const bool inplace_out = zc_out;
  const size_t in_need = zc_in ? 0 : chunk * n_in, out_need = inplace_out ? 0 : chunk * n_out * sizeof(float);
  const size_t rec_need = (rec || strm) ? (size_t)chunk * (size_t)record_bytes : 0;
  const size_t strm_need = strm ? (size_t)chunk * strm_stride : 0;
  if (c->pipe_in_cap < in_need || c->pipe_out_cap < out_need || c->pipe_rec_cap < rec_need || c->pipe_strm_cap < strm_need || !c->pipe_stream[0] ||
      !c->pipe_stream[1]) {
    const size_t in_cap = std::max(in_need, c->pipe_in_cap), out_cap = std::max(out_need, c->pipe_out_cap), rec_cap = std::max(rec_need, c->pipe_rec_cap);
    const size_t strm_cap = std::max(strm_need + strm_need / 4, c->pipe_strm_cap);  // (stream sizes vary from call to call: some headroom)
    c->pipe_in_cap = c->pipe_out_cap = c->pipe_rec_cap = c->pipe_strm_cap = 0;  // a failure part-way leaves "no slots", not stale capacities
    for (int k = 0; k < 2; k++) {
      if (c->pipe_stream[k]) MDC_HIP(c, hipStreamSynchronize(c->pipe_stream[k]));
      if (!c->pipe_stream[k]) MDC_HIP(c, hipStreamCreateWithFlags(&c->pipe_stream[k], hipStreamNonBlocking));
      if (!c->pipe_done[k]) MDC_HIP(c, hipEventCreateWithFlags(&c->pipe_done[k], hipEventDisableTiming));
      if (!c->pipe_dec[k]) MDC_HIP(c, hipEventCreateWithFlags(&c->pipe_dec[k], hipEventDisableTiming));
      for (void** p : {(void**)&c->d_pipe_in[k], (void**)&c->d_pipe_out[k], &c->d_pipe_rec[k], &c->d_pipe_strm[k], (void**)&c->d_pipe_status[k]})
        if (*p) {
          (void)hipFree(*p);
          *p = nullptr;
        }
      if (in_cap) MDC_HIP(c, hipMalloc(&c->d_pipe_in[k], in_cap));
      if (out_cap) MDC_HIP(c, hipMalloc(&c->d_pipe_out[k], out_cap));
      if (rec_cap) MDC_HIP(c, hipMalloc(&c->d_pipe_rec[k], rec_cap));
      if (strm_cap) MDC_HIP(c, hipMalloc(&c->d_pipe_strm[k], strm_cap));
      if (strm_cap) MDC_HIP(c, hipMalloc((void**)&c->d_pipe_status[k], 64 * sizeof(int)));
    }
    c->pipe_in_cap = in_cap;
    c->pipe_out_cap = out_cap;
    c->pipe_rec_cap = rec_cap;
    c->pipe_strm_cap = strm_cap;
  }
  if (strm && !c->pipe_up_stream) {
    MDC_HIP(c, hipStreamCreateWithFlags(&c->pipe_up_stream, hipStreamNonBlocking));
    for (int k = 0; k < 2; k++) {
      if (!c->pipe_up[k]) MDC_HIP(c, hipEventCreateWithFlags(&c->pipe_up[k], hipEventDisableTiming));
      if (!c->pipe_huff[k]) MDC_HIP(c, hipEventCreateWithFlags(&c->pipe_huff[k], hipEventDisableTiming));
    }
  }
  // chunk k runs entirely on stream k%2 (H2D, kernel(s), D2H in order); the two streams overlap one
  // chunk's copies with the other's kernel.  Re-using a slot waits for its previous chunk.
  // On a failure the loop stops, BOTH streams are drained (asynchronous copies into the caller's buffers
  // may still be in flight) and only then the error is returned.
  int rc = MDC_OK;
  hipError_t he = hipSuccess;
  const char* what = "";
#define MDC_PIPE(call)              \
  if (he == hipSuccess) {           \
    he = (call);                    \
    if (he != hipSuccess) what = #call; \
  }
  int n = 0;
  // MDC_PIPE_TRACE: per chunk 6 time stamps [upload: start, done; decode stream: Huffman done, decoded; output stream: start, done]
  const int64_t nchunks = (nframes + chunk - 1) / chunk;
  std::vector<hipEvent_t> tev(trace ? (size_t)nchunks * 6 : 0, (hipEvent_t) nullptr);
  auto stamp = [&](int64_t kk, int j, hipStream_t st) {
    if (!trace) return;
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) == hipSuccess) {
      (void)hipEventRecord(e, st);
      tev[(size_t)kk * 6 + j] = e;
    }
  };
  for (int64_t f0 = 0, k = 0; f0 < nframes && rc == MDC_OK && he == hipSuccess; f0 += n, k++) {
    const int slot = (int)(k & 1);
    // Streams: ALL chunks decode on stream 0 (upload, Huffman kernel, inverse DCT) and go out on stream 1 (fused pass into the
    // caller's images), tied by events -- the Huffman kernel takes ~1.3 ms whatever the frame count (one workgroup per frame)
    // and the output is PCIe-bound, so chunk k+1 decodes while chunk k goes out.  Otherwise chunk k runs on stream k % 2.
    hipStream_t s = strm ? c->pipe_stream[0] : c->pipe_stream[slot];
    hipStream_t s_out = strm ? c->pipe_stream[1] : s;
    // (a smaller first chunk, to get the output going earlier, is slower: the Huffman launch takes ~0.75 ms whatever its frame
    // count, so more chunks only lengthen the decode stream -- profiles/r03_experiments/19_*)
    n = (int)std::min<int64_t>(chunk, nframes - f0);
    if (k >= 2 && strm) {  // the slot's staging is free again: in stream order ...
      MDC_PIPE(hipStreamWaitEvent(s, c->pipe_done[slot], 0));
    } else if (k >= 2 && !(zc_in && zc_out)) {  // ... or on the host
      MDC_PIPE(hipEventSynchronize(c->pipe_done[slot]));
    }
    // a chunk whose sources lie at one stride in host memory (the reader's ring) goes up as ONE strided copy: 64 separate
    // copies of a 270-KB stream cost the decode stream ~1 ms of the ~2.5 ms a chunk takes
    auto upload = [&](int64_t f0, int n, void* d_dst, size_t d_stride, const void* const* src, const int64_t* bytes, size_t fixed_bytes, hipStream_t s) {
      size_t width = fixed_bytes;
      for (int i = 0; i < n && bytes; i++) width = std::max(width, (size_t)bytes[f0 + i]);
      ptrdiff_t pitch = n > 1 ? (const char*)src[f0 + 1] - (const char*)src[f0] : 0;
      for (int i = 2; i < n && pitch > 0; i++)
        if ((const char*)src[f0 + i] - (const char*)src[f0 + i - 1] != pitch) pitch = 0;
      // A strided copy reads `width` bytes of every row but the last: beyond a SHORTER stream's own bytes, up to the next
      // buffer.  The contract only promises bytes[i] readable bytes per stream, so the strided form is taken when every row
      // is `width` long anyway, or when the whole span is one page-locked allocation (the reader's ring: the gaps are its own
      // memory); separately allocated buffers that merely happen to sit at equal spacing go up one by one.
      bool rows_full = true;
      for (int i = 0; i + 1 < n && bytes; i++) rows_full = rows_full && (size_t)bytes[f0 + i] == width;
      if (n > 1 && pitch >= (ptrdiff_t)width && width <= d_stride &&
          (rows_full || one_host_allocation(src[f0], (size_t)pitch * (size_t)(n - 1) + (bytes ? (size_t)bytes[f0 + n - 1] : fixed_bytes)))) {
        // rows of `width` bytes: a shorter source is followed by the next one within the pitch, except the LAST -- it goes up
        // with its own size (nothing is read beyond the end of the caller's last buffer)
        const size_t last = bytes ? (size_t)bytes[f0 + n - 1] : fixed_bytes;
        const int rows2d = last == width ? n : n - 1;
        MDC_PIPE(hipMemcpy2DAsync(d_dst, d_stride, src[f0], (size_t)pitch, width, (size_t)rows2d, hipMemcpyHostToDevice, s));
        if (rows2d < n) MDC_PIPE(hipMemcpyAsync((char*)d_dst + (size_t)(n - 1) * d_stride, src[f0 + n - 1], last, hipMemcpyHostToDevice, s));
      } else {
        for (int i = 0; i < n; i++)
          MDC_PIPE(hipMemcpyAsync((char*)d_dst + (size_t)i * d_stride, src[f0 + i], bytes ? (size_t)bytes[f0 + i] : fixed_bytes, hipMemcpyHostToDevice, s));
      }
    };
    if (strm) {
      // Uploads run ahead on their own stream: chunk k+1's streams are ENQUEUED before anything of chunk k (the copy queues
      // work in submission order -- an upload submitted after chunk k's copy out would wait behind it) and go up as soon as the
      // Huffman launch of chunk k-1 has read the buffer, i.e. under the decode of chunk k and the output of chunk k-1.
      auto enqueue_upload = [&](int64_t kk) {
        const int sl = (int)(kk & 1);
        const int64_t uf0 = kk * chunk;
        const int un = (int)std::min<int64_t>(chunk, nframes - uf0);
        hipStream_t up = c->pipe_up_stream;
        if (kk >= 2) MDC_PIPE(hipStreamWaitEvent(up, c->pipe_huff[sl], 0));
        stamp(kk, 0, up);
        upload(uf0, un, c->d_pipe_strm[sl], strm_stride, strm, strm_bytes, 0, up);
        MDC_PIPE(hipEventRecord(c->pipe_up[sl], up));
        stamp(kk, 1, up);
      };
      if (k == 0) enqueue_upload(0);
      if (f0 + n < nframes) enqueue_upload(k + 1);
      MDC_PIPE(hipStreamWaitEvent(s, c->pipe_up[slot], 0));
      // the status words land in page-locked host memory directly (a copy would queue behind the results going out)
      int* d_status = (status && d_host_status) ? d_host_status + f0 : c->d_pipe_status[slot];
      MDC_PIPE(launch_jpeg_huffman(c->d_pipe_strm[slot], (int64_t)strm_stride, c->d_pipe_rec[slot], record_bytes, iw, ih, blocks_w, blocks_rows, n,
                                   d_status, s));
      MDC_PIPE(hipEventRecord(c->pipe_huff[slot], s));
      stamp(k, 2, s);
      if (status && !d_host_status)
        MDC_PIPE(hipMemcpyAsync(c->h_pipe_status + f0, c->d_pipe_status[slot], (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s));
      MDC_PIPE(launch_jpeg_idct(c->d_pipe_rec[slot], record_bytes, c->d_pipe_in[slot], iw, ih, blocks_w, blocks_rows, n, s));
    } else if (rec) {
      stamp(k, 0, s);
      upload(f0, n, c->d_pipe_rec[slot], (size_t)record_bytes, rec, nullptr, (size_t)record_bytes, s);
      stamp(k, 1, s);
      MDC_PIPE(launch_jpeg_idct(c->d_pipe_rec[slot], record_bytes, c->d_pipe_in[slot], iw, ih, blocks_w, blocks_rows, n, s));
    } else {
      stamp(k, 0, s);
      if (!zc_in) upload(f0, n, c->d_pipe_in[slot], n_in, reinterpret_cast<const void* const*>(raw), nullptr, n_in, s);
      stamp(k, 1, s);
    }
    if (he != hipSuccess) break;
    stamp(k, 3, s);
    if (strm) {
      MDC_PIPE(hipEventRecord(c->pipe_dec[slot], s));
      MDC_PIPE(hipStreamWaitEvent(s_out, c->pipe_dec[slot], 0));
      if (he != hipSuccess) break;
    }
    stamp(k, 4, s_out);
    if (inplace_out) {  // one launch per run of frames that lie back to back on both sides
      for (int i = 0; i < n && rc == MDC_OK;) {
        const uint8_t* src = zc_in ? z_in[(size_t)(f0 + i)] : c->d_pipe_in[slot] + (size_t)i * n_in;
        float* dst = z_out[(size_t)(f0 + i)];
        int run = 1;
        while (i + run < n && z_out[(size_t)(f0 + i + run)] == dst + (size_t)run * n_out &&
               (!zc_in || z_in[(size_t)(f0 + i + run)] == src + (size_t)run * n_in))
          run++;
        rc = enqueue_process(c, src, dst, run, flags, s_out);
        i += run;
      }
      if (rc != MDC_OK) break;
    } else {
      rc = enqueue_process(c, c->d_pipe_in[slot], c->d_pipe_out[slot], n, flags, s_out);
      if (rc != MDC_OK) break;
      for (int i = 0; i < n;) {  // one copy per run of images that lie back to back in the caller's memory
        int run = 1;
        while (i + run < n && out[f0 + i + run] == out[f0 + i] + (size_t)run * n_out) run++;
        MDC_PIPE(hipMemcpyAsync(out[f0 + i], c->d_pipe_out[slot] + (size_t)i * n_out, (size_t)run * n_out * sizeof(float),
                                hipMemcpyDeviceToHost, s_out));
        i += run;
      }
    }
    MDC_PIPE(hipEventRecord(c->pipe_done[slot], s_out));
    stamp(k, 5, s_out);
  }
  const double t_enqueued = since();
  if (strm && c->pipe_up_stream) {
    const hipError_t e = hipStreamSynchronize(c->pipe_up_stream);
    if (he == hipSuccess && e != hipSuccess) {
      he = e;
      what = "hipStreamSynchronize(pipe_up_stream)";
    }
  }
  for (int k = 0; k < 2; k++) {
    const hipError_t e = hipStreamSynchronize(c->pipe_stream[k]);
    if (he == hipSuccess && e != hipSuccess) {
      he = e;
      what = "hipStreamSynchronize(pipe_stream)";
    }
  }
#undef MDC_PIPE
  if (trace && !tev.empty() && tev[0] && he == hipSuccess && rc == MDC_OK) {
    std::fprintf(stderr, "%s: chunks, ms since the first upload [upload from-to | decode: Huffman done, decoded | out: from-to]", who);
    for (size_t q = 0; q + 5 < tev.size(); q += 6) {
      float t[6] = {-1, -1, -1, -1, -1, -1};
      for (int j = 0; j < 6; j++)
        if (tev[q + j]) (void)hipEventElapsedTime(&t[j], tev[0], tev[q + j]);
      std::fprintf(stderr, "  [%.2f-%.2f | %.2f, %.2f | %.2f-%.2f]", t[0], t[1], t[2], t[3], t[4], t[5]);
    }
    std::fprintf(stderr, "\n");
  }
  for (hipEvent_t e : tev)
    if (e) (void)hipEventDestroy(e);
  if (trace)
    std::fprintf(stderr, "%s: %lld frames: buffer queries %.2f ms, everything enqueued at %.2f ms, streams drained at %.2f ms\n", who, (long long)nframes,
                 t_views, t_enqueued, since());
  if (rc != MDC_OK) return rc;
  if (he != hipSuccess) return fail(c, MDC_ERR_HIP, "%s: %s", what, hipGetErrorString(he));
  if (strm && status) memcpy(status, c->h_pipe_status, (size_t)nframes * sizeof(int));
  return MDC_OK;
}

int mdc_process_frames_host(mdc_ctx* c, const uint8_t* const* raw, float* const* out, int64_t nframes, unsigned flags) try {
  return process_frames_pipeline(c, raw, nullptr, 0, 0, 0, out, nframes, flags, "mdc_process_frames_host");
} MDC_CATCH(c)

int mdc_process_jpeg_frames_host(mdc_ctx* c, const void* const* records, int64_t record_bytes, int blocks_w, int blocks_rows,
                                 float* const* out, int64_t nframes, unsigned flags) try {
  return process_frames_pipeline(c, nullptr, records, record_bytes, blocks_w, blocks_rows, out, nframes, flags, "mdc_process_jpeg_frames_host");
} MDC_CATCH(c)

int mdc_process_jpeg_streams_host(mdc_ctx* c, const void* const* streams, const int64_t* stream_bytes, float* const* out, int64_t nframes,
                                  unsigned flags, int* status) try {
  return process_frames_pipeline(c, nullptr, nullptr, 0, 0, 0, out, nframes, flags, "mdc_process_jpeg_streams_host", streams, stream_bytes, status);
} MDC_CATCH(c)

int mdc_jpeg_huffman_batch_device(mdc_ctx* c, const void* d_streams, int64_t stream_stride, void* d_records, int64_t record_bytes, int w, int h,
                                  int blocks_w, int blocks_rows, int64_t nframes, int* d_status, void* stream) try {
  if (!c) return MDC_ERR_ARG;
  if (!d_streams || !d_records || !d_status || nframes < 0 || w <= 0 || h <= 0) return fail(c, MDC_ERR_ARG, "mdc_jpeg_huffman_batch_device: bad argument");
  ReadLock lk(c->mu);
  DeviceGuard dg(c->device);
  MDC_HIP(c, launch_jpeg_huffman(d_streams, stream_stride, d_records, record_bytes, w, h, blocks_w, blocks_rows, nframes, d_status, (hipStream_t)stream));
  return MDC_OK;
} MDC_CATCH(c)

int mdc_jpeg_idct_batch_device(mdc_ctx* c, const void* d_records, int64_t record_bytes, uint8_t* d_frames, int w, int h, int blocks_w,
                               int blocks_rows, int64_t nframes, void* stream) try {
  if (!c) return MDC_ERR_ARG;
  if (!d_records || !d_frames || nframes < 0 || w <= 0 || h <= 0) return fail(c, MDC_ERR_ARG, "mdc_jpeg_idct_batch_device: bad argument");
  ReadLock lk(c->mu);
  DeviceGuard dg(c->device);
  MDC_HIP(c, launch_jpeg_idct(d_records, record_bytes, d_frames, w, h, blocks_w, blocks_rows, nframes, (hipStream_t)stream));
  return MDC_OK;
} MDC_CATCH(c)

// ---- table hand-over ------------------------------------------------------------------

int mdc_export_tables(mdc_ctx* c, void* blob, size_t cap, size_t* size) try {
  if (!c || !size) return MDC_ERR_ARG;
  ReadLock lk(c->mu);
  const size_t nv = c->valid_vignette ? c->h_vinv.size() : 0;
  const size_t nr = c->valid_remap ? c->h_rx.size() : 0;
  const size_t need = sizeof(BlobHeader) + 256 * 4 + nv * 4 + 2 * nr * 4;
  *size = need;
  if (!blob) return MDC_OK;
  if (cap < need) return fail(c, MDC_ERR_ARG, "export buffer too small (%zu < %zu)", cap, need);
  BlobHeader h{kMagic, 1, c->in_w, c->in_h, c->rm_in_w, c->rm_in_h, c->out_w, c->out_h,
               c->valid_gamma, c->valid_vignette, c->valid_remap, 0};
  char* p = (char*)blob;
  memcpy(p, &h, sizeof h);
  p += sizeof h;
  if (c->valid_gamma) memcpy(p, c->h_ginv.data(), 256 * 4);
  else memset(p, 0, 256 * 4);
  p += 256 * 4;
  if (nv) memcpy(p, c->h_vinv.data(), nv * 4);
  p += nv * 4;
  if (nr) {
    memcpy(p, c->h_rx.data(), nr * 4);
    p += nr * 4;
    memcpy(p, c->h_ry.data(), nr * 4);
  }
  return MDC_OK;
} MDC_CATCH(c)

int mdc_import_tables(mdc_ctx* c, const void* blob, size_t size) try {
  if (!c || !blob) return MDC_ERR_ARG;
  BlobHeader h;
  if (size < sizeof h) return fail(c, MDC_ERR_ARG, "table blob truncated");
  memcpy(&h, blob, sizeof h);
  if (h.magic != kMagic || h.version != 1) return fail(c, MDC_ERR_ARG, "table blob has wrong magic/version");
  const size_t nv = h.valid_vignette ? (size_t)h.in_w * h.in_h : 0;
  const size_t nr = h.valid_remap ? (size_t)h.out_w * h.out_h : 0;
  if (size != sizeof h + 256 * 4 + nv * 4 + 2 * nr * 4) return fail(c, MDC_ERR_ARG, "table blob has wrong size");
  const char* p = (const char*)blob + sizeof h;
  const float* ginv = (const float*)p;
  p += 256 * 4;
  const float* vinv = (const float*)p;
  p += nv * 4;
  const float* rx = (const float*)p;
  const float* ry = rx + nr;
  // one critical section: no other thread may see the new photometric tables next to the old remap
  WriteLock lk(c->mu);
  int rc = MDC_OK;
  if (h.in_w > 0 && h.in_h > 0) {
    rc = set_photometric_locked(c, h.valid_gamma ? ginv : nullptr, h.valid_vignette ? vinv : nullptr, h.in_w, h.in_h);
  } else {  // the blob carries no photometric calibration: neither does the context afterwards
    DeviceGuard dg(c->device);
    MDC_HIP(c, hipDeviceSynchronize());
    c->in_w = c->in_h = 0;
    c->valid_gamma = c->valid_vignette = false;
    c->h_vinv.clear();
    if (c->d_vinv) {
      (void)hipFree(c->d_vinv);
      c->d_vinv = nullptr;
    }
    rc = upload_luts(c);
  }
  if (rc != MDC_OK) return rc;
  if (h.valid_remap) rc = set_remap_locked(c, rx, ry, h.rm_in_w, h.rm_in_h, h.out_w, h.out_h);
  else rc = set_remap_locked(c, nullptr, nullptr, 0, 0, 0, 0);
  return rc;
} MDC_CATCH(c)

}  // extern "C"
