// libmdc_hip.so -- implementation of the C ABI in include/mdc_hip.h.
//
// Host-side responsibilities only: device copies of the calibration tables, the
// tile plan of the LDS-staged remap kernel, flag normalisation exactly as
// src/PhotometricUndistorter.cpp:173-189 (reference repo), staging for the
// host-pointer calls.  There is NO CPU implementation of the per-frame maths in
// this library: without a HIP device every entry point fails with
// MDC_ERR_NO_DEVICE.
#include "mdc_ctx.h"

using namespace mdc;


namespace mdc {

static thread_local std::string g_create_err;
// mdc_last_error(ctx) returns the calling thread's own last failure on that context if it had one (several threads may
// use one context), else the context's most recent one
static thread_local std::string t_err, t_err_other;
static thread_local const mdc_ctx* t_err_ctx = nullptr;

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

int frames_per_block(const mdc_ctx* c, int64_t nframes, int blocks_per_group, int target_wgs) {
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
                    float* const* pyr, bool* pyr_done) {
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

// base + box levels (+ gradient images when d_dI / d_abs_squared_grad are given) of nframes device-resident raw frames, in chunks;
// lock held by the caller (mdc_process_pyramid_gradients_batch_device and the *_to_device pipeline, mdc_pipeline.hip)
int enqueue_pyramid_gradients(mdc_ctx* c, const uint8_t* d_in, float* d_base, int levels, float* const* d_levels, float* const* d_dI,
                              float* const* d_abs_squared_grad, int64_t nframes, unsigned flags, int chunk_frames, hipStream_t s) {
  const bool with_gradients = d_dI != nullptr && d_abs_squared_grad != nullptr;
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
  // frame; chunks exist to bound the launch sizes, and below ~100 frames they lose to their tails.  Round 4, frames/s by chunk:
  // 91: 130.6 k, 6 x 86: 133 k, 96: 137-140 k, one chunk of 512: 134-137 k -- the automatic choice is 96 at 1280 x 1024.
  int64_t chunk = chunk_frames;
  if (chunk <= 0) {  // ~96 frames' worth at 1280 x 1024, a multiple of 32 (the remap launch's frames per workgroup divide it)
    chunk = (int64_t)((672ull << 20) / level_bytes);
    chunk = chunk >= 32 ? chunk / 32 * 32 : std::max<int64_t>(1, chunk);
  }
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
#if MDC_EXP_STRIP_FAKE_GRAD  // diagnosis: the strip launch writes level 0's gradient images (traffic and shapes only), the gradient launch starts at level 1
    g_fake_grad_dI = with_gradients ? d_dI[0] + (size_t)f0 * w0 * h0 * 3 : nullptr;
    g_fake_grad_abs = with_gradients ? d_abs_squared_grad[0] + (size_t)f0 * w0 * h0 : nullptr;
#endif
    int rc = enqueue_process(c, d_in + (size_t)f0 * npi, base, n, flags, s, levels > 1 ? pyr : nullptr, &fused);
#if MDC_EXP_STRIP_FAKE_GRAD
    g_fake_grad_dI = g_fake_grad_abs = nullptr;
#endif
    if (rc != MDC_OK) return rc;
    const int first = fused ? std::min(levels, 4) : 1;
    const float* src = first == 1 ? base : lv[first - 2];
    for (int l = first; l < levels; l++) {
      MDC_HIP(c, launch_pyramid_level(src, lv[l - 1], lw[l - 1], lh[l - 1], n, s));
      src = lv[l - 1];
    }
#if MDC_EXP_STRIP_FAKE_GRAD
    constexpr int kGradFirst = 1;
#else
    constexpr int kGradFirst = 0;
#endif
    for (int l0 = kGradFirst; l0 < levels && with_gradients; l0 += 4) {  // gradients: four levels per launch
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
}

struct BlobHeader {
  uint32_t magic, version;
  int32_t in_w, in_h, rm_in_w, rm_in_h, out_w, out_h;
  int32_t valid_gamma, valid_vignette, valid_remap, pad;
};
constexpr uint32_t kMagic = 0x4d444331u;  // "MDC1"

DistortModel distort_model(const mdc_fov_model* f) {
  return make_distort_model(f->in_calib, f->in_w, f->in_h, f->out_calib, f->out_w, f->out_h);
}

}  // namespace mdc

static int set_photometric_locked(mdc_ctx* c, const float* ginv, const float* vignette_inv, int w, int h);
static int set_remap_locked(mdc_ctx* c, const float* rx, const float* ry, int in_w, int in_h, int out_w, int out_h);


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
    {  // striped buffers of mdc_device_alloc the caller never gave back: their arenas cannot be reached without the context
      std::vector<void*> arenas;
      {
        std::lock_guard<std::mutex> lk(c->striped_mu);
        for (auto& kv : c->striped) arenas.push_back(kv.second);
        c->striped.clear();
      }
      for (void* a : arenas) {
        mdc_striped_set set;
        memset(&set, 0, sizeof set);
        set.handle = a;
        (void)mdc_free_striped_set_device(c, &set);
      }
    }
    void* ptrs[] = {c->d_luts, c->d_vinv, c->d_rx, c->d_ry, c->d_vcal_max, c->d_pipe_in[0], c->d_pipe_in[1], c->d_pipe_out[0], c->d_pipe_out[1],
                    c->d_pipe_rec[0], c->d_pipe_rec[1], c->d_pipe_strm[0], c->d_pipe_strm[1], c->d_pipe_status[0], c->d_pipe_status[1], c->d_pipe_seg[0], c->d_pipe_seg[1]};
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

int mdc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

const char* mdc_last_error(const mdc_ctx* c) {
  if (!c) return g_create_err.c_str();
  if (t_err_ctx == c) return t_err.c_str();
  std::lock_guard<std::mutex> lk(c->err_mu);
  t_err_other = c->err;
  return t_err_other.c_str();
}

int mdc_device_pci_bus_id(mdc_ctx* c, char* buf, size_t cap) try {
  if (!c) return MDC_ERR_ARG;
  if (!buf || cap < 13) return fail(c, MDC_ERR_ARG, "mdc_device_pci_bus_id: buffer of >= 13 bytes expected");
  char tmp[64] = {0};
  MDC_HIP(c, hipDeviceGetPCIBusId(tmp, (int)sizeof tmp, c->device));
  for (char* q = tmp; *q; q++) *q = (char)tolower((unsigned char)*q);  // sysfs spells the address in lower case
  snprintf(buf, cap, "%s", tmp);
  return MDC_OK;
} MDC_CATCH(c)

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
    case MDC_OPT_DEVICE_PIPELINE_CHUNK:
      if (value != 0 && (value < 16 || value > 256)) return fail(c, MDC_ERR_ARG, "device pipeline chunk must be 0 (automatic) or 16..256 frames");
      c->opt_dev_chunk = value;
      return MDC_OK;
    case MDC_OPT_DEVICE_PIPELINE_CHUNK_HINT:
      if (value != 0 && (value < 16 || value > 256)) return fail(c, MDC_ERR_ARG, "device pipeline chunk hint must be 0 (none) or 16..256 frames");
      c->opt_dev_chunk_hint = value;
      return MDC_OK;
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

int mdc_distort_points_device(mdc_ctx* c, const mdc_fov_model* model, float* d_x, float* d_y, int64_t n, void* stream) try {
  if (!c) return MDC_ERR_ARG;
  if (!model || n < 0 || (n > 0 && (!d_x || !d_y))) return fail(c, MDC_ERR_ARG, "mdc_distort_points_device: bad argument");
  ReadLock lk(c->mu);
  DeviceGuard dg(c->device);
  MDC_HIP(c, launch_distort_points(d_x, d_y, n, distort_model(model), (hipStream_t)stream));
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
  return enqueue_pyramid_gradients(c, d_in, d_base, levels, d_levels, d_dI, d_abs_squared_grad, nframes, flags, chunk_frames, (hipStream_t)stream);
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

// Which of the caller's candidate buffers does the pass run fastest on?  (include/mdc_hip.h)
int mdc_tune_placement_device(mdc_ctx* c, const uint8_t* const* d_in, int n_in, float* const* d_out, int n_out, int64_t nframes, unsigned flags,
                              void* stream, int* best_in, int* best_out, float* ms_out) try {
  if (!c) return MDC_ERR_ARG;
  if (!d_in || !d_out || n_in < 1 || n_out < 1 || (int64_t)n_in * n_out > 256 || nframes <= 0 || !best_in || !best_out)
    return fail(c, MDC_ERR_ARG, "mdc_tune_placement_device: bad argument");
  for (int k = 0; k < n_in; k++)
    if (!d_in[k]) return fail(c, MDC_ERR_ARG, "mdc_tune_placement_device: input candidate %d is NULL", k);
  for (int k = 0; k < n_out; k++)
    if (!d_out[k]) return fail(c, MDC_ERR_ARG, "mdc_tune_placement_device: output candidate %d is NULL", k);
  ReadLock lk(c->mu);
  DeviceGuard dg(c->device);
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t e0, e1;
  MDC_HIP(c, hipEventCreate(&e0));
  if (hipEventCreate(&e1) != hipSuccess) {
    (void)hipEventDestroy(e0);
    return fail(c, MDC_ERR_HIP, "hipEventCreate failed");
  }
  int rc = MDC_OK, bi = 0, bo = 0;
  float best = 1e30f;
  for (int i = 0; i < n_in && rc == MDC_OK; i++)
    for (int j = 0; j < n_out && rc == MDC_OK; j++) {
      float ms[5] = {0, 0, 0, 0, 0};
      for (int k = 0; k < 7 && rc == MDC_OK; k++) {  // 2 warm-up launches, 5 timed
        bool ok = k < 2 || hipEventRecord(e0, s) == hipSuccess;
        if (ok) rc = enqueue_process(c, d_in[i], d_out[j], nframes, flags, s);
        if (rc != MDC_OK) break;
        if (k >= 2) ok = ok && hipEventRecord(e1, s) == hipSuccess && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms[k - 2], e0, e1) == hipSuccess;
        if (!ok) rc = fail(c, MDC_ERR_HIP, "mdc_tune_placement_device: timing a launch failed");
      }
      if (rc != MDC_OK) break;
      std::sort(ms, ms + 5);
      if (ms_out) ms_out[(size_t)i * n_out + j] = ms[2];
      if (ms[2] < best) {
        best = ms[2];
        bi = i;
        bo = j;
      }
    }
  (void)hipStreamSynchronize(s);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *best_in = bi;
  *best_out = bo;
  return rc;
} MDC_CATCH(c)

int mdc_describe_launch(mdc_ctx* c, unsigned flags, int pyramid_levels, char* buf, size_t cap) try {
  if (!c || !buf || cap == 0) return MDC_ERR_ARG;
  ReadLock lk(c->mu);
  bool g, v, o;
  normalise(c, flags, g, v, o);
  char tmp[160];
  if (pyramid_levels == -1) {  // undistort<float> (mdc_undistort_batch_device_f32): its own plan, no LUT, no vignette
    if (!c->valid_remap) return fail(c, MDC_ERR_STATE, "no remap set (UndistorterFOV invalid)");
    const mdc_ctx::SrcPlan& p = c->plan[1];
    if (p.tiled && c->opt_kernel != MDC_KERNEL_GATHER)
      snprintf(tmp, sizeof tmp, "remap_tiled_kernel<false, %s, false, true, %d, %d, %d>", c->n_black > 0 ? "true" : "false", p.tile_w,
               tile_threads(p.tile_w, p.tile_h), p.nbuf);
    else snprintf(tmp, sizeof tmp, "remap_gather_f32_kernel");
  } else if (!(flags & MDC_RECTIFY)) {
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





// ---- host-pointer, many frames: copies and kernels overlapped ------------------------------






int mdc_jpeg_huffman_batch_device(mdc_ctx* c, const void* d_streams, int64_t stream_stride, void* d_records, int64_t record_bytes, int w, int h,
                                  int blocks_w, int blocks_rows, int64_t nframes, int* d_status, void* stream) try {
  if (!c) return MDC_ERR_ARG;
  if (!d_streams || !d_records || !d_status || nframes < 0 || w <= 0 || h <= 0) return fail(c, MDC_ERR_ARG, "mdc_jpeg_huffman_batch_device: bad argument");
  ReadLock lk(c->mu);
  DeviceGuard dg(c->device);
  // small batches spread a frame's stream over several workgroups, which talk through a scratch buffer: stream-ordered
  // allocation (nothing is shared between concurrent calls)
  void* scratch = nullptr;
  hipStream_t s = (hipStream_t)stream;
  if (jpeg_huffman_segments(nframes) > 1 && hipMallocAsync(&scratch, jpeg_huffman_scratch_bytes(nframes), s) != hipSuccess) {
    (void)hipGetLastError();
    scratch = nullptr;  // (no pool: one workgroup per frame, as for large batches)
  }
  const hipError_t e = launch_jpeg_huffman(d_streams, stream_stride, d_records, record_bytes, w, h, blocks_w, blocks_rows, nframes, d_status, s, 7u, scratch);
  if (scratch) (void)hipFreeAsync(scratch, s);
  MDC_HIP(c, e);
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
