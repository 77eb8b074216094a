// libmdc_hip.so: the pipelined many-frame host calls -- what this repository's DatasetReader::getImages hands its decoded
// (or still compressed) frames to.  One call walks its frames in chunks; per chunk an ingest stage that depends on what
// the caller brought, then the fused pass and the way out:
//
//   raw frames      (mdc_process_frames_host)        upload ------------------------------> fused pass -> results
//   JPEG records    (mdc_process_jpeg_frames_host)   upload -> inverse DCT ----------------> fused pass -> results
//   JPEG streams    (mdc_process_jpeg_streams_host)  upload (own stream, one chunk ahead) -> Huffman -> inverse DCT | fused pass -> results
//
// Raw frames and records: chunk k runs entirely on stream k % 2 (copies in, kernels, copies out, in order); the two streams
// overlap one chunk's copies with the other's kernels.  Streams: every chunk decodes on stream 0 and goes out on stream 1,
// tied by events, uploads run ahead on a third.  Page-locked caller memory mapped into the device is read / written in place
// (zero copy) when it lies in runs that make batched launches possible.
// On a failure the enqueue loop stops, ALL streams are drained (asynchronous copies into the caller's buffers may still be in
// flight) and only then the error is returned.
#include "mdc_ctx.h"

using namespace mdc;

namespace {

// `he` latches the first HIP failure of the enqueue loop; later calls are skipped
#define MDC_PIPE(call)                  \
  if (he == hipSuccess) {               \
    he = (call);                        \
    if (he != hipSuccess) what = #call; \
  }

struct PipelineCall {
  // ---- what the caller brought
  mdc_ctx* c;
  const char* who;
  const uint8_t* const* raw;
  const void* const* rec;
  const void* const* strm;
  const int64_t* strm_bytes;
  float* const* out;
  int64_t nframes;
  unsigned flags;
  int* status;
  int64_t record_bytes;
  int blocks_w, blocks_rows;
  const mdc_device_outputs* dev = nullptr;  // results stay on the device (the *_to_device calls): `out` is not used
  const int64_t* dev_index = nullptr;       // frame i -> position in dev's arrays (nullptr: i)
  // ---- geometry
  int iw = 0, ih = 0;
  size_t n_in = 0, n_out = 0, strm_stride = 0;
  // ---- mode
  static constexpr int kChunk = 16;  // frames per slot of the copy pipeline: one kernel launch (two with the inverse DCT), 2 x 16 async copies
  std::vector<float*> z_out;         // the device's view of the caller's images / frames (zero copy), when every one has one
  std::vector<const uint8_t*> z_in;
  bool zc_in = false, zc_out = false;
  int chunk = kChunk;
  int* d_host_status = nullptr;  // the device's view of the page-locked landing buffer of the status words
  // ---- state of the enqueue loop
  int rc = MDC_OK;
  hipError_t he = hipSuccess;
  const char* what = "";
  // ---- MDC_PIPE_TRACE: where a pipelined call spends its time (stderr)
  bool trace = false;
  std::chrono::steady_clock::time_point t_begin;
  std::vector<hipEvent_t> tev;  // per chunk 6 stamps [upload: start, done; decode stream: Huffman done, decoded; output stream: start, done]
  double t_views = 0, t_enqueued = 0;

  double since() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); }
  void stamp(int64_t kk, int j, hipStream_t st) {
    if (!trace) return;
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) == hipSuccess) {
      (void)hipEventRecord(e, st);
      tev[(size_t)kk * 6 + j] = e;
    }
  }

  // Arguments and geometry.  Streams are decoded into records of the reader's geometry (block grid rounded up to multiples of 12).
  int validate() {
    if (nframes < 0 || (nframes > 0 && ((!raw && !rec && !strm) || (!out && !dev) || (strm && !strm_bytes)))) return fail(c, MDC_ERR_ARG, "%s: bad argument", who);
    if (dev) {
      if (!dev->base || dev->levels < 1 || dev->levels > 4) return fail(c, MDC_ERR_ARG, "%s: device outputs need a base array and 1..4 levels", who);
      const bool grads = dev->dI[0] != nullptr;
      for (int l = 0; l < dev->levels; l++)
        if ((l && !dev->level[l - 1]) || (grads && (!dev->dI[l] || !dev->abs_squared_grad[l]))) return fail(c, MDC_ERR_ARG, "%s: level %d has a NULL device array", who, l);
      for (int64_t i = 0; i < nframes && dev_index; i++)
        if (dev_index[i] < 0) return fail(c, MDC_ERR_ARG, "%s: negative frame index", who);
    }
    const bool rect = (flags & MDC_RECTIFY) != 0;
    if (rect && !c->valid_remap) return fail(c, MDC_ERR_STATE, "no remap set (UndistorterFOV invalid)");
    iw = (rect || c->in_w <= 0) ? c->rm_in_w : c->in_w;
    ih = (rect || c->in_h <= 0) ? c->rm_in_h : c->in_h;
    if (iw <= 0 || ih <= 0) return fail(c, MDC_ERR_STATE, "frame size unknown");
    n_in = (size_t)iw * ih;
    n_out = rect ? (size_t)c->out_w * c->out_h : n_in;
    if (strm) {
      blocks_w = ((iw + 7) / 8 + 11) / 12 * 12;  // (MCUs are 1..4 x 1..4 luma blocks: a multiple of 12 holds every MCU-padded luma grid)
      blocks_rows = ((ih + 7) / 8 + 11) / 12 * 12;
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
      if (!(strm ? strm[i] : rec ? rec[i] : (const void*)raw[i]) || (!dev && !out[i]))
        return fail(c, MDC_ERR_ARG, "%s: frame %lld has a NULL buffer", who, (long long)i);
    if (status)
      for (int64_t i = 0; i < nframes; i++) status[i] = 0;
    return MDC_OK;
  }

  // Streams: the decode status words come back asynchronously into a page-locked landing buffer (a copy would queue behind
  // the results going out); the kernel writes them there directly when the device can see the buffer.
  int prepare_status() {
    if (!(strm && status)) return MDC_OK;
    if (c->pipe_status_cap < (size_t)nframes) {
      if (c->h_pipe_status) (void)hipHostFree(c->h_pipe_status);
      c->h_pipe_status = nullptr;
      c->pipe_status_cap = 0;
      const size_t cap = std::max<size_t>(256, (size_t)nframes * 2);
      MDC_HIP(c, hipHostMalloc((void**)&c->h_pipe_status, cap * sizeof(int), hipHostMallocDefault));
      c->pipe_status_cap = cap;
    }
    if (c->h_pipe_status && hipHostGetDevicePointer((void**)&d_host_status, c->h_pipe_status, 0) != hipSuccess) {
      (void)hipGetLastError();
      d_host_status = nullptr;
    }
    return MDC_OK;
  }

  // Zero copy (device_view): results go straight into the caller's images when every one of them is mapped page-locked
  // memory, frames are read straight from the caller's buffers when every one of them is (coefficient records are always
  // copied: the inverse DCT reads a record 16 bytes at a time per thread, uncached that would cross PCIe several times).
  // Streams: results are staged on the device and leave with ONE copy per run of images that lie back to back (the reader's pool
  // hands out slabs): while a kernel writes results across PCIe itself, the kernels of other streams make no progress -- the
  // Huffman launch of the next chunk finished 0.8 ms (its own time) after the output of the current one, however few
  // workgroups the output launch had --, the copy engine moves the same bytes at the same 52 GB/s and leaves the CUs alone
  // (256 frames: 10.4 -> 7.5 ms, profiles/r03_experiments/20_*).
  void choose_mode() {
    z_out.assign((size_t)nframes, nullptr);
    z_in.assign((rec || strm) ? 0 : (size_t)nframes, nullptr);
    if (dev) {  // results stay in HBM: nothing to map, frames go up by copy; chunks of 64 frames (MDC_PIPE_DEV_CHUNK: 16..256) -- the
      // way out costs nothing now, and the Huffman launch's time per frame falls with the frames per launch (6.7 us at 64, 4.1 at 256)
      zc_in = zc_out = false;
      static const int env_chunk = [] {
        const char* e = getenv("MDC_PIPE_DEV_CHUNK");
        return e ? std::max(16, std::min(256, atoi(e))) : 0;
      }();
      // the caller's option, else the environment, else the context's hint (MDC_OPT_DEVICE_PIPELINE_CHUNK_HINT), else 64
      chunk = c->opt_dev_chunk ? c->opt_dev_chunk : env_chunk ? env_chunk : c->opt_dev_chunk_hint ? c->opt_dev_chunk_hint : 64;
      return;
    }
    zc_out = nframes > 0 && !strm;
    zc_in = !rec && !strm && nframes > 0;
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
    // frames per slot.  Streams: 64 -- the Huffman kernel's time does not depend on the frame count up to ~64 (one workgroup per
    // frame, 1.3 ms), so small chunks would only repeat that latency; nothing staged: the chunk only alternates the streams
    chunk = (zc_in && zc_out) ? 64 : (strm ? 64 : kChunk);
  }

  // Two chunk slots of staging buffers, the two streams (+ the upload stream of the JPEG-stream mode) and their events.
  int ensure_buffers() {
    const size_t in_need = zc_in ? 0 : chunk * n_in, out_need = (zc_out || dev) ? 0 : chunk * n_out * sizeof(float);
    const size_t rec_need = (rec || strm) ? (size_t)chunk * (size_t)record_bytes : 0;
    const size_t strm_need = strm ? (size_t)chunk * strm_stride : 0;
    const int chunk_need = strm ? std::max(64, chunk) : 0;
    if (c->pipe_in_cap < in_need || c->pipe_out_cap < out_need || c->pipe_rec_cap < rec_need || c->pipe_strm_cap < strm_need || c->pipe_chunk_cap < chunk_need ||
        !c->pipe_stream[0] || !c->pipe_stream[1]) {
      const size_t in_cap = std::max(in_need, c->pipe_in_cap), out_cap = std::max(out_need, c->pipe_out_cap), rec_cap = std::max(rec_need, c->pipe_rec_cap);
      const size_t strm_cap = std::max(strm_need + strm_need / 4, c->pipe_strm_cap);  // (stream sizes vary from call to call: some headroom)
      const int chunk_cap = std::max(std::max(64, chunk_need), c->pipe_chunk_cap);
      c->pipe_in_cap = c->pipe_out_cap = c->pipe_rec_cap = c->pipe_strm_cap = 0;  // a failure part-way leaves "no slots", not stale capacities
      c->pipe_chunk_cap = 0;
      for (int k = 0; k < 2; k++) {
        if (c->pipe_stream[k]) MDC_HIP(c, hipStreamSynchronize(c->pipe_stream[k]));
        if (!c->pipe_stream[k]) MDC_HIP(c, hipStreamCreateWithFlags(&c->pipe_stream[k], hipStreamNonBlocking));
        if (!c->pipe_done[k]) MDC_HIP(c, hipEventCreateWithFlags(&c->pipe_done[k], hipEventDisableTiming));
        if (!c->pipe_dec[k]) MDC_HIP(c, hipEventCreateWithFlags(&c->pipe_dec[k], hipEventDisableTiming));
        for (void** p : {(void**)&c->d_pipe_in[k], (void**)&c->d_pipe_out[k], &c->d_pipe_rec[k], &c->d_pipe_strm[k], (void**)&c->d_pipe_status[k], &c->d_pipe_seg[k]})
          if (*p) {
            (void)hipFree(*p);
            *p = nullptr;
          }
        if (in_cap) MDC_HIP(c, hipMalloc(&c->d_pipe_in[k], in_cap));
        if (out_cap) MDC_HIP(c, hipMalloc(&c->d_pipe_out[k], out_cap));
        if (rec_cap) MDC_HIP(c, hipMalloc(&c->d_pipe_rec[k], rec_cap));
        if (strm_cap) MDC_HIP(c, hipMalloc(&c->d_pipe_strm[k], strm_cap));
        if (strm_cap) MDC_HIP(c, hipMalloc((void**)&c->d_pipe_status[k], (size_t)chunk_cap * sizeof(int)));
        if (strm_cap) MDC_HIP(c, hipMalloc(&c->d_pipe_seg[k], jpeg_huffman_scratch_bytes(chunk_cap)));
      }
      c->pipe_chunk_cap = strm_cap ? chunk_cap : 0;
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
    return MDC_OK;
  }

  // n buffers of the caller -> d_dst, d_stride apart.  A chunk whose sources lie at one stride in host memory (the reader's
  // ring) goes up as ONE strided copy: 64 separate copies of a 270-KB stream cost the decode stream ~1 ms of the ~2.5 ms a
  // chunk takes.
  void upload(int64_t f0, int n, void* d_dst, size_t d_stride, const void* const* src, const int64_t* bytes, size_t fixed_bytes, hipStream_t s) {
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
    const size_t last = bytes ? (size_t)bytes[f0 + n - 1] : fixed_bytes;
    if (n > 1 && pitch >= (ptrdiff_t)width && width <= d_stride && (rows_full || one_host_allocation(src[f0], (size_t)pitch * (size_t)(n - 1) + last))) {
      // rows of `width` bytes: a shorter source is followed by the next one within the pitch, except the LAST -- it goes up
      // with its own size (nothing is read beyond the end of the caller's last buffer)
      const int rows2d = last == width ? n : n - 1;
      MDC_PIPE(hipMemcpy2DAsync(d_dst, d_stride, src[f0], (size_t)pitch, width, (size_t)rows2d, hipMemcpyHostToDevice, s));
      if (rows2d < n) MDC_PIPE(hipMemcpyAsync((char*)d_dst + (size_t)(n - 1) * d_stride, src[f0 + n - 1], last, hipMemcpyHostToDevice, s));
    } else {
      for (int i = 0; i < n; i++)
        MDC_PIPE(hipMemcpyAsync((char*)d_dst + (size_t)i * d_stride, src[f0 + i], bytes ? (size_t)bytes[f0 + i] : fixed_bytes, hipMemcpyHostToDevice, s));
    }
  }

  // Ingest, JPEG streams.  Uploads run ahead on their own stream: chunk k+1's streams are ENQUEUED before anything of chunk k
  // (the copy queues work in submission order -- an upload submitted after chunk k's copy out would wait behind it) and go
  // up as soon as the Huffman launch of chunk k-1 has read the buffer, i.e. under the decode of chunk k and the output of
  // chunk k-1.
  void enqueue_stream_upload(int64_t kk) {
    const int sl = (int)(kk & 1);
    const int64_t uf0 = kk * chunk;
    const int un = (int)std::min<int64_t>(chunk, nframes - uf0);
    hipStream_t up = c->pipe_up_stream;
    if (kk >= 2) MDC_PIPE(hipStreamWaitEvent(up, c->pipe_huff[sl], 0));
    stamp(kk, 0, up);
    upload(uf0, un, c->d_pipe_strm[sl], strm_stride, strm, strm_bytes, 0, up);
    MDC_PIPE(hipEventRecord(c->pipe_up[sl], up));
    stamp(kk, 1, up);
  }
  void ingest_streams(int64_t k, int64_t f0, int n, int slot, hipStream_t s) {
    if (k == 0) enqueue_stream_upload(0);
    if (f0 + n < nframes) enqueue_stream_upload(k + 1);
    MDC_PIPE(hipStreamWaitEvent(s, c->pipe_up[slot], 0));
    // the status words land in page-locked host memory directly (a copy would queue behind the results going out)
    int* d_status = (status && d_host_status) ? d_host_status + f0 : c->d_pipe_status[slot];
    // which decode kernels this chunk needs (the headers are the caller's host memory: one component / three / restart intervals)
    unsigned kinds = 0;
    for (int i = 0; i < n; i++) {
      const mdc_jpeg_stream_header* hd = static_cast<const mdc_jpeg_stream_header*>(strm[f0 + i]);
      kinds |= hd->restart_interval ? 4u : ((hd->comp_info & 255u) == 3 ? 2u : 1u);
    }
    // (a chunk of <= 64 frames: each frame's stream goes over several workgroups, through the slot's segment states)
    MDC_PIPE(launch_jpeg_huffman(c->d_pipe_strm[slot], (int64_t)strm_stride, c->d_pipe_rec[slot], record_bytes, iw, ih, blocks_w, blocks_rows, n, d_status, s,
                                 kinds, c->d_pipe_seg[slot]));
    MDC_PIPE(hipEventRecord(c->pipe_huff[slot], s));
    stamp(k, 2, s);
    if (status && !d_host_status)
      MDC_PIPE(hipMemcpyAsync(c->h_pipe_status + f0, c->d_pipe_status[slot], (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s));
    MDC_PIPE(launch_jpeg_idct(c->d_pipe_rec[slot], record_bytes, c->d_pipe_in[slot], iw, ih, blocks_w, blocks_rows, n, s));
  }
  // Ingest, JPEG coefficient records: the host decoded the Huffman layer, the device dequantises and runs the inverse DCT.
  void ingest_records(int64_t k, int64_t f0, int n, int slot, hipStream_t s) {
    stamp(k, 0, s);
    upload(f0, n, c->d_pipe_rec[slot], (size_t)record_bytes, rec, nullptr, (size_t)record_bytes, s);
    stamp(k, 1, s);
    MDC_PIPE(launch_jpeg_idct(c->d_pipe_rec[slot], record_bytes, c->d_pipe_in[slot], iw, ih, blocks_w, blocks_rows, n, s));
  }
  // Ingest, raw 8-bit frames: a copy, or nothing at all when the kernels read the caller's buffers in place.
  void ingest_raw(int64_t k, int64_t f0, int n, int slot, hipStream_t s) {
    stamp(k, 0, s);
    if (!zc_in) upload(f0, n, c->d_pipe_in[slot], n_in, reinterpret_cast<const void* const*>(raw), nullptr, n_in, s);
    stamp(k, 1, s);
  }

  // The fused pass of one chunk and its way out, on s_out.
  void emit(int64_t f0, int n, int slot, hipStream_t s_out) {
    if (dev) {  // straight into the caller's device arrays: one launch set per run of frames with consecutive positions
      const bool rect = (flags & MDC_RECTIFY) != 0;
      const int w0 = rect ? c->out_w : iw, h0 = rect ? c->out_h : ih;
      const bool grads = dev->dI[0] != nullptr;
      for (int i = 0; i < n && rc == MDC_OK;) {
        const int64_t pos = dev_index ? dev_index[f0 + i] : f0 + i;
        int run = 1;
        while (i + run < n && (dev_index ? dev_index[f0 + i + run] : f0 + i + run) == pos + run) run++;
        const uint8_t* src = c->d_pipe_in[slot] + (size_t)i * n_in;
        float* base = dev->base + (size_t)pos * n_out;
        if (dev->levels == 1 && !grads) {
          rc = enqueue_process(c, src, base, run, flags, s_out);
        } else {
          float *lv[3] = {nullptr, nullptr, nullptr}, *gi[4], *ga[4];
          for (int l = 0; l < dev->levels; l++) {
            const size_t npl = (size_t)(w0 >> l) * (size_t)(h0 >> l);
            if (l) lv[l - 1] = dev->level[l - 1] + (size_t)pos * npl;
            gi[l] = grads ? dev->dI[l] + (size_t)pos * npl * 3 : nullptr;
            ga[l] = grads ? dev->abs_squared_grad[l] + (size_t)pos * npl : nullptr;
          }
          rc = enqueue_pyramid_gradients(c, src, base, dev->levels, lv, grads ? gi : nullptr, grads ? ga : nullptr, run, flags, 0, s_out);
        }
        i += run;
      }
      return;
    }
    if (zc_out) {  // in place: one launch per run of frames that lie back to back on both sides
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
      return;
    }
    rc = enqueue_process(c, c->d_pipe_in[slot], c->d_pipe_out[slot], n, flags, s_out);
    if (rc != MDC_OK) return;
    for (int i = 0; i < n;) {  // one copy per run of images that lie back to back in the caller's memory
      int run = 1;
      while (i + run < n && out[f0 + i + run] == out[f0 + i] + (size_t)run * n_out) run++;
      MDC_PIPE(hipMemcpyAsync(out[f0 + i], c->d_pipe_out[slot] + (size_t)i * n_out, (size_t)run * n_out * sizeof(float), hipMemcpyDeviceToHost, s_out));
      i += run;
    }
  }

  void enqueue_all() {
    const int64_t nchunks = (nframes + chunk - 1) / chunk;
    tev.assign(trace ? (size_t)nchunks * 6 : 0, (hipEvent_t) nullptr);
    int n = 0;
    for (int64_t f0 = 0, k = 0; f0 < nframes && rc == MDC_OK && he == hipSuccess; f0 += n, k++) {
      const int slot = (int)(k & 1);
      // Streams: ALL chunks decode on stream 0 (upload, Huffman kernel, inverse DCT) and go out on stream 1 (fused pass into the
      // caller's images), tied by events -- the Huffman kernel takes ~1.3 ms whatever the frame count (one workgroup per frame)
      // and the output is PCIe-bound, so chunk k+1 decodes while chunk k goes out.  Otherwise chunk k runs on stream k % 2.
      // (a smaller first chunk, to get the output going earlier, is slower: the Huffman launch takes ~0.75 ms whatever its frame
      // count, so more chunks only lengthen the decode stream -- profiles/r03_experiments/19_*)
      hipStream_t s = strm ? c->pipe_stream[0] : c->pipe_stream[slot];
      hipStream_t s_out = strm ? c->pipe_stream[1] : s;
      n = (int)std::min<int64_t>(chunk, nframes - f0);
      if (k >= 2 && strm) {  // the slot's staging is free again: in stream order ...
        MDC_PIPE(hipStreamWaitEvent(s, c->pipe_done[slot], 0));
      } else if (k >= 2 && !(zc_in && zc_out)) {  // ... or on the host
        MDC_PIPE(hipEventSynchronize(c->pipe_done[slot]));
      }
      if (strm) ingest_streams(k, f0, n, slot, s);
      else if (rec) ingest_records(k, f0, n, slot, s);
      else ingest_raw(k, f0, n, slot, s);
      if (he != hipSuccess) break;
      stamp(k, 3, s);
      if (strm) {
        MDC_PIPE(hipEventRecord(c->pipe_dec[slot], s));
        MDC_PIPE(hipStreamWaitEvent(s_out, c->pipe_dec[slot], 0));
        if (he != hipSuccess) break;
      }
      stamp(k, 4, s_out);
      emit(f0, n, slot, s_out);
      if (rc != MDC_OK) break;
      MDC_PIPE(hipEventRecord(c->pipe_done[slot], s_out));
      stamp(k, 5, s_out);
    }
    t_enqueued = since();
  }

  // Drain every stream the call touched -- also, and above all, after a failure --, report, hand the status words over.
  int finish() {
    auto drain = [&](hipStream_t st, const char* name) {
      const hipError_t e = hipStreamSynchronize(st);
      if (he == hipSuccess && e != hipSuccess) {
        he = e;
        what = name;
      }
    };
    if (strm && c->pipe_up_stream) drain(c->pipe_up_stream, "hipStreamSynchronize(pipe_up_stream)");
    for (int k = 0; k < 2; k++)
      if (c->pipe_stream[k]) drain(c->pipe_stream[k], "hipStreamSynchronize(pipe_stream)");
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

  int run() {
    static const bool trace_env = getenv("MDC_PIPE_TRACE") != nullptr;
    trace = trace_env;
    t_begin = std::chrono::steady_clock::now();
    int r = validate();
    if (r != MDC_OK) return r;
    if ((r = prepare_status()) != MDC_OK) return r;
    choose_mode();
    t_views = since();
    // (a failure while the staging is (re)made happens before anything of THIS call is in flight; ensure_buffers itself
    // synchronises the streams it re-uses)
    if ((r = ensure_buffers()) != MDC_OK) return r;
    enqueue_all();
    return finish();
  }
};
#undef MDC_PIPE

int process_frames_pipeline(mdc_ctx* c, const uint8_t* const* raw, const void* const* rec, int64_t record_bytes, int blocks_w, int blocks_rows,
                            float* const* out, int64_t nframes, unsigned flags, const char* who, const void* const* strm = nullptr,
                            const int64_t* strm_bytes = nullptr, int* status = nullptr, const mdc_device_outputs* dev = nullptr,
                            const int64_t* dev_index = nullptr) {
  if (!c) return MDC_ERR_ARG;
  ReadLock lk(c->mu);
  std::lock_guard<std::mutex> pipe_lk(c->pipe_mu);  // one pipelined call at a time per context (it overlaps internally)
  DeviceGuard dg(c->device);
  PipelineCall p{c, who, raw, rec, strm, strm_bytes, out, nframes, flags, status, record_bytes, blocks_w, blocks_rows, dev, dev_index};
  return p.run();
}

}  // namespace

extern "C" {

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

int mdc_process_frames_host_to_device(mdc_ctx* c, const uint8_t* const* raw, int64_t nframes, unsigned flags, const mdc_device_outputs* out,
                                      const int64_t* frame_index) try {
  if (c && !out) return fail(c, MDC_ERR_ARG, "mdc_process_frames_host_to_device: no device outputs");
  return process_frames_pipeline(c, raw, nullptr, 0, 0, 0, nullptr, nframes, flags, "mdc_process_frames_host_to_device", nullptr, nullptr, nullptr, out, frame_index);
} MDC_CATCH(c)

int mdc_process_jpeg_frames_host_to_device(mdc_ctx* c, const void* const* records, int64_t record_bytes, int blocks_w, int blocks_rows, int64_t nframes,
                                           unsigned flags, const mdc_device_outputs* out, const int64_t* frame_index) try {
  if (c && !out) return fail(c, MDC_ERR_ARG, "mdc_process_jpeg_frames_host_to_device: no device outputs");
  return process_frames_pipeline(c, nullptr, records, record_bytes, blocks_w, blocks_rows, nullptr, nframes, flags, "mdc_process_jpeg_frames_host_to_device", nullptr,
                                 nullptr, nullptr, out, frame_index);
} MDC_CATCH(c)

int mdc_process_jpeg_streams_host_to_device(mdc_ctx* c, const void* const* streams, const int64_t* stream_bytes, int64_t nframes, unsigned flags,
                                            const mdc_device_outputs* out, const int64_t* frame_index, int* status) try {
  if (c && !out) return fail(c, MDC_ERR_ARG, "mdc_process_jpeg_streams_host_to_device: no device outputs");
  return process_frames_pipeline(c, nullptr, nullptr, 0, 0, 0, nullptr, nframes, flags, "mdc_process_jpeg_streams_host_to_device", streams, stream_bytes, status, out,
                                 frame_index);
} MDC_CATCH(c)

int mdc_device_alloc(mdc_ctx* c, size_t bytes, void** d_ptr) try {
  if (!c) return MDC_ERR_ARG;
  if (!d_ptr) return fail(c, MDC_ERR_ARG, "mdc_device_alloc: bad argument");
  *d_ptr = nullptr;
  DeviceGuard dg(c->device);
  // a sequence-sized buffer (a GiB or more: results of getImagesDevice, a caller's frame store) is striped over the device's memory
  // classes like the pairs of mdc_alloc_placed_device; small ones and devices without virtual memory
  // management: hipMalloc
  if (bytes >= ((size_t)1 << 30)) {
    mdc_striped_set set;
    const size_t one = bytes;
    if (mdc_alloc_striped_set_device(c, 1, &one, nullptr, &set) == MDC_OK) {
      if (set.strategy == MDC_PLACE_VMM) {
        std::lock_guard<std::mutex> lk(c->striped_mu);
        c->striped[set.d_ptr[0]] = set.handle;
        *d_ptr = set.d_ptr[0];
        return MDC_OK;
      }
      (void)mdc_free_striped_set_device(c, &set);  // (plain allocations: the ordinary path below owns those)
    }
  }
  MDC_HIP(c, hipMalloc(d_ptr, std::max<size_t>(bytes, 1)));
  return MDC_OK;
} MDC_CATCH(c)

void mdc_device_free(mdc_ctx* c, void* d_ptr) {
  if (!c || !d_ptr) return;
  DeviceGuard dg(c->device);
  void* arena = nullptr;
  {
    std::lock_guard<std::mutex> lk(c->striped_mu);
    auto it = c->striped.find(d_ptr);
    if (it != c->striped.end()) {
      arena = it->second;
      c->striped.erase(it);
    }
  }
  if (arena) {
    mdc_striped_set set;
    memset(&set, 0, sizeof set);
    set.handle = arena;
    (void)mdc_free_striped_set_device(c, &set);
    return;
  }
  (void)hipFree(d_ptr);
}

int mdc_copy_to_host(mdc_ctx* c, void* dst, const void* d_src, size_t bytes) try {
  if (!c) return MDC_ERR_ARG;
  if (bytes && (!dst || !d_src)) return fail(c, MDC_ERR_ARG, "mdc_copy_to_host: bad argument");
  DeviceGuard dg(c->device);
  MDC_HIP(c, hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
  return MDC_OK;
} MDC_CATCH(c)

}  // extern "C"
