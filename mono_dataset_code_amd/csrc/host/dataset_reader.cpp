// class DatasetReader (include/mono_dataset_code/BenchmarkDatasetReader.h): the reference's sequence
// reader (src/BenchmarkDatasetReader.h:83-345) re-built around the fused GPU pass.
//
//   listing / times.txt / log lines : as the reference (:86-148, :282-324)
//   decode                          : own decoders (image_codecs.cpp), folder or images.zip (zip_reader.cpp),
//                                     on a pool of worker threads, into page-locked buffers
//   getImage                        : one mdc_process_host call into a pooled page-locked ExposureImage
//   getImages                       : decode pool -> ring of page-locked chunks -> mdc_process_frames_host per chunk;
//                                     the pool decodes the next chunks while chunk k is on the GPU
#include "BenchmarkDatasetReader.h"

#include <exception>
#include <dirent.h>
#include <dlfcn.h>
#include <algorithm>
#include <cctype>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <mutex>
#include <thread>
#include <vector>

#include "MdcBind.h"
#include "host_device.h"
#include "image_codecs.h"
#include "image_codecs_internal.h"
#include "mdc_hip.h"
#include "zip_reader.h"

namespace {

// One decode request: frame `id` into `dst`.  Filled in by whoever decodes it (a pool worker or the
// calling thread); `done` is published under State::mu.
struct Decode {
  int id = -1;
  unsigned char* dst = 0;
  size_t cap = 0;
  int w = 0, h = 0;
  bool ok = false, done = true, busy = false;  // busy: queued or being decoded
  bool consumed = true;                        // prefetch cache: already handed to the caller (or never filled)
  // getImages with the GPU JPEG stage: a JPEG file is only Huffman-decoded, into a coefficient record at dst (pitch in blocks
  // as asked for); is_record tells what dst holds afterwards (other formats still decode to pixels)
  int want_record_pitch = 0;
  bool is_record = false;
  int rec_rows = 0;
  // ... or, with the Huffman decoding on the GPU as well, only unstuffed into a stream (mdc_jpeg_stream_header + bytes) at dst
  bool want_stream = false, is_stream = false;
  size_t stream_bytes = 0;
  std::string err;
  unsigned long stamp = 0;  // prefetch cache: age
};

struct HostBuffer {  // page-locked when a GPU is there, plain otherwise (decode works without a GPU)
  unsigned char* p = 0;
  bool pinned = false;
  void alloc(size_t n) {
    p = static_cast<unsigned char*>(mdc_host_alloc(n));
    pinned = p != 0;
    if (!p) p = static_cast<unsigned char*>(std::malloc(n));
  }
  void release() {
    if (!p) return;
    if (pinned) mdc_host_free(p);
    else std::free(p);
    p = 0;
  }
};

unsigned flag_word(bool rectify, bool g, bool v, bool o) {
  return (rectify ? MDC_RECTIFY : 0u) | (g ? MDC_GAMMA : 0u) | (v ? MDC_VIGNETTE : 0u) | (o ? MDC_KILL_OVEREXPOSED : 0u);
}

}  // namespace

struct DatasetReader::State {
  std::string path;
  bool zipped = false;
  mdc_host::ZipArchive zip;
  std::vector<std::string> files;
  std::vector<int> zip_index;
  std::vector<double> timestamps;
  std::vector<float> exposures;

  UndistorterFOV* fov = 0;
  PhotometricUndistorter* photo = 0;
  mdc_ctx* gpu = 0;
  int W = 0, H = 0, w = 0, h = 0;
  std::string err;

  // decode pool
  std::vector<std::thread> workers;
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  std::deque<Decode*> jobs;
  bool stop = false;
  int want_threads = 0;

  // prefetch cache of getImage / getImageRaw
  int prefetch = 16;
  std::vector<Decode> slots;
  std::vector<HostBuffer> slot_mem;
  int in_use = -1;  // slot whose buffer the caller holds (getImageRaw's promise)
  unsigned long clock = 0;
  long cache_hits = 0, cache_misses = 0;  // frames found decoded (or being decoded) ahead / decoded by the caller itself

  // ring of getImages: chunks of 32 (64 in JPEG stage 2) page-locked frame buffers, 256 in all.  Chunk k is on the GPU while
  // the pool decodes chunks k+1 .. (up to 192 frames in flight): a decode thread that is slow on one frame delays only the
  // chunk that frame is in, not the pipeline (two half-rings of 64 stalled on every straggler: 2.5-2.9 k frames/s)
  enum { kRingFrames = 256 };  // page-locked decode buffers of getImages (335 MB at 1280x1024, 670 MB in stage 1 and for calls of > 256 frames in stage 2; first getImages)
  // One lane per device the reader may use (MDC_DEVICES; lanes[0] is `gpu`): its context and its own decode ring --
  // ONE page-locked block, slot i at ring_block.p + i * ring_stride (a chunk's uploads are then one strided copy instead of
  // one copy per frame), ring_bytes per buffer (a frame, or a record when the GPU JPEG stage is on).
  struct Lane {
    mdc_ctx* gpu = 0;
    int device = -1;
    bool twin = false;  // a second context on lane 0's device, made for getImagesDevice (ensure_device_lanes); owned by the reader
    HostBuffer ring_block;
    size_t ring_stride = 0, ring_bytes = 0;
    int ring_slots = 0;
    long frames = 0;  // statistics over the reader's life: frames produced, seconds waiting for the decoders / inside GPU calls
    double t_wait = 0, t_gpu = 0;
  };
  struct LaneRun;
  std::vector<Lane> lanes;
  void* multi = 0;  // libmdc_multi.so's object when the lanes' contexts are its (RCCL table broadcast), else the lanes own theirs
  std::mutex err_mu, image_mu;

  size_t frame_bytes() const { return (size_t)W * H; }
  // GPU JPEG stage of getImages: JPEG frames travel as coefficient records (2 bytes per pixel + table), the inverse DCT runs on
  // the device.  Default on; MDC_GPU_JPEG=0 or setGpuJpeg(false) keeps the whole decode on the host.
  int gpu_jpeg = 2;  // 0: JPEG decoded on the host; 1: host Huffman + device inverse DCT; 2: device Huffman + inverse DCT
  // getImage on a JPEG sequence read in order: after two consecutive ids the next `lookahead` frames go through the getImages
  // pipeline (Huffman decoding on the device) with the caller's switches, and the following calls hand those results out
  int lookahead = kRingFrames;  // the most; a run starts with 64 and doubles per batch (the longer the call, the less its fill and drain weigh)
  int ahead_batch = 64;
  std::vector<ExposureImage*> ahead;
  int ahead_first = -1;
  unsigned ahead_flags = 0;
  int seq_last = -2, seq_run = 0;
  bool quiet_batch = false;  // the batch behind getImage's lookahead: a frame that fails is reported when the caller asks for it
  void drop_ahead() {
    for (ExposureImage* e : ahead) delete e;
    ahead.clear();
    ahead_first = -1;
    ahead_batch = 64;
  }
  bool is_jpeg_name(size_t id) const {
    const std::string& f = files[id];
    const size_t dot = f.rfind('.');
    if (dot == std::string::npos) return false;
    std::string ext = f.substr(dot + 1);
    for (char& ch : ext) ch = (char)std::tolower((unsigned char)ch);
    return ext == "jpg" || ext == "jpeg";
  }
  int rec_pitch = 0, rec_rows = 0;
  size_t rec_bytes = 0;

  // ---- devices ---------------------------------------------------------------------------------
  // Frames of a sequence are independent (reference src/BenchmarkDatasetReader.h:188-243), so getImages deals its range to
  // every device listed in MDC_DEVICES in chunks (lane l takes chunks l, l + L, ...).  All lanes hold the SAME tables: with
  // libmdc_multi.so next to this library and distinct devices, rank 0's tables go out in one RCCL broadcast over xGMI
  // (mdc_multi_bcast_tables); otherwise (the library is missing, or a device is listed twice -- a test on a one-GPU box) every
  // context takes them from the host objects directly.  Either way the bytes are the host's.
  struct MultiApi {
    void* lib = 0;
    int (*create)(const int*, int, void**) = 0;
    void (*destroy)(void*) = 0;
    mdc_ctx* (*ctx)(void*, int) = 0;
    int (*bcast)(void*, int) = 0;
    const char* (*last_error)(const void*) = 0;
  } mapi;
  bool load_multi() {
    if (mapi.lib) return true;
    Dl_info info;
    std::string dir;
    if (dladdr((void*)&mdc_host::open_device_context, &info) && info.dli_fname) {
      dir = info.dli_fname;
      const size_t sl = dir.rfind('/');
      dir = sl == std::string::npos ? std::string() : dir.substr(0, sl + 1);
    }
    void* lib = dlopen((dir + "libmdc_multi.so").c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_NODELETE);  // RCCL keeps static state and helper threads: never unmapped again
    if (!lib) return false;
    mapi.create = (int (*)(const int*, int, void**))dlsym(lib, "mdc_multi_create");
    mapi.destroy = (void (*)(void*))dlsym(lib, "mdc_multi_destroy");
    mapi.ctx = (mdc_ctx * (*)(void*, int)) dlsym(lib, "mdc_multi_ctx");
    mapi.bcast = (int (*)(void*, int))dlsym(lib, "mdc_multi_bcast_tables");
    mapi.last_error = (const char* (*)(const void*))dlsym(lib, "mdc_multi_last_error");
    if (!mapi.create || !mapi.destroy || !mapi.ctx || !mapi.bcast || !mapi.last_error) {
      dlclose(lib);
      return false;
    }
    mapi.lib = lib;
    return true;
  }
  static std::vector<int> device_list() {
    std::vector<int> devs;
    const char* e = std::getenv("MDC_DEVICES");
    if (!e || !*e) return devs;
    if (std::string(e) == "all") {
      const int n = mdc_device_count();
      for (int i = 0; i < n; i++) devs.push_back(i);
      return devs;
    }
    for (const char* p = e; *p;) {
      char* end = 0;
      const long v = std::strtol(p, &end, 10);
      if (end == p) break;
      if (v >= 0) devs.push_back((int)v);
      p = *end == ',' ? end + 1 : end;
      if (*end && *end != ',') break;
    }
    return devs;
  }
  void open_devices() {
    const std::vector<int> devs = device_list();
    bool distinct = devs.size() > 1;
    for (size_t i = 0; i < devs.size(); i++)
      for (size_t j = i + 1; j < devs.size(); j++)
        if (devs[i] == devs[j]) distinct = false;
    // (MDC_READER_FORCE_RCCL=1: take the RCCL path for a single listed device too -- a world of one --, so that a one-GPU box
    // executes the dlopen, the communicator set-up, the broadcast and the lanes on libmdc_multi's contexts)
    const bool force_rccl = std::getenv("MDC_READER_FORCE_RCCL") != 0 && devs.size() == 1;
    if ((distinct || force_rccl) && load_multi()) {  // one RCCL broadcast of rank 0's tables
      void* m = 0;
      if (mapi.create(devs.data(), (int)devs.size(), &m) == MDC_OK && m) {
        mdc_ctx* root = mapi.ctx(m, 0);
        if (root && mdc_bind_objects(root, fov, photo) == MDC_OK && mapi.bcast(m, 0) == MDC_OK) {
          multi = m;
          for (size_t r = 0; r < devs.size(); r++) {
            Lane ln;
            ln.gpu = mapi.ctx(m, (int)r);
            ln.device = devs[r];
            lanes.push_back(ln);
          }
          gpu = lanes[0].gpu;
          std::printf("DatasetReader: %d devices, calibration tables broadcast over RCCL\n", (int)devs.size());
          return;
        }
        std::fprintf(stderr, "DatasetReader: RCCL table broadcast failed (%s); every device takes the tables from the host\n", mapi.last_error(m));
        mapi.destroy(m);
      }
    }
    if (devs.size() > 1) {
      for (size_t r = 0; r < devs.size(); r++) {
        mdc_ctx* c = 0;
        if (mdc_create(devs[r], &c) != MDC_OK || mdc_bind_objects(c, fov, photo) != MDC_OK) {
          std::fprintf(stderr, "DatasetReader: device %d: %s; not used\n", devs[r], mdc_last_error(c));
          if (c) mdc_destroy(c);
          continue;
        }
        Lane ln;
        ln.gpu = c;
        ln.device = devs[r];
        lanes.push_back(ln);
      }
      if (!lanes.empty()) {
        gpu = lanes[0].gpu;
        std::printf("DatasetReader: %d devices, calibration tables uploaded to each\n", (int)lanes.size());
        return;
      }
    }
    gpu = devs.size() == 1 ? 0 : mdc_host::open_device_context("DatasetReader");
    if (devs.size() == 1 && mdc_create(devs[0], &gpu) != MDC_OK) {
      std::fprintf(stderr, "DatasetReader: no GPU context on device %d (%s)\n", devs[0], mdc_last_error(0));
      gpu = 0;
    }
    if (gpu && mdc_bind_objects(gpu, fov, photo) != MDC_OK) {
      std::fprintf(stderr, "DatasetReader: table upload failed: %s\n", mdc_last_error(gpu));
      mdc_destroy(gpu);
      gpu = 0;
    }
    Lane ln;
    ln.gpu = gpu;
    mdc_info inf;
    ln.device = (gpu && mdc_get_info(gpu, &inf) == MDC_OK) ? inf.device : -1;
    lanes.push_back(ln);
  }
  // getImagesDevice: its results cross no bus on the way out, so what limits one pipelined call is its own fill and drain (upload of
  // the first chunk, fused pass of the last).  Two calls from two host threads on two contexts of the SAME device overlap them:
  // measured on a zipped 1280x1024 JPEG sequence 99.6 k frames/s with one lane, 126 k with two lanes and 128-frame chunks, 109 k with
  // three (profiles/r05_reader_device_rates.txt).  The twin is made at the first getImagesDevice call (MDC_DEVICE_LANES=1: never).
  int host_lanes = 0;  // lanes getImages deals its range to (what open_devices made)
  void ensure_device_lanes() {
    if (!host_lanes) host_lanes = (int)lanes.size();
    static const int want = [] {
      const char* e = std::getenv("MDC_DEVICE_LANES");
      return e ? std::max(1, std::min(4, std::atoi(e))) : 2;
    }();
    if (lanes.empty() || !lanes[0].gpu) return;
    int have = 0;
    for (const Lane& ln : lanes) have += ln.device == lanes[0].device && ln.gpu ? 1 : 0;
    bool made = false;
    for (; have < want; have++) {
      mdc_ctx* c = 0;
      if (mdc_create(lanes[0].device, &c) != MDC_OK || mdc_bind_objects(c, fov, photo) != MDC_OK) {
        if (c) mdc_destroy(c);
        return;  // one lane does the work
      }
      Lane ln;
      ln.gpu = c;
      ln.device = lanes[0].device;
      ln.twin = true;
      lanes.push_back(ln);
      made = true;
    }
    // with a second call to hide a chunk's fill and drain behind, longer chunks win (Huffman: 5.3 us per frame at 128, 6.7 at 64).  Given ONCE,
    // when a twin was made, and as a hint: a caller's own MDC_OPT_DEVICE_PIPELINE_CHUNK on the public context (getContext()) and
    // MDC_PIPE_DEV_CHUNK in the environment both stay in force
    if (made && have >= 2)
      for (Lane& ln : lanes)
        if (ln.device == lanes[0].device && ln.gpu) (void)mdc_set_option(ln.gpu, MDC_OPT_DEVICE_PIPELINE_CHUNK_HINT, 128);
  }
  void close_devices() {
    for (Lane& ln : lanes) {
      ln.ring_block.release();
      if ((!multi || ln.twin) && ln.gpu) mdc_destroy(ln.gpu);
    }
    if (multi) mapi.destroy(multi);
    multi = 0;
    lanes.clear();
    gpu = 0;
    // (no dlclose: libmdc_multi.so pulls in librccl, whose static state and helper threads must outlive this reader -- unloading it
    // mid-process risks a crash at exit or when the next reader loads it again; the handle is RTLD_NODELETE and simply dropped)
    mapi.lib = 0;
  }

  // ---- decoding (any thread) ------------------------------------------------------------------
  // Never throws: it runs in the decode pool's threads, where an escaping exception (bad_alloc on a corrupt size field,
  // ...) would terminate the process instead of reporting one bad frame.
  void decode_now(Decode& d) const {
    try {
      decode_unguarded(d);
    } catch (const std::exception& e) {
      d.ok = false;
      d.err = std::string("decode failed: ") + e.what();
    } catch (...) {
      d.ok = false;
      d.err = "decode failed";
    }
  }
  void decode_unguarded(Decode& d) const {
    static thread_local std::vector<unsigned char> bytes;  // per-thread scratch, keeps its capacity between frames
    d.ok = false;
    d.w = d.h = 0;
    if (d.id < 0 || d.id >= (int)files.size()) {
      d.err = "frame index out of range";
      return;
    }
    if (zipped) {
      if (!zip.read(zip_index[(size_t)d.id], bytes, &d.err)) return;
    } else if (!mdc_host::read_file(files[(size_t)d.id], bytes)) {
      d.err = "cannot read " + files[(size_t)d.id];
      return;
    }
    d.is_record = d.is_stream = false;
    const bool is_jpeg = bytes.size() > 4 && bytes[0] == 0xff && bytes[1] == 0xd8;
    if (d.want_stream && is_jpeg) {  // what the device decoder takes (grayscale baseline, no restart markers); else the record path
      std::string why;
      size_t used = 0;
      if (mdc_host::jpeg_stream(bytes.data(), bytes.size(), d.dst, d.cap, &used, &d.w, &d.h, &why)) {
        d.ok = d.is_stream = true;
        d.stream_bytes = used;
        return;
      }
    }
    if (d.want_record_pitch > 0 && is_jpeg && d.cap > 256) {
      mdc_host::JpegCoefSink sink;
      sink.coef = reinterpret_cast<int16_t*>(d.dst + 128);
      sink.cap_blocks = (d.cap - 128) / 128;
      sink.pitch_blocks = d.want_record_pitch;
      d.ok = mdc_host::decode_jpeg_coefs(bytes.data(), bytes.size(), &sink, &d.err);
      if (d.ok) {
        std::memcpy(d.dst, sink.quant, 128);
        d.w = sink.w;
        d.h = sink.h;
        d.rec_rows = sink.blocks_rows;
        d.is_record = true;
      } else {
        // a file whose blocks do not fit the record geometry (or that the coefficient path refuses for any other reason)
        // still decodes to pixels on the host, so that stage 1 gives the same images and the same failures as stages 0 and 2
        std::string e2;
        d.ok = mdc_host::decode_gray8(bytes.data(), bytes.size(), d.dst, d.cap, &d.w, &d.h, &e2);
        if (!d.ok) d.err = e2;
      }
    } else {
      d.ok = mdc_host::decode_gray8(bytes.data(), bytes.size(), d.dst, d.cap, &d.w, &d.h, &d.err);
    }
    if (!d.ok) d.err = files[(size_t)d.id] + ": " + d.err;
  }

  void worker() {
    for (;;) {
      Decode* d = 0;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_job.wait(lk, [&] { return stop || !jobs.empty(); });
        if (stop && jobs.empty()) return;
        d = jobs.front();
        jobs.pop_front();
      }
      decode_now(*d);
      {
        std::lock_guard<std::mutex> lk(mu);
        d->busy = false;
        d->done = true;
      }
      cv_done.notify_all();
    }
  }

  // CPUs this process may actually use: the hardware threads, cut down to the container's CFS quota if there is one
  // (cgroup v2 cpu.max / v1 cpu.cfs_quota_us).  More decode threads than that only get throttled -- together with the
  // HIP runtime's own threads (a box with 256 hardware threads and a 16-CPU quota decodes fastest with 16).
  static int usable_cpus() {
    unsigned hw = std::thread::hardware_concurrency();
    if (!hw) hw = 4;
    double quota = 0, period = 0;
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char q[64];
      if (std::fscanf(f, "%63s %lf", q, &period) == 2 && std::strcmp(q, "max") != 0) quota = std::atof(q);
      std::fclose(f);
    } else if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
      if (std::fscanf(g, "%lf", &quota) != 1) quota = 0;
      std::fclose(g);
      if (FILE* h = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
        if (std::fscanf(h, "%lf", &period) != 1) period = 0;
        std::fclose(h);
      }
    }
    if (quota > 0 && period > 0) hw = std::min<unsigned>(hw, std::max(1u, (unsigned)(quota / period + 0.5)));
    return (int)hw;
  }
  int thread_count() const {
    if (want_threads > 0) return want_threads;
    return std::max(1, std::min(usable_cpus(), 64));
  }
  void start_pool() {
    if (!workers.empty()) return;
    const int n = thread_count();
    for (int i = 0; i < n; i++) workers.emplace_back(&State::worker, this);
  }
  void stop_pool() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv_job.notify_all();
    for (auto& t : workers) t.join();
    workers.clear();
    stop = false;
  }
  void submit(Decode* d) {  // mu held by the caller
    d->busy = true;
    d->done = false;
    jobs.push_back(d);
  }

  // ---- prefetch cache ---------------------------------------------------------------------------
  void ensure_slots() {
    const size_t want = (size_t)std::max(prefetch, 0) + 2;
    if (slots.size() == want) return;
    drain();
    for (auto& m : slot_mem) m.release();
    slots.assign(want, Decode());
    slot_mem.assign(want, HostBuffer());
    for (size_t i = 0; i < want; i++) {
      slot_mem[i].alloc(frame_bytes());
      slots[i].dst = slot_mem[i].p;
      slots[i].cap = frame_bytes();
    }
    in_use = -1;
  }
  void drain() {  // wait for every queued decode
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [&] {
      for (auto& s : slots)
        if (s.busy) return false;
      return jobs.empty();
    });
  }
  // mu held: a slot that is neither being decoded nor lent to the caller and holds nothing of value -- empty, or a
  // frame the caller has already had (oldest first).  Frames decoded ahead and not yet asked for are never
  // evicted for another prefetch (force: the caller itself needs a slot -- then the one farthest ahead goes).
  int free_slot(bool force = false) {
    int best = -1;
    for (size_t i = 0; i < slots.size(); i++) {
      if (slots[i].busy || (int)i == in_use) continue;
      if (slots[i].id < 0) return (int)i;
      if (!slots[i].consumed) continue;
      if (best < 0 || slots[i].stamp < slots[(size_t)best].stamp) best = (int)i;
    }
    if (best < 0 && force)
      for (size_t i = 0; i < slots.size(); i++)
        if (!slots[i].busy && (int)i != in_use && (best < 0 || slots[i].id > slots[(size_t)best].id)) best = (int)i;
    return best;
  }
  // The decoded frame `id` (from the cache, or decoded here), then the next frames are queued.
  Decode* fetch(int id) {
    ensure_slots();
    int k = -1;
    {
      std::unique_lock<std::mutex> lk(mu);
      for (size_t i = 0; i < slots.size(); i++)
        if (slots[i].id == id) k = (int)i;
      if (k >= 0) {
        cache_hits++;
        cv_done.wait(lk, [&] { return slots[(size_t)k].done; });
      } else {
        cache_misses++;
        in_use = -1;
        k = free_slot(true);
        if (k < 0) {  // every other slot is being decoded into: wait for one
          cv_done.wait(lk, [&] { return (k = free_slot(true)) >= 0; });
        }
        slots[(size_t)k].id = id;
        slots[(size_t)k].done = false;
      }
      in_use = k;
      slots[(size_t)k].consumed = true;
      slots[(size_t)k].stamp = ++clock;
    }
    Decode& d = slots[(size_t)k];
    if (!d.done) {  // not in the cache: decode in this thread
      decode_now(d);
      std::lock_guard<std::mutex> lk(mu);
      d.done = true;
    }
    if (prefetch > 0 && (int)files.size() > 1) {
      start_pool();
      std::lock_guard<std::mutex> lk(mu);
      for (int a = 1; a <= prefetch && id + a < (int)files.size(); a++) {
        bool have = false;
        for (auto& s : slots)
          if (s.id == id + a) have = true;
        if (have) continue;
        const int f = free_slot();
        if (f < 0) break;
        slots[(size_t)f].id = id + a;
        slots[(size_t)f].consumed = false;
        slots[(size_t)f].stamp = ++clock;
        submit(&slots[(size_t)f]);
      }
      cv_job.notify_all();
    }
    return &d;
  }
};

namespace {

// name-sorted directory listing, full paths (reference getdir, :44-72)
void list_folder(const std::string& dir, std::vector<std::string>& files) {
  DIR* dp = opendir(dir.c_str());
  if (!dp) return;
  while (struct dirent* e = readdir(dp)) {
    const std::string name = e->d_name;
    if (name != "." && name != "..") files.push_back(name);
  }
  closedir(dp);
  std::sort(files.begin(), files.end());
  for (auto& f : files) f = dir + f;
}

}  // namespace

DatasetReader::DatasetReader(std::string folder) : s_(new State()) {
  if (const char* e = std::getenv("MDC_GPU_JPEG")) s_->gpu_jpeg = std::max(0, std::min(2, std::atoi(e)));
  if (const char* e = std::getenv("MDC_READER_LOOKAHEAD")) s_->lookahead = std::max(0, std::min((int)State::kRingFrames, std::atoi(e)));
  State& s = *s_;
  s.path = folder;
  list_folder(s.path + "images/", s.files);
  if (!s.files.empty()) {
    std::printf("Load Dataset %s: found %d files in folder /images; assuming that all images are there.\n", s.path.c_str(),
                (int)s.files.size());
  } else {
    std::printf("Load Dataset %s: found no in folder /images; assuming that images are zipped.\n", s.path.c_str());
    s.zipped = true;
    std::string zerr;
    if (!s.zip.open(s.path + "images.zip", &zerr)) {
      std::printf("ERROR %d reading archive %s!\n", 1, (s.path + "images.zip").c_str());
      std::fprintf(stderr, "DatasetReader: %s\n", zerr.c_str());
      std::exit(1);  // as the reference (:111-115): callers rely on never seeing a reader without frames
    }
    std::vector<std::pair<std::string, int>> named;
    for (int k = 0; k < s.zip.entries(); k++) {
      const std::string& n = s.zip.name(k);
      if (n == "." || n == "..") continue;
      named.push_back(std::make_pair(n, k));
    }
    std::printf("got %d entries and %d files from zipfile!\n", s.zip.entries(), (int)named.size());
    std::sort(named.begin(), named.end());
    for (auto& nk : named) {
      s.files.push_back(nk.first);
      s.zip_index.push_back(nk.second);
    }
  }

  // times.txt: "id stamp exposure" or "id stamp" per line (:282-324)
  {
    std::ifstream tr((s.path + "times.txt").c_str());
    std::string line;
    while (tr.good() && std::getline(tr, line)) {
      int id;
      double stamp;
      float exposure = 0;
      if (3 == std::sscanf(line.c_str(), "%d %lf %f", &id, &stamp, &exposure)) {
        s.timestamps.push_back(stamp);
        s.exposures.push_back(exposure);
      } else if (2 == std::sscanf(line.c_str(), "%d %lf", &id, &stamp)) {
        s.timestamps.push_back(stamp);
        s.exposures.push_back(0);
      }
    }
    if (s.exposures.size() != s.files.size()) {
      std::printf("DatasetReader: Mismatch between number of images and number of timestamps / exposure times. Set all to zero.");
      s.timestamps.assign(s.files.size(), 0.0);
      s.exposures.assign(s.files.size(), 0.f);
    }
  }

  s.fov = new UndistorterFOV((s.path + "camera.txt").c_str());
  s.photo = new PhotometricUndistorter(s.path + "pcalib.txt", s.path + "vignette.png", s.fov->getInputDims()[0], s.fov->getInputDims()[1]);
  s.W = s.fov->getInputDims()[0];
  s.H = s.fov->getInputDims()[1];
  s.w = s.fov->getOutputDims()[0];
  s.h = s.fov->getOutputDims()[1];

  // one context holding BOTH objects' tables: the fused pass needs them together -- per device the reader may use
  // (MDC_DEVICES=all | 0,1,...; unset: the one device of $MDC_DEVICE / the calling thread, as before)
  s.open_devices();
  std::printf("Dataset %s: Got %d files!\n", s.path.c_str(), getNumImages());
}

DatasetReader::~DatasetReader() {
  State& s = *s_;
  s.stop_pool();
  for (auto& m : s.slot_mem) m.release();
  s.drop_ahead();
  s.close_devices();
  delete s.fov;
  delete s.photo;
  delete s_;
}

UndistorterFOV* DatasetReader::getUndistorter() { return s_->fov; }
PhotometricUndistorter* DatasetReader::getPhotoUndistorter() { return s_->photo; }
int DatasetReader::getNumImages() { return (int)s_->files.size(); }
double DatasetReader::getTimestamp(int id) { return (id < 0 || id >= (int)s_->timestamps.size()) ? 0 : s_->timestamps[(size_t)id]; }
float DatasetReader::getExposure(int id) { return (id < 0 || id >= (int)s_->exposures.size()) ? 0 : s_->exposures[(size_t)id]; }
const char* DatasetReader::lastError() const { return s_->err.c_str(); }
void DatasetReader::getPrefetchStats(long* hits, long* misses) const {
  if (hits) *hits = s_->cache_hits;
  if (misses) *misses = s_->cache_misses;
}

// Devices in use = the lanes getImages deals its range to (one per entry of MDC_DEVICES).  The twin contexts that getImagesDevice adds on a
// device it already has a lane on are not devices of their own: their counters are folded into that lane's.
int DatasetReader::getDeviceCount() const { return s_->host_lanes ? s_->host_lanes : (int)s_->lanes.size(); }
void DatasetReader::getDeviceStats(int lane, int* device, long* frames, double* decoder_wait_s, double* gpu_call_s) const {
  if (lane < 0 || lane >= getDeviceCount()) return;
  const State::Lane& ln = s_->lanes[(size_t)lane];
  long fr = ln.frames;
  double tw = ln.t_wait, tg = ln.t_gpu;
  bool first_of_device = true;
  for (int k = 0; k < lane; k++) first_of_device = first_of_device && s_->lanes[(size_t)k].device != ln.device;
  if (first_of_device)  // (MDC_DEVICES=0,0: two host lanes on one device -- the twins go to the first of them)
    for (size_t k = (size_t)getDeviceCount(); k < s_->lanes.size(); k++)
      if (s_->lanes[k].twin && s_->lanes[k].device == ln.device) {
        fr += s_->lanes[k].frames;
        tw += s_->lanes[k].t_wait;
        tg += s_->lanes[k].t_gpu;
      }
  if (device) *device = ln.device;
  if (frames) *frames = fr;
  if (decoder_wait_s) *decoder_wait_s = tw;
  if (gpu_call_s) *gpu_call_s = tg;
}

void DatasetReader::setDecodeThreads(int n) {
  State& s = *s_;
  if (n < 0) n = 0;
  if (n == s.want_threads) return;
  if (!s.slots.empty()) s.drain();
  s.stop_pool();
  s.want_threads = n;
}

void DatasetReader::setResultLookahead(int frames) {
  s_->lookahead = std::max(0, std::min(frames, (int)State::kRingFrames));
  if (!s_->lookahead) s_->drop_ahead();
}
void DatasetReader::setGpuJpeg(bool on) { s_->gpu_jpeg = on ? 2 : 0; }
void DatasetReader::setGpuJpegStage(int stage) { s_->gpu_jpeg = std::max(0, std::min(2, stage)); }

void DatasetReader::setPrefetch(int frames) {
  State& s = *s_;
  s.prefetch = std::max(0, std::min(frames, 64));
}

const unsigned char* DatasetReader::getImageRaw(int id, int* width, int* height) {
  State& s = *s_;
  s.err.clear();
  if (id < 0 || id >= (int)s.files.size()) {
    s.err = "frame index out of range";
    return 0;
  }
  Decode* d = s.fetch(id);
  if (width) *width = d->w;
  if (height) *height = d->h;
  if (!d->ok) {
    s.err = d->err;
    return 0;
  }
  return d->dst;
}

ExposureImage* DatasetReader::getImage(int id, bool rectify, bool removeGamma, bool removeVignette, bool nanOverexposed) {
  State& s = *s_;
  if (id >= 0 && id < (int)s.files.size() && s.gpu && s.lookahead > 0 && s.gpu_jpeg >= 2) {
    const unsigned flags = flag_word(rectify, removeGamma, removeVignette, nanOverexposed);
    if (!s.ahead.empty()) {  // results made ahead: hand this one out, or drop them when the caller went elsewhere
      const int k = id - s.ahead_first;
      if (flags == s.ahead_flags && k >= 0 && k < (int)s.ahead.size() && s.ahead[(size_t)k]) {
        ExposureImage* ret = s.ahead[(size_t)k];
        s.ahead[(size_t)k] = 0;
        if (k + 1 == (int)s.ahead.size()) {  // used up in order: the next batch is twice as long
          const int grown = std::min(2 * s.ahead_batch, (int)State::kRingFrames);
          s.drop_ahead();
          s.ahead_batch = grown;
        }
        s.seq_run = id == s.seq_last + 1 ? s.seq_run + 1 : 0;
        s.seq_last = id;
        return ret;
      }
      if (flags != s.ahead_flags || k < 0 || k >= (int)s.ahead.size()) s.drop_ahead();
    }
    s.seq_run = id == s.seq_last + 1 ? s.seq_run + 1 : 0;
    s.seq_last = id;
    if (s.ahead.empty() && s.seq_run >= 2 && s.is_jpeg_name((size_t)id)) {
      const int n = std::min(std::min(s.lookahead, s.ahead_batch), (int)s.files.size() - id);
      s.ahead.assign((size_t)n, (ExposureImage*)0);
      s.quiet_batch = true;
      getImages(id, n, rectify, removeGamma, removeVignette, nanOverexposed, s.ahead.data());
      s.quiet_batch = false;
      s.ahead_first = id;
      s.ahead_flags = flags;
      if (s.ahead[0]) {
        ExposureImage* ret = s.ahead[0];
        s.ahead[0] = 0;
        if (n == 1) s.drop_ahead();
        return ret;
      }
      // (this frame failed in the batch: the single-frame path below says why, as the reference would)
    }
  }
  int fw = 0, fh = 0;
  const unsigned char* raw = getImageRaw(id, &fw, &fh);
  if (id < 0 || id >= (int)s.files.size()) return 0;
  if (fh != s.H || fw != s.W) {  // also what an undecodable file leads to in the reference: an empty cv::Mat (:194-199)
    std::printf("ERROR: expected cv-mat to have dimensions %d x %d; found %d x %d (image %s)!\n", s.W, s.H, fw, fh,
                s.files[(size_t)id].c_str());
    if (!raw && !s.err.empty()) std::fprintf(stderr, "DatasetReader: %s\n", s.err.c_str());
    return 0;
  }
  if (!raw) {
    std::fprintf(stderr, "DatasetReader: %s\n", s.err.c_str());
    return 0;
  }
  if (!s.gpu) {
    s.err = "no GPU context: the per-frame pass has no CPU fallback";
    std::fprintf(stderr, "DatasetReader::getImage: %s\n", s.err.c_str());
    return 0;
  }
  ExposureImage* ret = rectify ? new ExposureImage(s.w, s.h, s.timestamps[(size_t)id], s.exposures[(size_t)id], id)
                               : new ExposureImage(s.W, s.H, s.timestamps[(size_t)id], s.exposures[(size_t)id], id);
  // the four switches are the library's flag word; every combination -- also "none" (plain cast, :234-240)
  // and "rectify only" (undistort<unsigned char>, :228-233) -- is one pass of the same kernel family
  if (mdc_process_host(s.gpu, raw, ret->image, flag_word(rectify, removeGamma, removeVignette, nanOverexposed)) != MDC_OK) {
    s.err = mdc_last_error(s.gpu);
    std::fprintf(stderr, "DatasetReader::getImage: %s\n", s.err.c_str());
    delete ret;
    return 0;
  }
  return ret;
}

// A lane's host thread issues its device's copies and launches: it belongs on the CPUs next to that GPU (on a two-socket 8-GPU node
// half of the devices hang off the other socket; a thread there pays the socket hop on every doorbell and staging copy).
// /sys/bus/pci/devices/<pci>/local_cpulist ("0-63,128-191") -> sched_setaffinity of the calling thread.  MDC_NUMA_PIN=0 turns it off;
// any failure (no sysfs, empty list, a cpuset that excludes those CPUs) leaves the thread where it is.
std::vector<int> parse_cpulist(const std::string& text) {
  std::vector<int> cpus;
  size_t i = 0;
  while (i < text.size()) {
    while (i < text.size() && !isdigit((unsigned char)text[i])) i++;
    if (i >= text.size()) break;
    int a = 0, b;
    while (i < text.size() && isdigit((unsigned char)text[i])) a = a * 10 + (text[i++] - '0');
    b = a;
    if (i < text.size() && text[i] == '-') {
      i++;
      b = 0;
      while (i < text.size() && isdigit((unsigned char)text[i])) b = b * 10 + (text[i++] - '0');
    }
    for (int c = a; c <= b && c < CPU_SETSIZE && cpus.size() < 4096; c++) cpus.push_back(c);
  }
  return cpus;
}
void pin_thread_near_device(mdc_ctx* gpu) {
  static const bool enabled = [] {
    const char* e = std::getenv("MDC_NUMA_PIN");
    return !e || std::atoi(e) != 0;
  }();
  char pci[32];
  if (!enabled || !gpu || mdc_device_pci_bus_id(gpu, pci, sizeof pci) != MDC_OK) return;
  std::ifstream f(std::string("/sys/bus/pci/devices/") + pci + "/local_cpulist");
  std::string line;
  if (!f || !std::getline(f, line)) return;
  const std::vector<int> cpus = parse_cpulist(line);
  if (cpus.empty()) return;
  cpu_set_t set;
  CPU_ZERO(&set);
  for (int c : cpus) CPU_SET(c, &set);
  (void)sched_setaffinity(0, sizeof set, &set);  // refused (cpuset): stay
}

// One device of a sharded getImages call: chunks k = lane, lane + L, lane + 2L, ... of the range, each on the lane's own
// context, decode ring and GPU calls (reference src/BenchmarkDatasetReader.h:188-243: a frame depends on nothing but itself
// and the immutable tables, so the chunks of a range are independent).  The decode pool is shared; results land in the
// caller's order because every chunk writes its own slice of `out`.
struct DatasetReader::State::LaneRun {
  State& s;
  Lane& lane;
  int first, count, C, RG, L, li;
  bool rectify;
  unsigned flags;
  ExposureImage** out;
  std::vector<Decode>& rec;
  int produced = 0;
  const mdc_device_outputs* dev = nullptr;  // getImagesDevice: results stay in the caller's device arrays (out == 0)
  unsigned char* valid = nullptr;           // ... position i holds a result
  int cur_i0 = 0, cur_i1 = 0;  // the chunk in flight: images allocated, pixels not (yet) written -- see drop_chunk_in_flight()
  double t_wait = 0, t_gpu = 0;
  LaneRun(State& s_, Lane& lane_, int first_, int count_, int C_, int RG_, int L_, int li_, bool rectify_, unsigned flags_, ExposureImage** out_,
          std::vector<Decode>& rec_)
      : s(s_), lane(lane_), first(first_), count(count_), C(C_), RG(RG_), L(L_), li(li_), rectify(rectify_), flags(flags_), out(out_), rec(rec_) {}

  static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  int nchunks() const { return (count + C - 1) / C; }
  // the lane's j-th chunk is chunk li + j * L of the range; its buffers are ring position j % RG
  void submit(int j) {
    const int k = li + j * L;
    std::lock_guard<std::mutex> lk(s.mu);
    for (int i = k * C; i < std::min(count, (k + 1) * C); i++) {
      Decode& d = rec[(size_t)i];
      d.id = first + i;
      d.dst = lane.ring_block.p + (size_t)((j % RG) * C + (i - k * C)) * lane.ring_stride;
      d.cap = lane.ring_bytes;
      d.want_record_pitch = (s.gpu_jpeg && lane.ring_bytes >= s.rec_bytes) ? s.rec_pitch : 0;
      d.want_stream = s.gpu_jpeg >= 2;
      s.submit(&d);
    }
    s.cv_job.notify_all();
  }
  void note_error(const std::string& e) {
    std::lock_guard<std::mutex> lk(s.err_mu);
    s.err = e;
  }
  void bad_frame(int id, int w, int h) {
    if (!s.quiet_batch)
      std::printf("ERROR: expected cv-mat to have dimensions %d x %d; found %d x %d (image %s)!\n", s.W, s.H, w, h, s.files[(size_t)id].c_str());
  }

  // run() threw (out of memory for an image or a list of pointers): the images it had made for the current chunk hold no
  // pixels yet and are not counted in `produced` -- a caller walking out[] for non-null entries must not meet them
  void drop_chunk_in_flight() {
    for (int i = cur_i0; i < cur_i1; i++) {
      if (out) {
        delete out[i];
        out[i] = 0;
      }
      if (valid) valid[i] = 0;
    }
    cur_i0 = cur_i1 = 0;
  }

  void run() {
    const int mine = (nchunks() - li + L - 1) / L;  // chunks of this lane
    for (int j = 0; j < std::min(mine, RG); j++) submit(j);
    std::vector<const uint8_t*> src;
    std::vector<float*> dst;
    std::vector<const void*> rsrc;  // frames of the chunk that arrived as JPEG coefficient records
    std::vector<float*> rdst;
    std::vector<const void*> ssrc;  // ... as JPEG streams (Huffman decoding on the device)
    std::vector<float*> sdst;
    std::vector<int64_t> ssize;
    std::vector<int> sstatus, sidx;
    std::vector<int64_t> pidx, ridx, spos;  // getImagesDevice: positions (in the caller's device arrays) of the chunk's plain / record / stream frames
    for (int j = 0; j < mine; j++) {
      const int k = li + j * L, i0 = k * C, i1 = std::min(count, (k + 1) * C);
      const double tw = now();
      {
        std::unique_lock<std::mutex> lk(s.mu);
        s.cv_done.wait(lk, [&] {
          for (int i = i0; i < i1; i++)
            if (!rec[(size_t)i].done) return false;
          return true;
        });
      }
      t_wait += now() - tw;
      cur_i0 = i0;
      cur_i1 = i1;
      src.clear();
      dst.clear();
      rsrc.clear();
      rdst.clear();
      ssrc.clear();
      sdst.clear();
      ssize.clear();
      sidx.clear();
      pidx.clear();
      ridx.clear();
      spos.clear();
      {
        // a chunk's images are made in one go: the pool hands out consecutive blocks of a slab (lowest free address first),
        // and a chunk whose results lie back to back leaves the device with one copy -- another lane allocating in between
        // would interleave the two chunks' images
        std::lock_guard<std::mutex> alk(s.image_mu);
        for (int i = i0; i < i1; i++) {
          const Decode& d = rec[(size_t)i];
          const int id = first + i;
          if (!d.ok || d.w != s.W || d.h != s.H) {
            bad_frame(id, d.w, d.h);
            if (!d.ok) note_error(d.err);
            continue;
          }
          if (!dev)
            out[i] = rectify ? new ExposureImage(s.w, s.h, s.timestamps[(size_t)id], s.exposures[(size_t)id], id)
                             : new ExposureImage(s.W, s.H, s.timestamps[(size_t)id], s.exposures[(size_t)id], id);
          if (valid) valid[i] = 1;
          if (d.is_stream) {
            ssrc.push_back(d.dst);
            if (!dev) sdst.push_back(out[i]->image);
            spos.push_back(i);
            ssize.push_back((int64_t)d.stream_bytes);
            sidx.push_back(i);
          } else if (d.is_record) {
            if (d.rec_rows > s.rec_rows) {  // cannot happen while the decoder checks the sink's capacity: never hand a record on as pixels
              if (!dev) {
                delete out[i];
                out[i] = 0;
              }
              if (valid) valid[i] = 0;
              note_error(s.files[(size_t)id] + ": coefficient record larger than the frame's geometry");
              continue;
            }
            rsrc.push_back(d.dst);
            if (!dev) rdst.push_back(out[i]->image);
            ridx.push_back(i);
          } else {
            src.push_back(d.dst);
            if (!dev) dst.push_back(out[i]->image);
            pidx.push_back(i);
          }
        }
      }
      // chunk k on the GPU (uploads, kernels and downloads pipelined inside the call) while the pool decodes the next chunks
      const double tg = now();
      int refused = 0;  // streams neither the device nor the host decoder could read
      mdc_ctx* gpu = lane.gpu;
      int grc = MDC_OK;
      if (!src.empty())
        grc = dev ? mdc_process_frames_host_to_device(gpu, src.data(), (int64_t)src.size(), flags, dev, pidx.data())
                  : mdc_process_frames_host(gpu, src.data(), dst.data(), (int64_t)src.size(), flags);
      if (grc == MDC_OK && !rsrc.empty())  // records: Huffman-decoded on the host, inverse DCT on the device
        grc = dev ? mdc_process_jpeg_frames_host_to_device(gpu, rsrc.data(), (int64_t)s.rec_bytes, s.rec_pitch, s.rec_rows, (int64_t)rsrc.size(), flags, dev, ridx.data())
                  : mdc_process_jpeg_frames_host(gpu, rsrc.data(), (int64_t)s.rec_bytes, s.rec_pitch, s.rec_rows, rdst.data(), (int64_t)rsrc.size(), flags);
      if (grc == MDC_OK && !ssrc.empty()) {  // streams: Huffman decoding, inverse DCT and the fused pass on the device
        sstatus.assign(ssrc.size(), 0);
        grc = dev ? mdc_process_jpeg_streams_host_to_device(gpu, ssrc.data(), ssize.data(), (int64_t)ssrc.size(), flags, dev, spos.data(), sstatus.data())
                  : mdc_process_jpeg_streams_host(gpu, ssrc.data(), ssize.data(), sdst.data(), (int64_t)ssrc.size(), flags, sstatus.data());
        for (size_t q = 0; q < ssrc.size() && grc == MDC_OK; q++)
          if (sstatus[q] != 0) {  // a stream the device could not decode (damaged file): the host decoder has the last word
            const int i = sidx[q];
            Decode one;
            one.id = first + i;
            one.dst = const_cast<unsigned char*>(static_cast<const unsigned char*>(ssrc[q]));  // the ring buffer of this frame
            one.cap = lane.ring_bytes;
            s.decode_now(one);
            if (one.ok && one.w == s.W && one.h == s.H) {
              const uint8_t* one_src = one.dst;
              const int64_t one_pos = i;
              grc = dev ? mdc_process_frames_host_to_device(gpu, &one_src, 1, flags, dev, &one_pos) : mdc_process_host(gpu, one.dst, out[i]->image, flags);
            } else {
              bad_frame(first + i, one.w, one.h);
              if (!one.ok) note_error(one.err);
              if (!dev) {
                delete out[i];
                out[i] = 0;
              }
              if (valid) valid[i] = 0;
              refused++;
            }
          }
      }
      t_gpu += now() - tg;
      if (grc != MDC_OK) {
        note_error(mdc_last_error(gpu));
        std::fprintf(stderr, "DatasetReader::getImages: %s\n", mdc_last_error(gpu));
        for (int i = i0; i < i1; i++) {
          if (!dev) {
            delete out[i];
            out[i] = 0;
          }
          if (valid) valid[i] = 0;
        }
      } else {
        produced += (int)src.size() + (int)rsrc.size() + (int)ssrc.size() - refused;
      }
      cur_i0 = cur_i1 = 0;  // the chunk is settled: its images are results (or gone)
      if (j + RG < mine) submit(j + RG);  // the buffers of the lane's chunk j are free again
    }
    lane.frames += produced;
    lane.t_wait += t_wait;
    lane.t_gpu += t_gpu;
  }
};

int DatasetReader::getImages(int first, int count, bool rectify, bool removeGamma, bool removeVignette, bool nanOverexposed,
                             ExposureImage** out) {
  if (!out || count <= 0) return 0;
  return run_batch(first, count, rectify, removeGamma, removeVignette, nanOverexposed, out, 0, 0);
}

int DatasetReader::getImagesDevice(int first, int count, bool rectify, bool removeGamma, bool removeVignette, bool nanOverexposed,
                                   const mdc_device_outputs* out, unsigned char* valid) {
  if (!out || !out->base || count <= 0) {
    s_->err = "getImagesDevice: no device outputs";
    return 0;
  }
  return run_batch(first, count, rectify, removeGamma, removeVignette, nanOverexposed, 0, out, valid);
}

mdc_ctx* DatasetReader::getContext() { return s_->gpu; }
int DatasetReader::getDevice() const { return s_->lanes.empty() || !s_->gpu ? -1 : s_->lanes[0].device; }

// getImages (out) / getImagesDevice (dev, valid): the decode pool -> per-device lanes -> pipelined GPU calls
int DatasetReader::run_batch(int first, int count, bool rectify, bool removeGamma, bool removeVignette, bool nanOverexposed, ExposureImage** out,
                             const mdc_device_outputs* dev, unsigned char* valid) {
  State& s = *s_;
  s.err.clear();
  for (int i = 0; i < count && out; i++) out[i] = 0;
  for (int i = 0; i < count && valid; i++) valid[i] = 0;
  if (first < 0 || first + count > (int)s.files.size()) {
    s.err = "frame range outside the sequence";
    return 0;
  }
  if (!s.gpu) {
    s.err = "no GPU context: the per-frame pass has no CPU fallback";
    std::fprintf(stderr, "DatasetReader::getImages: %s\n", s.err.c_str());
    return 0;
  }
  // Frames per GPU call / calls in a lane's ring.  Stage 2 hands over up to a whole ring at a time -- its host work is ~0.1 ms
  // per frame and thread, and inside the GPU call a 64-frame chunk decodes while the one before it goes out: the longer the
  // call, the less its first decode and last output weigh (128 per call: 16.5 k frames/s, 256: 20+ k) -- and a lane with
  // more than one call gets a second ring's worth of buffers, so that the pool parses the next files while the GPU call of
  // the current ones runs (one ring: parse and GPU call take turns, 22 k frames/s).  With several devices (MDC_DEVICES) the
  // range is dealt to them in chunks of at least 64 frames, round-robin.
  // device outputs live on ONE device, the first lane's: the lanes on that device take part (MDC_DEVICES=0,0: two lanes on one GPU --
  // two host threads whose pipelined calls overlap, one lane's fill and drain under the other's decode)
  if (!s.host_lanes) s.host_lanes = (int)s.lanes.size();
  int L = s.host_lanes;  // (twin lanes made for getImagesDevice take no part in getImages)
  std::vector<State::Lane*> use;
  if (dev) {
    s.ensure_device_lanes();
    for (State::Lane& ln : s.lanes)
      if (ln.device == s.lanes[0].device && ln.gpu) use.push_back(&ln);
    L = (int)use.size();
  } else {
    for (int l = 0; l < L; l++) use.push_back(&s.lanes[(size_t)l]);
  }
  int C = s.gpu_jpeg >= 2 ? State::kRingFrames : 32;
  if (L > 1 && s.gpu_jpeg >= 2) C = std::min<int>(State::kRingFrames, std::max(64, ((count + L - 1) / L + 63) / 64 * 64));
  const int nchunks = (count + C - 1) / C;
  const int per_lane = (nchunks + L - 1) / L;
  const int slots = (s.gpu_jpeg >= 2 && per_lane > 1) ? 2 * C : (s.gpu_jpeg >= 2 ? C : State::kRingFrames), RG = slots / C;
  // coefficient records (include/mdc_hip.h): MCUs are 1..4 x 1..4 blocks, so a grid rounded up to multiples of 12 blocks
  // holds every sampling layout of a W x H file (the same rule as mdch_jpeg_record_bytes)
  s.rec_pitch = ((s.W + 7) / 8 + 11) / 12 * 12;
  s.rec_rows = ((s.H + 7) / 8 + 11) / 12 * 12;
  s.rec_bytes = 128 + (size_t)s.rec_pitch * s.rec_rows * 128;
  // a ring buffer holds a decoded frame, or (stage 1) a coefficient record -- 2 bytes per pixel --, or (stage 2) a stream: the
  // compressed bytes + 5 KB; a file stage 2 does not take, or whose stream does not fit, is decoded to pixels on the host
  const size_t want_bytes = s.gpu_jpeg == 1 ? std::max(s.frame_bytes(), s.rec_bytes) : s.frame_bytes();
  const int active = std::min(L, nchunks);
  for (int l = 0; l < active; l++) {
    State::Lane& ln = *use[(size_t)l];
    if (!ln.ring_block.p || ln.ring_bytes < want_bytes || ln.ring_slots < slots) {
      ln.ring_block.release();
      ln.ring_stride = (want_bytes + 4095) & ~(size_t)4095;
      ln.ring_slots = slots;
      ln.ring_block.alloc(ln.ring_stride * (size_t)ln.ring_slots);
      ln.ring_bytes = want_bytes;
    }
  }
  s.start_pool();
  std::vector<Decode> rec((size_t)count);
  const unsigned flags = flag_word(rectify, removeGamma, removeVignette, nanOverexposed);
  const bool trace = std::getenv("MDC_READER_TRACE") != 0;  // where a getImages call spends its time (stderr)
  std::vector<State::LaneRun> runs;
  runs.reserve((size_t)active);
  for (int l = 0; l < active; l++) {
    runs.push_back(State::LaneRun(s, *use[(size_t)l], first, count, C, RG, active, l, rectify, flags, out, rec));
    runs.back().dev = dev;
    runs.back().valid = valid;
  }
  // every lane runs to its end whatever happens in another one (an exception -- out of memory for a list of pointers -- ends
  // that lane's chunks with an error, not the process: a std::thread must not be left joinable, a lane's images must not leak)
  auto run_lane = [&runs, &s](int l, bool helper_thread) {
    try {
      // only a helper thread of this call is pinned near its GPU: the caller's own thread -- lane 0, and any lane that runs here because no
      // thread could be made -- keeps the affinity the application gave it
      if (helper_thread) pin_thread_near_device(runs[(size_t)l].lane.gpu);
      runs[(size_t)l].run();
    } catch (const std::exception& e) {
      runs[(size_t)l].drop_chunk_in_flight();
      std::lock_guard<std::mutex> lk(s.err_mu);
      s.err = std::string("getImages: lane failed: ") + e.what();
    } catch (...) {
      runs[(size_t)l].drop_chunk_in_flight();
      std::lock_guard<std::mutex> lk(s.err_mu);
      s.err = "getImages: lane failed";
    }
  };
  std::vector<std::thread> helpers;
  for (int l = 1; l < active; l++) {
    try {
      helpers.emplace_back(run_lane, l, true);
    } catch (...) {  // no thread to be had: the lane's chunks run here, after lane 0's
      helpers.emplace_back();
    }
  }
  run_lane(0, false);
  for (int l = 1; l < active; l++) {
    if (helpers[(size_t)(l - 1)].joinable()) helpers[(size_t)(l - 1)].join();
    else run_lane(l, false);
  }
  {  // `rec` dies with this call: no decode job may still point into it (a lane that ended early leaves some queued)
    std::unique_lock<std::mutex> lk(s.mu);
    s.cv_done.wait(lk, [&] {
      for (const Decode& d : rec)
        if (!d.done) return false;
      return true;
    });
  }
  int produced = 0;
  for (const State::LaneRun& r : runs) produced += r.produced;
  if (trace)
    for (const State::LaneRun& r : runs)
      std::fprintf(stderr, "DatasetReader::getImages: device %d (lane %d of %d): %d of %d frames, %d decode threads: waited %.1f ms for the decoders, %.1f ms in the GPU calls\n",
                   r.lane.device, r.li, active, r.produced, count, (int)s.workers.size(), r.t_wait * 1e3, r.t_gpu * 1e3);
  return produced;
}
