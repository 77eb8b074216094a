// libmdc_multi.so -- implementation of include/mdc_multi.h: one process, N devices, one mdc_ctx and one
// RCCL communicator per device.  The only collective of the whole path is the table broadcast; the
// data path is N independent streams (frame f -> device f % N).
#include "../../include/mdc_multi.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

// One PERSISTENT host thread per device (created with the object, parked on a condition variable between calls): it has
// the device current once and for all and launches that device's work, so a call costs a wake-up, not a thread spawn.
struct Worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<int()> job;  // set = work pending
  bool has_job = false, done = false, stop = false;
  int rc = 0;
};

struct mdc_multi {
  std::vector<int> dev;
  std::vector<mdc_ctx*> ctx;
  std::vector<ncclComm_t> comm;
  std::vector<hipStream_t> stream;
  std::vector<Worker*> worker;
  std::mutex call_mu;  // one multi-device call at a time
  std::string err;
};

namespace {

thread_local std::string g_err;

int fail(mdc_multi* m, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (m) m->err = buf;
  else g_err = buf;
  return code;
}

void worker_main(Worker* w, int device) {
  (void)hipSetDevice(device);
  std::unique_lock<std::mutex> lk(w->mu);
  for (;;) {
    w->cv.wait(lk, [&] { return w->stop || w->has_job; });
    if (w->stop) return;
    std::function<int()> job = std::move(w->job);
    lk.unlock();
    const int rc = job();
    lk.lock();
    w->rc = rc;
    w->has_job = false;
    w->done = true;
    w->cv.notify_all();
  }
}

// runs fn(rank) on every device's worker thread, concurrently; the first non-zero status wins
template <typename F>
int per_device(mdc_multi* m, F fn) {
  const int n = (int)m->dev.size();
  for (int r = 0; r < n; r++) {
    Worker* w = m->worker[(size_t)r];
    std::lock_guard<std::mutex> lk(w->mu);
    w->job = [fn, r]() { return fn(r); };
    w->has_job = true;
    w->done = false;
    w->cv.notify_all();
  }
  int rc = MDC_OK;
  for (int r = 0; r < n; r++) {
    Worker* w = m->worker[(size_t)r];
    std::unique_lock<std::mutex> lk(w->mu);
    w->cv.wait(lk, [&] { return w->done; });
    if (rc == MDC_OK && w->rc != MDC_OK) rc = w->rc;
  }
  return rc;
}

}  // namespace

extern "C" {

int mdc_multi_create(const int* devices, int ndev, mdc_multi** out) try {
  if (!out) return fail(nullptr, MDC_ERR_ARG, "mdc_multi_create: out is NULL");
  *out = nullptr;
  int visible = 0;
  if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0)
    return fail(nullptr, MDC_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU fallback");
  if (ndev <= 0) {
    ndev = visible;
    devices = nullptr;
  }
  mdc_multi* m = new mdc_multi();
  for (int r = 0; r < ndev; r++) m->dev.push_back(devices ? devices[r] : r);
  for (int r = 0; r < ndev; r++) {
    if (m->dev[(size_t)r] < 0 || m->dev[(size_t)r] >= visible) {
      fail(nullptr, MDC_ERR_ARG, "device %d out of range [0,%d)", m->dev[(size_t)r], visible);
      mdc_multi_destroy(m);
      return MDC_ERR_ARG;
    }
    mdc_ctx* c = nullptr;
    const int rc = mdc_create(m->dev[(size_t)r], &c);
    if (rc != MDC_OK) {
      fail(nullptr, rc, "mdc_create(device %d): %s", m->dev[(size_t)r], mdc_last_error(nullptr));
      mdc_multi_destroy(m);
      return rc;
    }
    m->ctx.push_back(c);
    hipStream_t s = nullptr;
    (void)hipSetDevice(m->dev[(size_t)r]);
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
      fail(nullptr, MDC_ERR_HIP, "hipStreamCreate on device %d failed", m->dev[(size_t)r]);
      mdc_multi_destroy(m);
      return MDC_ERR_HIP;
    }
    m->stream.push_back(s);
    Worker* w = new Worker();
    w->th = std::thread(worker_main, w, m->dev[(size_t)r]);
    m->worker.push_back(w);
  }
  // one communicator per device, all in this process (xGMI between the devices of a node)
  m->comm.assign((size_t)ndev, nullptr);
  const ncclResult_t nr = ncclCommInitAll(m->comm.data(), ndev, m->dev.data());
  if (nr != ncclSuccess) {
    m->comm.clear();
    fail(nullptr, MDC_ERR_HIP, "ncclCommInitAll(%d devices): %s", ndev, ncclGetErrorString(nr));
    mdc_multi_destroy(m);
    return MDC_ERR_HIP;
  }
  *out = m;
  return MDC_OK;
} catch (const std::exception& e_) {
  return fail(nullptr, MDC_ERR_NOMEM, "mdc_multi_create: %s", e_.what());
} catch (...) {
  return fail(nullptr, MDC_ERR_HIP, "mdc_multi_create: unexpected exception");
}

void mdc_multi_destroy(mdc_multi* m) {
  if (!m) return;
  for (Worker* w : m->worker) {
    {
      std::lock_guard<std::mutex> lk(w->mu);
      w->stop = true;
      w->cv.notify_all();
    }
    w->th.join();
    delete w;
  }
  for (size_t r = 0; r < m->comm.size(); r++)
    if (m->comm[r]) (void)ncclCommDestroy(m->comm[r]);
  for (size_t r = 0; r < m->stream.size(); r++) {
    (void)hipSetDevice(m->dev[r]);
    (void)hipStreamSynchronize(m->stream[r]);
    (void)hipStreamDestroy(m->stream[r]);
  }
  for (mdc_ctx* c : m->ctx) mdc_destroy(c);
  delete m;
}

int mdc_multi_size(const mdc_multi* m) { return m ? (int)m->dev.size() : 0; }
int mdc_multi_comm_count(const mdc_multi* m, int rank) try {
  if (!m || rank < 0 || rank >= (int)m->comm.size()) return -1;
  int n = -1;
  return ncclCommCount(m->comm[(size_t)rank], &n) == ncclSuccess ? n : -1;
} catch (const std::exception& e_) {
  return fail(nullptr, MDC_ERR_NOMEM, "mdc_multi_comm_count: %s", e_.what());
} catch (...) {
  return fail(nullptr, MDC_ERR_HIP, "mdc_multi_comm_count: unexpected exception");
}
void* mdc_multi_stream(mdc_multi* m, int rank) { return (m && rank >= 0 && rank < (int)m->stream.size()) ? (void*)m->stream[(size_t)rank] : nullptr; }
mdc_ctx* mdc_multi_ctx(mdc_multi* m, int rank) { return (m && rank >= 0 && rank < (int)m->ctx.size()) ? m->ctx[(size_t)rank] : nullptr; }
int mdc_multi_device(const mdc_multi* m, int rank) { return (m && rank >= 0 && rank < (int)m->dev.size()) ? m->dev[(size_t)rank] : -1; }
const char* mdc_multi_last_error(const mdc_multi* m) { return m ? m->err.c_str() : g_err.c_str(); }

int mdc_multi_bcast_tables(mdc_multi* m, int root) try {
  if (!m) return MDC_ERR_ARG;
  std::lock_guard<std::mutex> call(m->call_mu);
  const int n = (int)m->dev.size();
  if (root < 0 || root >= n) return fail(m, MDC_ERR_ARG, "root %d out of range [0,%d)", root, n);
  size_t bytes = 0;
  int rc = mdc_export_tables(m->ctx[(size_t)root], nullptr, 0, &bytes);
  if (rc != MDC_OK) return fail(m, rc, "export on root: %s", mdc_last_error(m->ctx[(size_t)root]));
  std::vector<unsigned char> blob(bytes);
  rc = mdc_export_tables(m->ctx[(size_t)root], blob.data(), blob.size(), &bytes);
  if (rc != MDC_OK) return fail(m, rc, "export on root: %s", mdc_last_error(m->ctx[(size_t)root]));

  // device-side staging: the broadcast moves the blob GPU -> GPU over xGMI
  std::vector<void*> d_blob((size_t)n, nullptr);
  auto cleanup = [&] {
    for (int r = 0; r < n; r++)
      if (d_blob[(size_t)r]) {
        (void)hipSetDevice(m->dev[(size_t)r]);
        (void)hipFree(d_blob[(size_t)r]);
      }
  };
  for (int r = 0; r < n; r++) {
    (void)hipSetDevice(m->dev[(size_t)r]);
    if (hipMalloc(&d_blob[(size_t)r], bytes) != hipSuccess) {
      cleanup();
      return fail(m, MDC_ERR_HIP, "hipMalloc(%zu) on device %d failed", bytes, m->dev[(size_t)r]);
    }
  }
  (void)hipSetDevice(m->dev[(size_t)root]);
  if (hipMemcpyAsync(d_blob[(size_t)root], blob.data(), bytes, hipMemcpyHostToDevice, m->stream[(size_t)root]) != hipSuccess) {
    cleanup();
    return fail(m, MDC_ERR_HIP, "upload of the table blob to the root failed");
  }
  ncclResult_t nr = ncclGroupStart();
  for (int r = 0; r < n && nr == ncclSuccess; r++)
    nr = ncclBroadcast(d_blob[(size_t)root], d_blob[(size_t)r], bytes, ncclUint8, root, m->comm[(size_t)r], m->stream[(size_t)r]);
  const ncclResult_t ne = ncclGroupEnd();
  if (nr == ncclSuccess) nr = ne;
  if (nr != ncclSuccess) {
    cleanup();
    return fail(m, MDC_ERR_HIP, "ncclBroadcast of the tables: %s", ncclGetErrorString(nr));
  }
  // every rank (the root too) imports what it holds after the collective
  rc = per_device(m, [&](int r) {
    (void)hipSetDevice(m->dev[(size_t)r]);
    std::vector<unsigned char> mine(bytes);
    if (hipMemcpyAsync(mine.data(), d_blob[(size_t)r], bytes, hipMemcpyDeviceToHost, m->stream[(size_t)r]) != hipSuccess ||
        hipStreamSynchronize(m->stream[(size_t)r]) != hipSuccess)
      return (int)MDC_ERR_HIP;
    return mdc_import_tables(m->ctx[(size_t)r], mine.data(), mine.size());
  });
  cleanup();
  if (rc != MDC_OK) return fail(m, rc, "import of the broadcast tables failed on a rank");
  return MDC_OK;
} catch (const std::exception& e_) {
  return fail(m, MDC_ERR_NOMEM, "mdc_multi_bcast_tables: %s", e_.what());
} catch (...) {
  return fail(m, MDC_ERR_HIP, "mdc_multi_bcast_tables: unexpected exception");
}

int64_t mdc_multi_frames_of_rank(const mdc_multi* m, int64_t total, int rank) {
  const int64_t n = m ? (int64_t)m->dev.size() : 0;
  if (n <= 0 || rank < 0 || rank >= n || total <= rank) return 0;
  return (total - rank + n - 1) / n;
}

int mdc_multi_process_sequence_device(mdc_multi* m, const uint8_t* const* d_in, float* const* d_out, int64_t total,
                                      unsigned flags) try {
  if (!m || !d_in || !d_out || total < 0) return fail(m, MDC_ERR_ARG, "mdc_multi_process_sequence_device: bad argument");
  std::lock_guard<std::mutex> call(m->call_mu);
  const int rc = per_device(m, [&](int r) {
    const int64_t mine = mdc_multi_frames_of_rank(m, total, r);
    if (mine == 0) return (int)MDC_OK;
    if (!d_in[r] || !d_out[r]) return (int)MDC_ERR_ARG;
    (void)hipSetDevice(m->dev[(size_t)r]);
    return mdc_process_batch_device(m->ctx[(size_t)r], d_in[r], d_out[r], mine, flags, m->stream[(size_t)r]);
  });
  if (rc != MDC_OK) return fail(m, rc, "a rank failed to launch its shard");
  return MDC_OK;
} catch (const std::exception& e_) {
  return fail(m, MDC_ERR_NOMEM, "mdc_multi_process_sequence_device: %s", e_.what());
} catch (...) {
  return fail(m, MDC_ERR_HIP, "mdc_multi_process_sequence_device: unexpected exception");
}

int mdc_multi_synchronize(mdc_multi* m) try {
  if (!m) return MDC_ERR_ARG;
  for (size_t r = 0; r < m->dev.size(); r++) {
    (void)hipSetDevice(m->dev[r]);
    if (hipStreamSynchronize(m->stream[r]) != hipSuccess) return fail(m, MDC_ERR_HIP, "stream of rank %zu failed", r);
  }
  return MDC_OK;
} catch (const std::exception& e_) {
  return fail(m, MDC_ERR_NOMEM, "mdc_multi_synchronize: %s", e_.what());
} catch (...) {
  return fail(m, MDC_ERR_HIP, "mdc_multi_synchronize: unexpected exception");
}

}  // extern "C"
