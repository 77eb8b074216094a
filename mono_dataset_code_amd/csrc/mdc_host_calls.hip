// libmdc_hip.so: the synchronous host-pointer entry points -- the semantics of the reference methods they stand in for
// (PhotometricUndistorter::unMapImage, UndistorterFOV::undistort<T>, distortCoordinates, the per-frame body of
// DatasetReader::getImage): host pointers, blocking, the caller's buffers borrowed for the duration of the call only.
// Every call leases a slot (its own stream + staging buffers) so that calls from several host threads overlap; buffers in
// page-locked memory mapped into the device are read / written by the kernels in place (zero copy).
#include "mdc_ctx.h"

using namespace mdc;

namespace mdc {

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
const void* device_view_raw(const mdc_ctx* c, const void* p, size_t bytes) {
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
  return a.devicePointer;
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

SlotLease::SlotLease(mdc_ctx* ctx, bool wait) : c(ctx) {
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

SlotLease::~SlotLease() {
  if (!s) return;
  if (in_flight) (void)hipStreamSynchronize(s->stream);
  {
    std::lock_guard<std::mutex> lk(c->slot_mu);
    s->busy = false;
  }
  c->slot_cv.notify_one();
}

}  // namespace mdc

extern "C" {

int mdc_distort_points_host(mdc_ctx* c, const mdc_fov_model* model, float* x, float* y, int64_t n) try {
  if (!c) return MDC_ERR_ARG;
  if (!model || n < 0 || (n > 0 && (!x || !y))) return fail(c, MDC_ERR_ARG, "mdc_distort_points_host: bad argument");
  if (n == 0) return MDC_OK;
  ReadLock lk(c->mu);
  DeviceGuard dg(c->device);
  const size_t bytes = (size_t)n * sizeof(float);
  // Results reach the caller's arrays only after everything succeeded: a failure half way (x copied back, y not) would leave a
  // caller that falls back to its own loop (UndistorterFOV::distortCoordinates) distorting x a second time.  The landing buffers
  // are declared BEFORE the slot lease: on an early return the lease's destructor drains the stream first (a copy into them may
  // still be in flight), only then do they go away.
  std::vector<float> hx((size_t)n), hy((size_t)n);
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
  MDC_HIP(c, hipMemcpyAsync(hx.data(), dx, bytes, hipMemcpyDeviceToHost, st));
  MDC_HIP(c, hipMemcpyAsync(hy.data(), dy, bytes, hipMemcpyDeviceToHost, st));
  MDC_HIP(c, hipStreamSynchronize(st));
  slot.drained();
  memcpy(x, hx.data(), bytes);
  memcpy(y, hy.data(), bytes);
  return MDC_OK;
} MDC_CATCH(c)

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

void* mdc_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}

void mdc_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

}  // extern "C"
