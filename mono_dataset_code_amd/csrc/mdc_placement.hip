// Buffer placement: mdc_alloc_placed_device / mdc_free_placed_device (include/mdc_hip.h).
//
// Why this exists.  On MI355X the time of one and the same launch depends on WHERE its frame and result
// buffers lie: device memory comes in classes (runs of whole GiB of the driver's physical blocks); a read stream and a write stream
// that lie in the same class run 6-9 % slower than a pair from different classes (the headline launch: 1.48 against 1.61 ms), and
// neighbouring allocations are mostly of one class -- so the first two hipMalloc's of a process are often a slow pair.  User space
// cannot ask for a class; it can only measure.  Two ways of getting a fast pair, both by measurement:
//
//   MDC_PLACE_MALLOC  K hipMalloc'ed candidates for the frames and K for the results, spread over the device's memory by spacer
//                     allocations (given back before anything is timed); the pass itself is timed on every pair, the fastest pair is
//                     handed out, the others are freed.  Needs room for K pairs.
//   MDC_PLACE_VMM     the buffers are ASSEMBLED: physical pieces (hipMemCreate, 1 GiB by default) are mapped once into a probe range
//                     and sorted into classes by timing a linear read / write stream between a reference piece and every other one
//                     (same class = slow); the frames are then mapped from pieces of one class and the results from pieces of
//                     another.  Every virtual address is mapped exactly once and stays mapped until mdc_free_placed_device
//                     (round 5's map / probe / unmap churn ended in GPU memory faults: profiles/r05_experiments/07_*).  Works for
//                     buffers that leave no room for candidates (the 50,000-frame sequence: 65 + 61 GB).
//
// The reference call site this serves: the frame / result buffers a reader keeps for a sequence, src/BenchmarkDatasetReader.h:218-224
// (internalTempBuffer and the ExposureImage blocks), here device-resident.
#include "mdc_ctx.h"

namespace mdc {
namespace {

constexpr int MDC_PLACE_DEFAULT = MDC_PLACE_MALLOC;  // what MDC_PLACE_AUTO means (DESIGN.md section 6.1 has the measurements behind it)

__global__ __launch_bounds__(256) void placement_fill_kernel(uint32_t* __restrict__ p, size_t nwords, uint32_t seed) {
  // byte noise (a multiplicative hash per word): candidates are timed on frames that look like frames, not on zero pages
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (size_t)gridDim.x * 256) {
    uint32_t h = (uint32_t)i * 0x9E3779B1u + seed;
    h ^= h >> 15;
    h *= 0x85EBCA77u;
    h ^= h >> 13;
    p[i] = h;
  }
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// linear 16-byte reads of `rd16` chunks beside wave-contiguous nontemporal dword writes of `wr_words` words: the traffic mix of the
// path (1 byte read : 2 bytes written) without its arithmetic -- what two pieces of memory are worth TOGETHER
__global__ __launch_bounds__(256) void placement_stream_kernel(const u32x4* __restrict__ rd, size_t rd16, float* __restrict__ wr,
                                                               size_t wr_words) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
  uint32_t acc = 0;
  size_t i = t, j = t;
  while (i < rd16 || j < wr_words) {
    if (i < rd16) {
      const u32x4 v = __builtin_nontemporal_load(rd + i);
      acc ^= v.x ^ v.y ^ v.z ^ v.w;
      i += step;
    }
#pragma unroll
    for (int k = 0; k < 8; k++)
      if (j < wr_words) {
        __builtin_nontemporal_store(1.0f, wr + j);
        j += step;
      }
  }
  if (acc == 0x12345679u && wr_words) wr[0] = 2.0f;  // keeps the reads alive
}

struct Arena {
  int device = 0;
  int strategy = MDC_PLACE_FIRST;
  // hipMalloc strategies
  void* m_in = nullptr;
  void* m_out = nullptr;
  // assembled ranges
  std::vector<hipMemGenericAllocationHandle_t> handles;
  size_t piece = 0;
  void* probe_va = nullptr;  // handles.size() pieces, creation order
  size_t probe_mapped = 0;   // pieces mapped there
  void* in_va = nullptr;
  size_t in_pieces = 0, in_mapped = 0;
  void* out_va = nullptr;
  size_t out_pieces = 0, out_mapped = 0;
};

void release_arena(Arena* a) {
  if (!a) return;
  DeviceGuard dg(a->device);
  (void)hipDeviceSynchronize();  // nothing may still touch a range that is about to lose its pages
  if (a->m_in) (void)hipFree(a->m_in);
  if (a->m_out) (void)hipFree(a->m_out);
  auto unmap = [&](void* va, size_t mapped, size_t reserved) {
    if (!va) return;
    for (size_t k = 0; k < mapped; k++) (void)hipMemUnmap(static_cast<char*>(va) + k * a->piece, a->piece);
    (void)hipMemAddressFree(va, reserved * a->piece);
  };
  unmap(a->in_va, a->in_mapped, a->in_pieces);
  unmap(a->out_va, a->out_mapped, a->out_pieces);
  unmap(a->probe_va, a->probe_mapped, a->handles.size());
  for (hipMemGenericAllocationHandle_t h : a->handles) (void)hipMemRelease(h);  // after the last mapping is gone
  (void)hipDeviceSynchronize();
  delete a;
}

bool fill_noise(void* p, size_t bytes, uint32_t seed, hipStream_t s) {
  hipLaunchKernelGGL(placement_fill_kernel, dim3(4096), dim3(256), 0, s, static_cast<uint32_t*>(p), bytes / 4, seed);
  return hipGetLastError() == hipSuccess;
}

// median of 5 timed passes after 2 untimed ones
int time_pass(mdc_ctx* c, const uint8_t* d_in, float* d_out, int64_t nframes, unsigned flags, hipStream_t s, hipEvent_t e0, hipEvent_t e1,
              float* ms_out) {
  float ms[5] = {0, 0, 0, 0, 0};
  for (int k = 0; k < 7; k++) {
    if (k >= 2 && hipEventRecord(e0, s) != hipSuccess) return fail(c, MDC_ERR_HIP, "mdc_alloc_placed_device: hipEventRecord failed");
    const int rc = enqueue_process(c, d_in, d_out, nframes, flags, s);
    if (rc != MDC_OK) return rc;
    if (k >= 2 && (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
                   hipEventElapsedTime(&ms[k - 2], e0, e1) != hipSuccess))
      return fail(c, MDC_ERR_HIP, "mdc_alloc_placed_device: timing a pass failed");
  }
  std::sort(ms, ms + 5);
  *ms_out = ms[2];
  return MDC_OK;
}

// median of 3 timed linear streams (after one untimed) reading `rd` and writing `wr`
bool time_stream(const void* rd, size_t rd_bytes, void* wr, size_t wr_bytes, hipStream_t s, hipEvent_t e0, hipEvent_t e1, float* ms_out) {
  float ms[3] = {0, 0, 0};
  for (int k = 0; k < 4; k++) {
    if (k >= 1 && hipEventRecord(e0, s) != hipSuccess) return false;
    hipLaunchKernelGGL(placement_stream_kernel, dim3(8192), dim3(256), 0, s, static_cast<const u32x4*>(rd), rd_bytes / 16,
                       static_cast<float*>(wr), wr_bytes / 4);
    if (hipGetLastError() != hipSuccess) return false;
    if (k >= 1 && (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
                   hipEventElapsedTime(&ms[k - 1], e0, e1) != hipSuccess))
      return false;
  }
  std::sort(ms, ms + 3);
  *ms_out = ms[1];
  return true;
}

int env_int(const char* v, int dflt) {  // (called as env_int(getenv("MDC_..."), default): tests/test_abi.py finds every variable by that spelling)
  if (!v || !*v) return dflt;
  char* end = nullptr;
  const long x = strtol(v, &end, 10);
  return (end && *end == 0) ? (int)x : dflt;
}

struct Events {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  bool make() { return hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess; }
  ~Events() {
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
  }
};

// ---- strategy: the first allocations, as they come ---------------------------------------------------------------------
int place_first(mdc_ctx* c, Arena* a, size_t in_bytes, size_t out_bytes) {
  MDC_HIP(c, hipMalloc(&a->m_in, in_bytes));
  MDC_HIP(c, hipMalloc(&a->m_out, out_bytes));
  return MDC_OK;
}

// ---- strategy: hipMalloc'ed candidates, every pair timed ------------------------------------------------------------------
int place_malloc(mdc_ctx* c, Arena* a, size_t in_bytes, size_t out_bytes, size_t frame_in, int64_t probe_frames, unsigned flags, hipStream_t s,
                 int want, mdc_placed_buffers* r) {
  size_t free_b = 0, total_b = 0;
  MDC_HIP(c, hipMemGetInfo(&free_b, &total_b));
  int K = std::max(1, std::min(want, 8));
  K = (int)std::max<size_t>(1, std::min<size_t>((size_t)K, (size_t)((double)free_b * 0.45) / (in_bytes + out_bytes)));
  // candidates SPREAD over the device's memory: neighbouring allocations are mostly of one class (on some devices the first
  // 60 GB are), so a spacer allocation goes between successive candidate pairs and is given back before anything is timed
  size_t spacer = 0;
  if (K > 1) {
    const double room = (double)free_b * 0.85 - (double)K * (double)(in_bytes + out_bytes);
    const double per = std::min((double)env_int(getenv("MDC_PLACE_SPREAD_MB"), 28000) * 1e6, room / (K - 1));
    spacer = per > 0 ? ((size_t)per >> 21) << 21 : 0;
  }
  std::vector<void*> ins, outs, spacers;
  auto drop = [&](std::vector<void*>& v) {
    for (void* p : v)
      if (p) (void)hipFree(p);
    v.clear();
  };
  int rc = MDC_OK;
  for (int k = 0; k < K && rc == MDC_OK; k++) {  // frames 0, results 0, (spacer,) frames 1, ...: pair (0, 0) = the first allocations
    void *pi = nullptr, *po = nullptr, *sp = nullptr;
    if (hipMalloc(&pi, in_bytes) != hipSuccess || hipMalloc(&po, out_bytes) != hipSuccess) {
      (void)hipGetLastError();
      if (pi) (void)hipFree(pi);
      if (k == 0) rc = fail(c, MDC_ERR_HIP, "mdc_alloc_placed_device: hipMalloc of %zu + %zu bytes failed", in_bytes, out_bytes);
      break;  // fewer candidates than wanted: go on with what there is
    }
    ins.push_back(pi);
    outs.push_back(po);
    if (spacer && k + 1 < K) {
      if (hipMalloc(&sp, spacer) == hipSuccess) spacers.push_back(sp);
      else (void)hipGetLastError(), spacer = 0;
    }
  }
  drop(spacers);
  if (rc != MDC_OK) {
    drop(ins);
    drop(outs);
    return rc;
  }
  K = (int)ins.size();
  int bi = 0, bo = 0;
  float best = 1e30f, first = 0.f;
  Events ev;
  if (K > 1) {
    if (!ev.make()) rc = fail(c, MDC_ERR_HIP, "mdc_alloc_placed_device: hipEventCreate failed");
    for (int k = 0; k < K && rc == MDC_OK; k++)
      if (!fill_noise(ins[k], std::min(in_bytes, (size_t)probe_frames * frame_in) & ~(size_t)3, 0x1234u, s))
        rc = fail(c, MDC_ERR_HIP, "mdc_alloc_placed_device: fill launch failed");
    for (int i = 0; i < K && rc == MDC_OK; i++)
      for (int j = 0; j < K && rc == MDC_OK; j++) {
        float ms = 0.f;
        rc = time_pass(c, (const uint8_t*)ins[i], (float*)outs[j], probe_frames, flags, s, ev.e0, ev.e1, &ms);
        if (rc != MDC_OK) break;
        r->pair_ms[i * K + j] = ms;
        if (i == 0 && j == 0) first = ms;
        if (ms < best) best = ms, bi = i, bo = j;
      }
    (void)hipStreamSynchronize(s);
  }
  if (rc == MDC_OK) {
    a->m_in = ins[bi];
    a->m_out = outs[bo];
    ins[bi] = outs[bo] = nullptr;
    r->candidates_in = r->candidates_out = K;
    r->picked_in = bi;
    r->picked_out = bo;
    r->ms_chosen = K > 1 ? best : 0.f;
    r->ms_first = first;
    snprintf(r->note, sizeof r->note, "hipMalloc: %d x %d candidate buffers (%.1f-GB spacers), pass timed on every pair, picked frames %d / results %d",
             K, K, spacer / 1e9, bi, bo);
  }
  drop(ins);
  drop(outs);
  return rc;
}

// ---- strategy: ranges assembled from classified physical pieces -----------------------------------------------------------
int place_vmm(mdc_ctx* c, Arena* a, size_t in_bytes, size_t out_bytes, size_t frame_in, int64_t probe_frames, unsigned flags, hipStream_t s,
              mdc_placed_buffers* r) {
  Events ev;
  if (!ev.make()) return fail(c, MDC_ERR_HIP, "mdc_alloc_placed_device: hipEventCreate failed");
  {  // beside it: the pass on the first two hipMalloc's, as a caller gets them who takes allocations as they come (where they fit twice)
    size_t free0 = 0, total0 = 0;
    MDC_HIP(c, hipMemGetInfo(&free0, &total0));
    void *fi = nullptr, *fo = nullptr;
    if ((double)(in_bytes + out_bytes) < 0.25 * (double)free0 && hipMalloc(&fi, in_bytes) == hipSuccess && hipMalloc(&fo, out_bytes) == hipSuccess &&
        fill_noise(fi, std::min(in_bytes, (size_t)probe_frames * frame_in) & ~(size_t)3, 0x1234u, s)) {
      const int rc0 = time_pass(c, (const uint8_t*)fi, (float*)fo, probe_frames, flags, s, ev.e0, ev.e1, &r->ms_first);
      (void)hipStreamSynchronize(s);
      if (rc0 != MDC_OK) r->ms_first = 0.f;
    }
    (void)hipGetLastError();
    if (fi) (void)hipFree(fi);
    if (fo) (void)hipFree(fo);
  }
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = c->device;
  size_t gran = 0;
  if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0)
    return fail(c, MDC_ERR_HIP, "mdc_alloc_placed_device: no virtual memory management on this device");
  size_t piece = (size_t)std::max(2, env_int(getenv("MDC_PLACE_PIECE_MIB"), 1024)) << 20;
  piece = (piece + gran - 1) / gran * gran;
  a->piece = piece;
  const size_t n_in = (in_bytes + piece - 1) / piece, n_out = (out_bytes + piece - 1) / piece, need = n_in + n_out;
  size_t free_b = 0, total_b = 0;
  MDC_HIP(c, hipMemGetInfo(&free_b, &total_b));
  const int compose = env_int(getenv("MDC_PLACE_COMPOSE"), 0);  // 0 = frames and results from different classes, 1 = creation order (no
                                                        // classification), 2 = both ranges striped over all classes
  // more pieces than needed, for the choice: twice the need + 4 where there is room (the surplus is returned at the end)
  size_t M = std::min<size_t>(2 * need + 4, (size_t)((double)free_b * 0.9 / (double)piece));
  if (compose == 1) M = std::min(M, need);
  if (M < need) return fail(c, MDC_ERR_HIP, "mdc_alloc_placed_device: %zu + %zu bytes do not fit the device's free memory", in_bytes, out_bytes);
  a->handles.reserve(M);
  for (size_t k = 0; k < M; k++) {
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, piece, &prop, 0) != hipSuccess) {
      (void)hipGetLastError();
      break;
    }
    a->handles.push_back(h);
  }
  M = a->handles.size();
  if (M < need) return fail(c, MDC_ERR_HIP, "mdc_alloc_placed_device: only %zu of %zu pieces of %zu MiB could be created", M, need, piece >> 20);
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  MDC_HIP(c, hipMemAddressReserve(&a->probe_va, M * piece, gran, nullptr, 0));
  for (size_t k = 0; k < M; k++) {
    MDC_HIP(c, hipMemMap(static_cast<char*>(a->probe_va) + k * piece, piece, 0, a->handles[k], 0));
    a->probe_mapped = k + 1;
  }
  MDC_HIP(c, hipMemSetAccess(a->probe_va, M * piece, &acc, 1));
  auto probe_ptr = [&](size_t k) { return static_cast<char*>(a->probe_va) + k * piece; };

  // ---- classes: 0 = the class of piece 0, 1 = the class of the first piece that is fast with piece 0, 2 = fast with both
  std::vector<int> cls(M, 0);
  int n_cls[3] = {(int)M, 0, 0};
  float spread[2] = {0.f, 0.f};
  if (compose != 1 && M >= 3) {
    const size_t rd = piece / 2;  // 1 byte read : 2 bytes written, the path's ratio
    auto split = [&](size_t ref, const std::vector<size_t>& members, std::vector<size_t>& slow, std::vector<size_t>& fast, float* rel) -> bool {
      std::vector<float> t(members.size());
      float lo = 1e30f, hi = 0.f;
      for (size_t q = 0; q < members.size(); q++) {
        if (!time_stream(probe_ptr(ref), rd, probe_ptr(members[q]), piece, s, ev.e0, ev.e1, &t[q])) return false;
        lo = std::min(lo, t[q]);
        hi = std::max(hi, t[q]);
      }
      *rel = lo > 0 ? (hi - lo) / lo : 0.f;
      const float cut = 0.5f * (lo + hi);
      for (size_t q = 0; q < members.size(); q++) ((*rel > 0.03f && t[q] > cut) ? slow : fast).push_back(members[q]);
      return true;
    };
    std::vector<size_t> others, slow0, fast0;
    for (size_t k = 1; k < M; k++) others.push_back(k);
    if (!split(0, others, slow0, fast0, &spread[0])) return fail(c, MDC_ERR_HIP, "mdc_alloc_placed_device: timing a stream failed");
    n_cls[0] = 1 + (int)slow0.size();
    if (!fast0.empty()) {
      const size_t ref1 = fast0[0];
      std::vector<size_t> rest(fast0.begin() + 1, fast0.end()), slow1, fast1;
      if (!rest.empty() && !split(ref1, rest, slow1, fast1, &spread[1])) return fail(c, MDC_ERR_HIP, "mdc_alloc_placed_device: timing a stream failed");
      cls[ref1] = 1;
      for (size_t k : slow1) cls[k] = 1;
      for (size_t k : fast1) cls[k] = 2;
      n_cls[1] = 1 + (int)slow1.size();
      n_cls[2] = (int)fast1.size();
    }
    (void)hipStreamSynchronize(s);
  }
  // ---- which pieces make up which range
  std::vector<size_t> by_cls[3];
  for (size_t k = 0; k < M; k++) by_cls[cls[k]].push_back(k);
  std::vector<size_t> pin, pout;
  int cin = 0, cout = 0;
  if (compose == 1 || n_cls[0] == (int)M) {  // creation order (no structure found: every piece is as good as any other)
    for (size_t k = 0; k < n_in; k++) pin.push_back(k);
    for (size_t k = 0; k < n_out; k++) pout.push_back(n_in + k);
  } else if (compose == 2) {  // both ranges striped over the classes, piece by piece
    size_t at[3] = {0, 0, 0};
    int turn = 0;
    auto take = [&]() -> size_t {
      for (int q = 0; q < 3; q++) {
        const int k = (turn + q) % 3;
        if (at[k] < by_cls[k].size()) {
          turn = (k + 1) % 3;
          return by_cls[k][at[k]++];
        }
      }
      return 0;
    };
    for (size_t k = 0; k < n_out; k++) pout.push_back(take());
    for (size_t k = 0; k < n_in; k++) pin.push_back(take());
  } else {
    // frames <- class X, results <- class Y != X; what a class cannot cover comes from the third class, then from anywhere.
    // The best (X, Y) covers the most pieces without putting the two ranges into one class.
    size_t best_cov = 0;
    for (int x = 0; x < 3; x++)
      for (int y = 0; y < 3; y++) {
        if (x == y) continue;
        const int z = 3 - x - y;
        const size_t a_in = std::min(n_in, by_cls[x].size()), a_out = std::min(n_out, by_cls[y].size());
        const size_t cov = a_in + a_out + std::min((n_in - a_in) + (n_out - a_out), by_cls[z].size());
        if (cov > best_cov) best_cov = cov, cin = x, cout = y;
      }
    size_t at[3] = {0, 0, 0};
    auto take_from = [&](int k) -> long {
      return at[k] < by_cls[k].size() ? (long)by_cls[k][at[k]++] : -1L;
    };
    const int cz = 3 - cin - cout;
    for (size_t k = 0; k < n_out; k++) {
      long p = take_from(cout);
      if (p < 0) p = take_from(cz);
      if (p < 0) p = take_from(cin);
      pout.push_back((size_t)p);
    }
    for (size_t k = 0; k < n_in; k++) {
      long p = take_from(cin);
      if (p < 0) p = take_from(cz);
      if (p < 0) p = take_from(cout);
      pin.push_back((size_t)p);
    }
  }
  // ---- the two ranges: every piece mapped a second time, in its final order (the probe mappings stay: nothing is ever unmapped
  // while the arena lives)
  a->in_pieces = n_in;
  a->out_pieces = n_out;
  MDC_HIP(c, hipMemAddressReserve(&a->in_va, n_in * piece, gran, nullptr, 0));
  MDC_HIP(c, hipMemAddressReserve(&a->out_va, n_out * piece, gran, nullptr, 0));
  for (size_t k = 0; k < n_in; k++) {
    MDC_HIP(c, hipMemMap(static_cast<char*>(a->in_va) + k * piece, piece, 0, a->handles[pin[k]], 0));
    a->in_mapped = k + 1;
  }
  for (size_t k = 0; k < n_out; k++) {
    MDC_HIP(c, hipMemMap(static_cast<char*>(a->out_va) + k * piece, piece, 0, a->handles[pout[k]], 0));
    a->out_mapped = k + 1;
  }
  MDC_HIP(c, hipMemSetAccess(a->in_va, n_in * piece, &acc, 1));
  MDC_HIP(c, hipMemSetAccess(a->out_va, n_out * piece, &acc, 1));
  // the pass on the pair that is handed out
  if (!fill_noise(a->in_va, std::min(in_bytes, (size_t)probe_frames * frame_in) & ~(size_t)3, 0x1234u, s))
    return fail(c, MDC_ERR_HIP, "mdc_alloc_placed_device: fill launch failed");
  int rc = time_pass(c, (const uint8_t*)a->in_va, (float*)a->out_va, probe_frames, flags, s, ev.e0, ev.e1, &r->ms_chosen);
  (void)hipStreamSynchronize(s);
  if (rc != MDC_OK) return rc;
  int used[3] = {0, 0, 0}, used_out[3] = {0, 0, 0};
  for (size_t p : pin) used[cls[p]]++;
  for (size_t p : pout) used_out[cls[p]]++;
  r->pieces = (int)M;
  r->piece_mib = (int)(piece >> 20);
  for (int k = 0; k < 3; k++) r->class_count[k] = n_cls[k];
  r->candidates_in = r->candidates_out = 1;
  snprintf(r->note, sizeof r->note,
           "assembled: %zu pieces of %zu MiB mapped once; classes by a timed read/write stream against a reference piece: %d / %d / %d "
           "(spread %.1f %%, %.1f %%); frames <- %d+%d+%d pieces of classes 0/1/2, results <- %d+%d+%d%s",
           M, piece >> 20, n_cls[0], n_cls[1], n_cls[2], spread[0] * 100, spread[1] * 100, used[0], used[1], used[2], used_out[0], used_out[1],
           used_out[2], compose == 1 ? " (creation order)" : compose == 2 ? " (striped)" : "");
  return MDC_OK;
}

}  // namespace
}  // namespace mdc

using namespace mdc;

extern "C" {

int mdc_alloc_placed_device(mdc_ctx* c, size_t in_bytes, size_t out_bytes, int64_t nframes, unsigned flags, int strategy, void* stream,
                            mdc_placed_buffers* out) try {
  if (!c) return MDC_ERR_ARG;
  if (!out || nframes <= 0 || strategy < MDC_PLACE_AUTO || strategy > MDC_PLACE_VMM) return fail(c, MDC_ERR_ARG, "mdc_alloc_placed_device: bad argument");
  memset(out, 0, sizeof *out);
  ReadLock lk(c->mu);
  DeviceGuard dg(c->device);
  // the pass that will run on the buffers: its frame sizes say how large they must be
  const bool rect = (flags & MDC_RECTIFY) != 0;
  if (rect && !c->valid_remap) return fail(c, MDC_ERR_STATE, "mdc_alloc_placed_device: no remap set (UndistorterFOV invalid)");
  const int fw = rect ? c->rm_in_w : (c->in_w > 0 ? c->in_w : c->rm_in_w), fh = rect ? c->rm_in_h : (c->in_h > 0 ? c->in_h : c->rm_in_h);
  if (fw <= 0 || fh <= 0) return fail(c, MDC_ERR_STATE, "mdc_alloc_placed_device: frame size unknown: set the photometric tables or a remap first");
  const size_t frame_in = (size_t)fw * fh, frame_out = rect ? (size_t)c->out_w * c->out_h * 4 : frame_in * 4;
  if (in_bytes == 0) in_bytes = (size_t)nframes * frame_in;
  if (out_bytes == 0) out_bytes = (size_t)nframes * frame_out;
  if (in_bytes < (size_t)nframes * frame_in || out_bytes < (size_t)nframes * frame_out)
    return fail(c, MDC_ERR_SIZE, "mdc_alloc_placed_device: %lld frames need %zu + %zu bytes, %zu + %zu given", (long long)nframes,
                (size_t)nframes * frame_in, (size_t)nframes * frame_out, in_bytes, out_bytes);
  in_bytes = (in_bytes + 255) & ~(size_t)255;
  out_bytes = (out_bytes + 255) & ~(size_t)255;
  out->nframes = nframes;
  out->in_bytes = in_bytes;
  out->out_bytes = out_bytes;
  const int64_t probe_frames = std::min<int64_t>(nframes, 4096);
  out->probe_frames = probe_frames;
  if (strategy == MDC_PLACE_AUTO) {
    const char* e = getenv("MDC_PLACEMENT");
    if (e && !strcmp(e, "first")) strategy = MDC_PLACE_FIRST;
    else if (e && !strcmp(e, "malloc")) strategy = MDC_PLACE_MALLOC;
    else if (e && !strcmp(e, "vmm")) strategy = MDC_PLACE_VMM;
    else strategy = MDC_PLACE_DEFAULT;
    // batches that fit the 256-MiB Infinity Cache several times over do not see HBM placement: no search
    if (in_bytes + out_bytes < ((size_t)1 << 30)) strategy = MDC_PLACE_FIRST;
  }
  Arena* a = new Arena();
  a->device = c->device;
  a->strategy = strategy;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (strategy == MDC_PLACE_VMM) rc = place_vmm(c, a, in_bytes, out_bytes, frame_in, probe_frames, flags, s, out);
  else if (strategy == MDC_PLACE_MALLOC) rc = place_malloc(c, a, in_bytes, out_bytes, frame_in, probe_frames, flags, s, std::max(1, env_int(getenv("MDC_PLACE_CANDIDATES"), 6)), out);
  else rc = place_first(c, a, in_bytes, out_bytes);
  if (rc != MDC_OK) {
    release_arena(a);
    memset(out, 0, sizeof *out);
    return rc;
  }
  out->strategy = strategy;
  out->d_in = static_cast<uint8_t*>(a->in_va ? a->in_va : a->m_in);
  out->d_out = static_cast<float*>(a->out_va ? a->out_va : a->m_out);
  if (strategy == MDC_PLACE_FIRST) {
    out->candidates_in = out->candidates_out = 1;
    snprintf(out->note, sizeof out->note, "first allocations, as they come (hipMalloc)");
  }
  out->handle = a;
  return MDC_OK;
} MDC_CATCH(c)

int mdc_free_placed_device(mdc_ctx* c, mdc_placed_buffers* b) {
  if (!c || !b) return MDC_ERR_ARG;
  if (!b->handle) return MDC_OK;
  Arena* a = static_cast<Arena*>(b->handle);
  b->handle = nullptr;
  b->d_in = nullptr;
  b->d_out = nullptr;
  release_arena(a);
  return MDC_OK;
}

}  // extern "C"
