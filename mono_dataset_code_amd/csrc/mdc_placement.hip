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
#include "placement_classes.h"

namespace mdc {
namespace {

constexpr int MDC_PLACE_DEFAULT = MDC_PLACE_VMM;  // what MDC_PLACE_AUTO means (DESIGN.md section 6.1 has the measurements behind it); falls back to
                                                  // MDC_PLACE_MALLOC where the device has no virtual memory management

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
  std::vector<void*> plain;  // a set's plain allocations
  // assembled ranges
  std::vector<hipMemGenericAllocationHandle_t> handles;
  size_t piece = 0;
  void* probe_va = nullptr;  // handles.size() pieces, creation order
  void* in_va = nullptr;
  void* out_va = nullptr;
  std::vector<std::pair<void*, size_t>> mapped;    // every hipMemMap that succeeded (address, size): undone one by one at the end
  std::vector<std::pair<void*, size_t>> reserved;  // every hipMemAddressReserve
};

void release_arena(Arena* a) {
  if (!a) return;
  DeviceGuard dg(a->device);
  (void)hipDeviceSynchronize();  // nothing may still touch a range that is about to lose its pages
  if (a->m_in) (void)hipFree(a->m_in);
  if (a->m_out) (void)hipFree(a->m_out);
  for (void* p : a->plain) (void)hipFree(p);
  for (size_t k = a->mapped.size(); k-- > 0;) (void)hipMemUnmap(a->mapped[k].first, a->mapped[k].second);
  // The address ranges are NOT given back (hipMemAddressFree): ROCm 7.2 serves stale translations for a range that is reserved and
  // mapped again after hipMemUnmap -- kernels then read and write the physical pages of the PREVIOUS mapping (tools/vmm_offset_check.hip:
  // 650 of 768 pages wrong in the second round; with the reservations kept, none in any round; round 5's "intermittent memory faults"
  // of churned candidates, profiles/r05_experiments/07_*, were this).  No address of an arena is ever mapped twice; what a process
  // spends is address space only (tens of GiB per allocation of a 47-bit space: thousands of allocations), and when a reservation
  // fails the allocator falls back to plain allocations.
  for (hipMemGenericAllocationHandle_t h : a->handles)
    if (h) (void)hipMemRelease(h);  // after the last mapping is gone
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
// What the classes are (profiles/r06_experiments/02_*): a device's memory falls into THREE classes of a third of its size each (244 pieces
// of 1 GiB: 86 / 84 / 74), handed out by the driver in runs of whole GiB -- the signature of the three ranks (stack IDs) of a 12-high
// HBM3E stack selected by high physical address bits.  Ranks share a channel's data bus but have their own banks: the hundreds of
// row-sized streams of a launch (a workgroup's output rows, its source windows) conflict in the banks of ONE rank far more often than
// when they are spread over all three.  Hence the composition: EVERY range is striped over all classes in equal shares, piece by piece
// (512 MiB: less than the part of a range a launch works on at one time; measured: 0.66-0.67 of 8 TB/s for the headline against 0.63-0.64
// for the best pair of plain allocations and 0.58-0.60 for the first pair; frames in one class and results in another: 0.63).
struct Range {
  size_t bytes = 0, pieces = 0;
  void* va = nullptr;
  std::vector<size_t> piece_ids;
};

int assemble_ranges(mdc_ctx* c, Arena* a, std::vector<Range>& ranges, hipStream_t s, Events& ev, int* n_cls_out, int* pieces_out, int* piece_mib_out,
                    char* note, size_t note_cap) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = c->device;
  size_t gran = 0;
  if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) {
    (void)hipGetLastError();
    return fail(c, MDC_ERR_HIP, "mdc_alloc_placed_device: no virtual memory management on this device");
  }
  // pieces of 2 GiB ended in GPU memory faults (profiles/r06_experiments/02_*): 1 GiB at most
  size_t piece = (size_t)std::min(1024, std::max(2, env_int(getenv("MDC_PLACE_PIECE_MIB"), 512))) << 20;
  piece = (piece + gran - 1) / gran * gran;
  a->piece = piece;
  size_t need = 0;
  for (Range& r : ranges) {
    r.pieces = (r.bytes + piece - 1) / piece;
    need += r.pieces;
  }
  const int compose = env_int(getenv("MDC_PLACE_COMPOSE"), 0);  // 0 = every range striped over all classes (the product), 1 = creation order,
                                                                // no classification, 2 = range k in class k mod 3 (diagnosis: the r06 experiments)
  size_t free_b = 0, total_b = 0;
  MDC_HIP(c, hipMemGetInfo(&free_b, &total_b));
  const size_t cap = (size_t)((double)free_b * 0.92 / (double)piece);  // pieces the device has room for
  if (cap < need) return fail(c, MDC_ERR_HIP, "mdc_alloc_placed_device: %zu pieces of %zu MiB do not fit the device's free memory", need, piece >> 20);
  // how far the scan for all three classes may go: on some devices the third class only shows after 125 GiB of allocations
  // (profiles/r05_experiments/11_*), on others 145 of the first 160 GiB are ONE class (session r06r); what is not used goes back at once
  const size_t scan = ((size_t)std::max(0, env_int(getenv("MDC_PLACE_SCAN_GIB"), 256)) << 30) / piece;
  const size_t max_pieces = compose == 1 ? need : std::min(cap, std::max(4 * need + 24, scan));
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  std::vector<char*> piece_va;  // where piece k is mapped for the probes: batches of whole groups, every batch a reservation of its own
  auto probe_ptr = [&](size_t k) { return piece_va[k]; };
  auto grow = [&](size_t upto) -> int {  // more pieces, each mapped once for the probes
    const size_t from = a->handles.size();
    if (upto <= from) return MDC_OK;
    void* base = nullptr;
    MDC_HIP(c, hipMemAddressReserve(&base, (upto - from) * piece, gran, nullptr, 0));
    a->reserved.push_back({base, (upto - from) * piece});
    if (!a->probe_va) a->probe_va = base;
    while (a->handles.size() < upto) {
      hipMemGenericAllocationHandle_t h;
      if (hipMemCreate(&h, piece, &prop, 0) != hipSuccess) {
        (void)hipGetLastError();
        break;
      }
      a->handles.push_back(h);
      char* at = static_cast<char*>(base) + (a->handles.size() - 1 - from) * piece;
      piece_va.push_back(at);
      MDC_HIP(c, hipMemMap(at, piece, 0, h, 0));
      a->mapped.push_back({at, piece});
    }
    if (a->handles.size() > from) MDC_HIP(c, hipMemSetAccess(base, (a->handles.size() - from) * piece, &acc, 1));
    return MDC_OK;
  };
  // ---- classes: 0 = the class of piece 0, 1 = the class of the first piece that is fast with piece 0, 2 = fast with both.
  // t0[k] = a linear stream reading piece 0 and writing piece k; t1[k] = the same against the first piece of class 1.  A set of times is
  // cut at its widest gap (if that is wide enough to be a gap at all).
  // The unit that is timed is a GROUP of consecutive pieces of 1 GiB in all (pieces made one after the other come out of the same block of
  // the driver's): a stream over less than that partly lives in the 256-MiB Infinity Cache and the classes blur (clusters 3-4 % apart
  // for 512-MiB units, noise for 256-MiB ones, against 5-9 % for 1 GiB: profiles/r06_experiments/02_*).
  const size_t G = std::max<size_t>(1, ((size_t)1 << 30) / piece), group_bytes = G * piece;
  std::vector<float> t0, t1;  // per group
  std::vector<int> cls;       // per piece
  long ref1 = -1;             // group: the second reference (the first group that is fast with the first one)
  float gap[2] = {0.f, 0.f};
  const size_t rd = group_bytes / 2;  // 1 byte read : 2 bytes written, the path's ratio
  auto group_ptr = [&](size_t g) { return probe_ptr(g * G); };
  auto timed = [&](size_t ref, size_t g, float* ms) -> bool {  // the stream, once more if the figure is out of any class's range (a hiccup)
    if (!time_stream(group_ptr(ref), rd, group_ptr(g), group_bytes, s, ev.e0, ev.e1, ms)) return false;
    float lo = *ms;
    for (float x : t0)
      if (x > 0) lo = std::min(lo, x);
    for (int again = 0; again < 2 && *ms > 1.12f * lo; again++)
      if (!time_stream(group_ptr(ref), rd, group_ptr(g), group_bytes, s, ev.e0, ev.e1, ms)) return false;
    return true;
  };
  size_t ref0 = 0;  // group
  size_t ref0_tried = 0, ref1_tried = 0;
  float ref0_rel = -1.f, ref1_rel = -1.f;
  std::vector<char> ref1_was(0);
  auto classify = [&]() -> int {
    const size_t NG = a->handles.size() / G;  // whole groups (grow() makes pieces in whole groups)
    // The reference: one of the first eight groups, the one that splits the groups seen so far best -- tried three per call until one
    // splits them cleanly (clusters 5.5 % apart).  A group whose pieces the driver took from two blocks straddles two classes; against
    // such a reference everything looks half slow and the split is weak or missing.
    for (size_t g = t0.size(); g < NG; g++) {  // the groups that are new since the last call, against the reference of the moment
      float ms = 0.f;
      if (g != ref0 && !timed(ref0, g, &ms)) return fail(c, MDC_ERR_HIP, "mdc_alloc_placed_device: timing a stream failed");
      t0.push_back(ms);
    }
    if (ref0_tried > 0) {  // how the reference of the moment splits what is there now
      std::vector<float> v;
      for (size_t g = 0; g < NG; g++)
        if (g != ref0) v.push_back(t0[g]);
      const float cut = placement_cut(v, &ref0_rel);
      size_t slow = 0;
      for (float x : v) slow += (std::isfinite(cut) && x > cut) ? 1 : 0;
      if (slow * 3 > v.size() * 2) ref0_rel *= 0.4f;  // (the same plausibility rule as for the candidates below)
    }
    // (no structure at all -- everything seen so far is one class -- is no reason to doubt the reference: wait for more groups)
    const bool doubt = ref0_tried == 0 || (ref0_rel >= 0.025f && ref0_rel < 0.055f);
    for (int tries = 0; NG >= 4 && doubt && tries < 3 && ref0_rel < 0.055f && ref0_tried < std::min<size_t>(NG, 8); tries++) {
      const size_t r = ref0_tried++;
      std::vector<float> t(NG, 0.f), v;
      for (size_t g = 0; g < NG; g++)
        if (g != r) {
          if (r == ref0 && g < t0.size() && t0[g] > 0) t[g] = t0[g];  // (already measured against this one)
          else if (!timed(r, g, &t[g])) return fail(c, MDC_ERR_HIP, "mdc_alloc_placed_device: timing a stream failed");
          v.push_back(t[g]);
        }
      float rel = 0.f;
      const float cut = placement_cut(v, &rel);
      // A reference's own class is a third of the device: a split that puts more than two thirds of the groups into the slow cluster is
      // an odd reference (a group of scattered pages that is slow with almost everything), however far apart its clusters are.
      size_t slow = 0;
      for (float x : v) slow += (std::isfinite(cut) && x > cut) ? 1 : 0;
      if (slow * 3 > v.size() * 2) rel *= 0.4f;
      if (rel > ref0_rel || (r == ref0 && t0.size() < NG)) {
        if (r != ref0) {  // another first reference: the second level starts over
          t1.clear();
          ref1 = -1;
          ref1_tried = 0;
          ref1_rel = -1.f;
        }
        ref0_rel = rel, ref0 = r, t0 = t;
      }
    }
    cls.assign(a->handles.size(), 0);
    if (NG < 2) return MDC_OK;
    std::vector<size_t> todo;
    std::vector<int> gcls = placement_classes(t0, ref0, t1, &ref1, gap, &todo);  // which fast groups still need their time against the second reference
    if (!todo.empty()) {
      t1.resize(NG, -1.f);
      for (size_t g : todo)
        if (!timed((size_t)ref1, g, &t1[g])) return fail(c, MDC_ERR_HIP, "mdc_alloc_placed_device: timing a stream failed");
      gcls = placement_classes(t0, ref0, t1, &ref1, gap, nullptr);
    }
    // the second reference likewise: while the fast groups do not fall apart cleanly, another of them is tried (two per call, six in all)
    std::vector<size_t> fast;
    for (size_t g = 0; g < NG; g++)
      if (gcls[g] != 0) fast.push_back(g);
    ref1_rel = std::max(ref1_rel, gap[1]);
    ref1_was.resize(NG, 0);
    if (ref1 >= 0) ref1_was[(size_t)ref1] = 1;
    for (int tries = 0; fast.size() >= 4 && tries < 2 && ref1_rel < 0.055f && ref1_tried < 6; tries++) {
      long cand = -1;
      for (size_t g : fast)
        if (!ref1_was[g]) {
          cand = (long)g;
          break;
        }
      if (cand < 0) break;
      ref1_was[(size_t)cand] = 1;
      ref1_tried++;
      std::vector<float> t(NG, -1.f), v;
      for (size_t g : fast)
        if ((long)g != cand) {
          if (!timed((size_t)cand, g, &t[g])) return fail(c, MDC_ERR_HIP, "mdc_alloc_placed_device: timing a stream failed");
          v.push_back(t[g]);
        }
      float rel = 0.f;
      (void)placement_cut(v, &rel);
      if (rel > ref1_rel) {
        ref1_rel = rel, ref1 = cand, t1 = t;
        gcls = placement_classes(t0, ref0, t1, &ref1, gap, nullptr);
      }
    }
    for (size_t k = 0; k < NG * G; k++) cls[k] = gcls[k / G];
    return MDC_OK;
  };
  auto whole_groups = [&](size_t n) { return std::min(max_pieces / G * G, (n + G - 1) / G * G); };
  int rc = grow(compose == 1 ? need : std::max(need, whole_groups(need + std::max<size_t>(need / 2, 6))));
  if (rc != MDC_OK) return rc;
  if (a->handles.size() < need) return fail(c, MDC_ERR_HIP, "mdc_alloc_placed_device: only %zu of %zu pieces of %zu MiB could be created", a->handles.size(), need, piece >> 20);
  int n_cls[3] = {(int)a->handles.size(), 0, 0};
  const size_t share = (need + 2) / 3;  // pieces wanted of every class
  if (compose != 1) {
    for (;;) {
      if ((rc = classify()) != MDC_OK) return rc;
      n_cls[0] = n_cls[1] = n_cls[2] = 0;
      for (int k : cls) n_cls[k]++;
      if (env_int(getenv("MDC_PLACE_DEBUG"), 0)) {
        fprintf(stderr, "mdc placement: %zu pieces, classes %d / %d / %d, separation %.1f %% / %.1f %%; ms against piece 0:", a->handles.size(), n_cls[0], n_cls[1], n_cls[2],
                gap[0] * 100, gap[1] * 100);
        for (size_t k = 0; k < t0.size(); k++) fprintf(stderr, " %.4f", t0[k]);  // (per group of 1 GiB)
        fprintf(stderr, "\n");
      }
      const bool balanced = (size_t)std::min(n_cls[0], std::min(n_cls[1], n_cls[2])) >= share;
      if (balanced || a->handles.size() >= max_pieces) break;
      const size_t before = a->handles.size();
      if ((rc = grow(std::max(before, whole_groups(before + std::max<size_t>(need / 2, 8))))) != MDC_OK) return rc;
      if (a->handles.size() == before) break;  // the device gave what it had
    }
    (void)hipStreamSynchronize(s);
  }
  const size_t M = a->handles.size();
  cls.resize(M, 0);
  // ---- which pieces make up which range
  // The pieces of a class, the surest first: a group whose time lies at its cluster's centre before one at its edge (a piece the driver
  // put together from several blocks straddles two classes and shows an in-between time).
  std::vector<size_t> by_cls[3];
  {
    const size_t NG = M / G;
    std::vector<float> centre_t[3];
    auto group_time = [&](size_t g) -> float {  // the time that decided the group's class
      const int cg = cls[g * G];
      if (cg == 0) return g < t0.size() ? t0[g] : 0.f;
      return (g < t1.size() && t1[g] >= 0) ? t1[g] : -1.f;  // (the second reference itself: no time, sure by definition)
    };
    for (size_t g = 0; g < NG; g++)
      if (g != ref0 && group_time(g) >= 0) centre_t[cls[g * G]].push_back(group_time(g));
    float centre[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < 3; k++)
      if (!centre_t[k].empty()) {
        std::sort(centre_t[k].begin(), centre_t[k].end());
        centre[k] = centre_t[k][centre_t[k].size() / 2];
      }
    std::vector<std::pair<float, size_t>> order[3];  // (distance from the centre, group)
    for (size_t g = 0; g < NG; g++) {
      const int cg = cls[g * G];
      const float t = group_time(g);
      order[cg].push_back({(g == ref0 || t < 0) ? 0.f : std::fabs(t - centre[cg]), g});
    }
    for (int k = 0; k < 3; k++) {
      std::stable_sort(order[k].begin(), order[k].end(), [](const std::pair<float, size_t>& x, const std::pair<float, size_t>& y) { return x.first < y.first; });
      for (const auto& e : order[k])
        for (size_t q = 0; q < G; q++) by_cls[k].push_back(e.second * G + q);
    }
    for (size_t k = NG * G; k < M; k++) by_cls[cls[k]].push_back(k);  // (pieces beyond the last whole group: creation order, no classification)
  }
  size_t at[3] = {0, 0, 0};
  auto take = [&](int& turn) -> size_t {  // the next piece of class `turn`, or of the class after it that still has one
    for (int q = 0; q < 3; q++) {
      const int k = (turn + q) % 3;
      if (at[k] < by_cls[k].size()) {
        turn = (k + 1) % 3;
        return by_cls[k][at[k]++];
      }
    }
    return 0;
  };
  if (compose == 1) {
    size_t next = 0;
    for (Range& r : ranges)
      for (size_t k = 0; k < r.pieces; k++) r.piece_ids.push_back(next++);
  } else if (compose == 2) {  // diagnosis: a whole range in one class (range k: class k mod 3), the rest from wherever
    for (size_t i = 0; i < ranges.size(); i++)
      for (size_t k = 0; k < ranges[i].pieces; k++) {
        int turn = (int)(i % 3);
        ranges[i].piece_ids.push_back(take(turn));
      }
  } else {  // piece by piece round the classes, all ranges in step (placement_classes.h)
    std::vector<size_t> want;
    for (const Range& r : ranges) want.push_back(r.pieces);
    const std::vector<std::vector<size_t>> ids = placement_compose(by_cls, want);
    for (size_t i = 0; i < ranges.size(); i++) ranges[i].piece_ids = ids[i];
  }
  // ---- the surplus goes back to the device before anything else is mapped (MDC_PLACE_KEEP_SURPLUS=1: stays mapped until the end).
  // Only pieces no range uses are touched, after everything that ever ran on them has finished.
  std::vector<char> used(M, 0);
  for (const Range& r : ranges)
    for (size_t id : r.piece_ids) used[id] = 1;
  size_t returned = 0;
  if (!env_int(getenv("MDC_PLACE_KEEP_SURPLUS"), 0)) {
    MDC_HIP(c, hipDeviceSynchronize());
    for (size_t k = 0; k < M; k++)
      if (!used[k]) {
        for (size_t q = 0; q < a->mapped.size(); q++)
          if (a->mapped[q].first == (void*)probe_ptr(k)) {
            (void)hipMemUnmap(a->mapped[q].first, a->mapped[q].second);
            a->mapped.erase(a->mapped.begin() + (long)q);
            break;
          }
        (void)hipMemRelease(a->handles[k]);
        a->handles[k] = nullptr;
        returned++;
      }
    MDC_HIP(c, hipDeviceSynchronize());
  }
  // ---- the ranges: every piece mapped a second time, in its final place (the probe mappings of the pieces in use stay: no address a
  // kernel may still know is ever unmapped while the arena lives).  Whole pieces: hipMemMap refuses an offset into a handle
  // (hipErrorInvalidValue on ROCm 7.2, as on CUDA), so a range's classes alternate piece by piece.
  for (Range& r : ranges) {
    MDC_HIP(c, hipMemAddressReserve(&r.va, r.pieces * piece, gran, nullptr, 0));
    a->reserved.push_back({r.va, r.pieces * piece});
    for (size_t t = 0; t < r.piece_ids.size(); t++) {
      char* where = static_cast<char*>(r.va) + t * piece;
      MDC_HIP(c, hipMemMap(where, piece, 0, a->handles[r.piece_ids[t]], 0));
      a->mapped.push_back({where, piece});
    }
    MDC_HIP(c, hipMemSetAccess(r.va, r.pieces * piece, &acc, 1));
  }
  MDC_HIP(c, hipDeviceSynchronize());
  for (int k = 0; k < 3; k++) n_cls_out[k] = n_cls[k];
  *pieces_out = (int)M;
  *piece_mib_out = (int)(piece >> 20);
  int used_cls[3] = {0, 0, 0};
  for (size_t k = 0; k < M; k++)
    if (used[k]) used_cls[cls[k]]++;
  snprintf(note, note_cap,
           "assembled: %zu ranges from %zu pieces of %zu MiB (%zu created, %zu returned); memory classes by a timed read/write stream against reference "
           "pieces: %d / %d / %d (clusters %.1f %%, %.1f %% apart), in use %d / %d / %d; %s",
           ranges.size(), M - returned, piece >> 20, M, returned, n_cls[0], n_cls[1], n_cls[2], gap[0] * 100, gap[1] * 100, used_cls[0], used_cls[1], used_cls[2],
           compose == 1 ? "creation order, no classification" : compose == 2 ? "a range in ONE class (diagnosis)" : "every range striped over the classes");
  return MDC_OK;
}

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
  std::vector<Range> ranges(2);
  ranges[0].bytes = in_bytes;
  ranges[1].bytes = out_bytes;
  int rc = assemble_ranges(c, a, ranges, s, ev, r->class_count, &r->pieces, &r->piece_mib, r->note, sizeof r->note);
  if (rc != MDC_OK) return rc;
  a->in_va = ranges[0].va;
  a->out_va = ranges[1].va;
  // the pass on the pair that is handed out
  if (!fill_noise(a->in_va, std::min(in_bytes, (size_t)probe_frames * frame_in) & ~(size_t)3, 0x1234u, s))
    return fail(c, MDC_ERR_HIP, "mdc_alloc_placed_device: fill launch failed");
  rc = time_pass(c, (const uint8_t*)a->in_va, (float*)a->out_va, probe_frames, flags, s, ev.e0, ev.e1, &r->ms_chosen);
  (void)hipStreamSynchronize(s);
  if (rc != MDC_OK) return rc;
  r->candidates_in = r->candidates_out = 1;
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
  const bool asked_auto = strategy == MDC_PLACE_AUTO;
  if (strategy == MDC_PLACE_AUTO) {
    const char* e = getenv("MDC_PLACEMENT");
    if (e && !strcmp(e, "first")) strategy = MDC_PLACE_FIRST;
    else if (e && !strcmp(e, "malloc")) strategy = MDC_PLACE_MALLOC;
    else if (e && !strcmp(e, "vmm")) strategy = MDC_PLACE_VMM;
    else strategy = MDC_PLACE_DEFAULT;
    // batches that fit the 256-MiB Infinity Cache several times over do not see HBM placement: no search
    if (in_bytes + out_bytes < ((size_t)1 << 30)) strategy = MDC_PLACE_FIRST;
  }
  const bool by_default = strategy == MDC_PLACE_DEFAULT && !getenv("MDC_PLACEMENT");
  Arena* a = new Arena();
  a->device = c->device;
  a->strategy = strategy;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (strategy == MDC_PLACE_VMM) {
    rc = place_vmm(c, a, in_bytes, out_bytes, frame_in, probe_frames, flags, s, out);
    if (rc != MDC_OK && by_default && asked_auto) {  // the library's own choice did not work here: candidates instead
      release_arena(a);
      a = new Arena();
      a->device = c->device;
      a->strategy = strategy = MDC_PLACE_MALLOC;
      const int64_t nf = out->nframes, pf = out->probe_frames;
      memset(out, 0, sizeof *out);
      out->nframes = nf, out->probe_frames = pf, out->in_bytes = in_bytes, out->out_bytes = out_bytes;
      rc = place_malloc(c, a, in_bytes, out_bytes, frame_in, probe_frames, flags, s, std::max(1, env_int(getenv("MDC_PLACE_CANDIDATES"), 6)), out);
    }
  }
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

// Buffers of a step with more outputs than one result per frame (config 5's pyramid levels, the DSO hand-off's gradient images): every
// one of them striped over the memory classes by the same machinery.
int mdc_alloc_striped_set_device(mdc_ctx* c, int n, const size_t* bytes, void* stream, mdc_striped_set* out) try {
  if (!c) return MDC_ERR_ARG;
  if (!out || !bytes || n < 1 || n > MDC_STRIPED_SET_MAX) return fail(c, MDC_ERR_ARG, "mdc_alloc_striped_set_device: bad argument");
  memset(out, 0, sizeof *out);
  for (int k = 0; k < n; k++)
    if (bytes[k] == 0) return fail(c, MDC_ERR_ARG, "mdc_alloc_striped_set_device: buffer %d has no size", k);
  ReadLock lk(c->mu);
  DeviceGuard dg(c->device);
  Arena* a = new Arena();
  a->device = c->device;
  a->strategy = MDC_PLACE_VMM;
  const char* e = getenv("MDC_PLACEMENT");
  int rc = MDC_ERR_HIP;
  if (!(e && (!strcmp(e, "first") || !strcmp(e, "malloc")))) {
    Events ev;
    // buffers below a piece's size share ONE range (each at a 2-MiB boundary): a range takes whole pieces
    const size_t piece_guess = (size_t)std::min(1024, std::max(2, env_int(getenv("MDC_PLACE_PIECE_MIB"), 512))) << 20, align = (size_t)2 << 20;
    std::vector<Range> ranges;
    std::vector<std::pair<size_t, size_t>> where((size_t)n);  // buffer k = range index, offset
    size_t small_total = 0;
    long small_range = -1;
    for (int k = 0; k < n; k++) {
      if (bytes[k] >= piece_guess) {
        where[(size_t)k] = {ranges.size(), 0};
        ranges.push_back(Range());
        ranges.back().bytes = bytes[k];
      } else {
        if (small_range < 0) {
          small_range = (long)ranges.size();
          ranges.push_back(Range());
        }
        where[(size_t)k] = {(size_t)small_range, small_total};
        small_total += (bytes[k] + align - 1) / align * align;
      }
    }
    if (small_range >= 0) ranges[(size_t)small_range].bytes = small_total;
    rc = ev.make() ? assemble_ranges(c, a, ranges, (hipStream_t)stream, ev, out->class_count, &out->pieces, &out->piece_mib, out->note, sizeof out->note)
                   : fail(c, MDC_ERR_HIP, "mdc_alloc_striped_set_device: hipEventCreate failed");
    if (rc == MDC_OK)
      for (int k = 0; k < n; k++) out->d_ptr[k] = static_cast<char*>(ranges[where[(size_t)k].first].va) + where[(size_t)k].second;
  }
  if (rc != MDC_OK) {  // no virtual memory management here (or switched off): plain allocations, as they come
    release_arena(a);
    a = new Arena();
    a->device = c->device;
    a->strategy = MDC_PLACE_FIRST;
    rc = MDC_OK;
    for (int k = 0; k < n && rc == MDC_OK; k++) {
      void* p = nullptr;
      if (hipMalloc(&p, bytes[k]) != hipSuccess) rc = fail(c, MDC_ERR_HIP, "mdc_alloc_striped_set_device: hipMalloc of %zu bytes failed", bytes[k]);
      else a->plain.push_back(p), out->d_ptr[k] = p;
    }
    if (rc != MDC_OK) {
      release_arena(a);
      memset(out, 0, sizeof *out);
      return rc;
    }
    snprintf(out->note, sizeof out->note, "plain allocations, as they come (hipMalloc)");
  }
  out->n = n;
  for (int k = 0; k < n; k++) out->bytes[k] = bytes[k];
  out->strategy = a->strategy;
  out->handle = a;
  return MDC_OK;
} MDC_CATCH(c)

int mdc_free_striped_set_device(mdc_ctx* c, mdc_striped_set* b) {
  if (!c || !b) return MDC_ERR_ARG;
  if (!b->handle) return MDC_OK;
  Arena* a = static_cast<Arena*>(b->handle);
  memset(b, 0, sizeof *b);
  release_arena(a);
  return MDC_OK;
}

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
