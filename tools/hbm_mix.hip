// HBM ceilings of this box for the traffic MIX and the ACCESS PATTERNS of the fused
// photometric + remap kernel (reads of u8 source windows, writes of f32 output tiles),
// stripped of all arithmetic.  Tells how far remap_tiled_kernel is from what the
// memory system can deliver for its pattern, and which pattern changes would pay.
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_mix.hip -o /tmp/hbm_mix && /tmp/hbm_mix
// Geometry = bench workload: 1024 frames, 1280x1024 u8 in, 640x480 f32 out.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int IW = 1280, IH = 1024, OW = 640, OH = 480;
constexpr long long NIN = (long long)IW * IH, NOUT = (long long)OW * OH;

enum { ST_PLAIN = 0, ST_NT = 1 };
template <int MODE, typename T>
__device__ __forceinline__ void st(T v, T* p) {
  if (MODE == ST_NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}
template <int MODE, typename T>
__device__ __forceinline__ T ld(const T* p) {
  if (MODE == ST_NT) return __builtin_nontemporal_load(p);
  return *p;
}

// ---- linear streams -------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void w_lin4(float* __restrict__ p, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, s = (size_t)gridDim.x * 256;
  for (; i < n; i += s) st<MODE>((float)i, p + i);
}
template <int MODE>
__global__ __launch_bounds__(256) void w_lin16(f32x4* __restrict__ p, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, s = (size_t)gridDim.x * 256;
  for (; i < n; i += s) { f32x4 v = {(float)i, 1.f, 2.f, 3.f}; st<MODE>(v, p + i); }
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void w_lin8(f32x2* __restrict__ p, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, s = (size_t)gridDim.x * 256;
  for (; i < n; i += s) { f32x2 v = {(float)i, 1.f}; st<MODE>(v, p + i); }
}
// block-contiguous: each workgroup owns a contiguous span (no grid stride) and walks it
template <int MODE>
__global__ __launch_bounds__(256) void w_span4(float* __restrict__ p, size_t n, size_t span) {
  size_t b = (size_t)blockIdx.x * span, e = std::min(n, b + span);
  for (size_t i = b + threadIdx.x; i < e; i += 256) st<MODE>((float)i, p + i);
}
template <int MODE>
__global__ __launch_bounds__(256) void r_lin16(const u32x4* __restrict__ p, uint32_t* out, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, s = (size_t)gridDim.x * 256;
  uint32_t acc = 0;
  for (; i < n; i += s) { u32x4 v = ld<MODE>(p + i); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[0] = acc;
}
// linear mix: per 16 B read, WPR dwords written per thread (wave-contiguous 256 B each)
template <int LMODE, int SMODE, int WPR>
__global__ __launch_bounds__(256) void mix_lin(const u32x4* __restrict__ a, float* __restrict__ b, size_t n16) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, s = (size_t)gridDim.x * 256;
  for (; i < n16; i += s) {
    u32x4 v = ld<LMODE>(a + i);
    const float x = __uint_as_float((v.x ^ v.y ^ v.z ^ v.w) & 0x3fffffffu);
    const size_t blk = (i / 256) * 256 * WPR + (i % 256);
#pragma unroll
    for (int k = 0; k < WPR; k++) st<SMODE>(x, b + blk + k * 256);
  }
}

// ---- tile patterns (the fused kernel's skeleton) ---------------------------------------
// Workgroup = TWxTH output tile, NT = TW*TH/4 threads, lane = output column, 4 rows/thread,
// loops over fpb frames.  READ: a WINW x WINH byte window of the source frame at the tile's
// nominal source position, 16-B chunks, one or more chunks per thread.  WRITE: the tile.
template <int TW, int TH, int LMODE, int SMODE, bool DO_READ, bool DO_WRITE>
__global__ __launch_bounds__(TW* TH / 4) void tile_rw(const uint8_t* __restrict__ in, float* __restrict__ out,
                                                       int nframes, int fpb, int winw, int winh, int band) {
  constexpr int NT = TW * TH / 4;
  constexpr int TX = OW / TW, TY = OH / TH, NTILES = TX * TY;
  __shared__ u32x4 sink[NT];
  const int ntp = gridDim.x;
  int tile = band ? (blockIdx.x & 7) * (ntp >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  if (tile >= NTILES) return;
  const int tx = tile % TX, ty = tile / TX;
  const int f0 = blockIdx.y * fpb, f1 = min(nframes, f0 + fpb);
  const int tid = threadIdx.x;
  // source window: scale 1.345 about the centre, x0 16-B aligned
  const int cx = (int)(640.f + (tx * TW + TW / 2 - 320) * 1.345f), cy = (int)(512.f + (ty * TH + TH / 2 - 240) * 1.52f);
  const int x0 = max(0, (cx - winw / 2)) & ~15, y0 = max(0, cy - winh / 2);
  const int cpr = winw / 16, nch = cpr * winh;
  const int lane_x = tid % TW, row0 = (tid / TW) * 4;
  const long long obase = (long long)(ty * TH + row0) * OW + tx * TW + lane_x;
  for (int f = f0; f < f1; f++) {
    uint32_t acc = 0;
    if (DO_READ) {
      const uint8_t* src = in + (long long)f * NIN;
      for (int c = tid; c < nch; c += NT) {
        const int r = c / cpr;
        u32x4 v = ld<LMODE>(reinterpret_cast<const u32x4*>(src + (long long)(y0 + r) * IW + x0 + (c - r * cpr) * 16));
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
        sink[tid] = v;
      }
      __syncthreads();
    }
    if (DO_WRITE) {
      float* dst = out + (long long)f * NOUT + obase;
      const float x = __uint_as_float(acc & 0x3fffffffu);
#pragma unroll
      for (int j = 0; j < 4; j++) st<SMODE>(x, dst + j * OW);
    } else if (acc == 0x12345678u) out[0] = 1.f;
  }
}

// Wave-private variant of the tile skeleton: the 8 waves of a 64x32 tile each own a 64x4 output strip and
// read their OWN window (128 B x 8 rows = one 16-byte load per lane), so nothing is shared between the
// waves of a workgroup and no barrier is needed.  PF = loads of frame f+1 issued before frame f is
// stored; BAR = keep a workgroup barrier per frame anyway (isolates what the barrier costs).
template <int PF, bool BAR, int SMODE>
__global__ __launch_bounds__(512) void strip_rw(const uint8_t* __restrict__ in, float* __restrict__ out, int nframes,
                                                int fpb, int band) {
  constexpr int TW = 64, TH = 32, TX = OW / TW, TY = OH / TH, NTILES = TX * TY;
  __shared__ u32x4 sink[512];
  const int ntp = gridDim.x;
  int tile = band ? (blockIdx.x & 7) * (ntp >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  if (tile >= NTILES) return;
  const int tx = tile % TX, ty = tile / TX;
  const int f0 = blockIdx.y * fpb, f1 = min(nframes, f0 + fpb);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cx = (int)(640.f + (tx * TW + TW / 2 - 320) * 1.345f);
  const int cy = (int)(512.f + (ty * TH + wave * 4 + 2 - 240) * 1.52f);
  const int x0 = max(0, (cx - 64)) & ~15, y0 = max(0, cy - 4);
  const long long soff = (long long)(y0 + (lane >> 3)) * IW + x0 + (lane & 7) * 16;
  const long long obase = (long long)(ty * TH + wave * 4) * OW + tx * TW + lane;
  u32x4 cur = *reinterpret_cast<const u32x4*>(in + (long long)f0 * NIN + soff);
  for (int f = f0; f < f1; f++) {
    u32x4 nxt = cur;
    if (PF) {
      if (f + 1 < f1) nxt = *reinterpret_cast<const u32x4*>(in + (long long)(f + 1) * NIN + soff);
    } else if (f > f0) {
      cur = *reinterpret_cast<const u32x4*>(in + (long long)f * NIN + soff);
    }
    const uint32_t acc = cur.x ^ cur.y ^ cur.z ^ cur.w;
    if (BAR) {
      sink[tid] = cur;
      __syncthreads();
    }
    float* dst = out + (long long)f * NOUT + obase;
    const float x = __uint_as_float(acc & 0x3fffffffu);
#pragma unroll
    for (int j = 0; j < 4; j++) st<SMODE>(x, dst + j * OW);
    if (PF) cur = nxt;
  }
}

// tile skeleton with the loads of frame f+1 issued before frame f is written (software prefetch), barrier kept
// il != 0: group g of G = gridDim.y takes frames g, g+G, g+2G, ... (all resident workgroups sweep the batch side
// by side) instead of fpb consecutive frames.  Dynamic LDS passed at launch is unused: it only lowers occupancy.
template <int TW, int TH, int SMODE, bool BAR>
__global__ __launch_bounds__(TW* TH / 4) void tile_rw_pf(const uint8_t* __restrict__ in, float* __restrict__ out,
                                                          int nframes, int fpb, int winw, int winh, int band, int il = 0) {
  constexpr int NT = TW * TH / 4;
  constexpr int TX = OW / TW, TY = OH / TH, NTILES = TX * TY;
  __shared__ u32x4 sink[NT];
  const int ntp = gridDim.x;
  int tile = band ? (blockIdx.x & 7) * (ntp >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  if (tile >= NTILES) return;
  const int tx = tile % TX, ty = tile / TX;
  const int fs = il ? (int)gridDim.y : 1;
  const int f0 = il ? (int)blockIdx.y : blockIdx.y * fpb, f1 = il ? nframes : min(nframes, f0 + fpb);
  const int tid = threadIdx.x;
  const int cx = (int)(640.f + (tx * TW + TW / 2 - 320) * 1.345f), cy = (int)(512.f + (ty * TH + TH / 2 - 240) * 1.52f);
  const int x0 = max(0, (cx - winw / 2)) & ~15, y0 = max(0, cy - winh / 2);
  const int cpr = winw / 16, nch = cpr * winh;
  const int lane_x = tid % TW, row0 = (tid / TW) * 4;
  const long long obase = (long long)(ty * TH + row0) * OW + tx * TW + lane_x;
  // at most 2 chunks per thread (nch <= 2 NT)
  long long so[2];
  bool has[2];
  for (int k = 0; k < 2; k++) {
    const int c = tid + k * NT;
    has[k] = c < nch;
    const int r = has[k] ? c / cpr : 0;
    so[k] = (long long)(y0 + r) * IW + x0 + (c - r * cpr) * 16;
  }
  u32x4 cur[2] = {}, nxt[2] = {};
  for (int k = 0; k < 2; k++)
    if (has[k]) cur[k] = *reinterpret_cast<const u32x4*>(in + (long long)f0 * NIN + so[k]);
  for (int f = f0; f < f1; f += fs) {
    if (f + fs < f1)
      for (int k = 0; k < 2; k++)
        if (has[k]) nxt[k] = *reinterpret_cast<const u32x4*>(in + (long long)(f + fs) * NIN + so[k]);
    uint32_t acc = 0;
    for (int k = 0; k < 2; k++) acc ^= cur[k].x ^ cur[k].y ^ cur[k].z ^ cur[k].w;
    if (BAR) {
      sink[tid] = cur[0];
      __syncthreads();
    }
    float* dst = out + (long long)f * NOUT + obase;
    const float x = __uint_as_float(acc & 0x3fffffffu);
#pragma unroll
    for (int j = 0; j < 4; j++) st<SMODE>(x, dst + j * OW);
    for (int k = 0; k < 2; k++) cur[k] = nxt[k];
  }
}

// The tile skeleton with a workgroup's fpb frames taken as chunks of `chunk` consecutive frames, the chunks
// of the G = gridDim.y groups interleaved: iteration i of group y works on frame ((i / chunk) * G + y) * chunk
// + i % chunk.  The groups resident at one time then cover a COMPACT range of frames (like a small fpb), while
// every workgroup still lives for fpb frames (one prologue).  chunk = fpb is the plain consecutive assignment.
template <int TW, int TH, int SMODE>
__global__ __launch_bounds__(TW* TH / 4) void tile_rw_chunked(const uint8_t* __restrict__ in, float* __restrict__ out,
                                                               int nframes, int fpb, int winw, int winh, int chunk) {
  constexpr int NT = TW * TH / 4;
  constexpr int TX = OW / TW, TY = OH / TH, NTILES = TX * TY;
  __shared__ u32x4 sink[NT];
  const int ntp = gridDim.x;
  int tile = (blockIdx.x & 7) * (ntp >> 3) + (blockIdx.x >> 3);
  if (tile >= NTILES) return;
  const int tx = tile % TX, ty = tile / TX;
  const int G = gridDim.y, y = blockIdx.y;
  const int tid = threadIdx.x;
  const int cx = (int)(640.f + (tx * TW + TW / 2 - 320) * 1.345f), cy = (int)(512.f + (ty * TH + TH / 2 - 240) * 1.52f);
  const int x0 = max(0, (cx - winw / 2)) & ~15, y0 = max(0, cy - winh / 2);
  const int cpr = winw / 16, nch = cpr * winh;
  const int lane_x = tid % TW, row0 = (tid / TW) * 4;
  const long long obase = (long long)(ty * TH + row0) * OW + tx * TW + lane_x;
  long long so[2];
  bool has[2];
  for (int k = 0; k < 2; k++) {
    const int c = tid + k * NT;
    has[k] = c < nch;
    const int r = has[k] ? c / cpr : 0;
    so[k] = (long long)(y0 + r) * IW + x0 + (c - r * cpr) * 16;
  }
  auto frame_of = [&](int i) { return ((i / chunk) * G + y) * chunk + i % chunk; };
  u32x4 cur[2] = {}, nxt[2] = {};
  for (int k = 0; k < 2; k++)
    if (has[k]) cur[k] = *reinterpret_cast<const u32x4*>(in + (long long)frame_of(0) * NIN + so[k]);
  for (int i = 0; i < fpb; i++) {
    const int f = frame_of(i);
    if (i + 1 < fpb)
      for (int k = 0; k < 2; k++)
        if (has[k]) nxt[k] = *reinterpret_cast<const u32x4*>(in + (long long)frame_of(i + 1) * NIN + so[k]);
    uint32_t acc = 0;
    for (int k = 0; k < 2; k++) acc ^= cur[k].x ^ cur[k].y ^ cur[k].z ^ cur[k].w;
    sink[tid] = cur[0];
    __syncthreads();
    float* dst = out + (long long)f * NOUT + obase;
    const float x = __uint_as_float(acc & 0x3fffffffu);
#pragma unroll
    for (int j = 0; j < 4; j++) st<SMODE>(x, dst + j * OW);
    for (int k = 0; k < 2; k++) cur[k] = nxt[k];
  }
}

// tile skeleton with explicit frame strides (bytes in, floats out): do the power-of-two-ish frame sizes
// (1280*1024 B, 640*480*4 B) alias the frames of one tile onto the same channels / banks?
template <int TW, int TH, int SMODE>
__global__ __launch_bounds__(TW* TH / 4) void tile_rw_strided(const uint8_t* __restrict__ in, float* __restrict__ out, int nframes,
                                                               int fpb, int winw, int winh, long long sin, long long sout) {
  constexpr int NT = TW * TH / 4;
  constexpr int TX = OW / TW, TY = OH / TH, NTILES = TX * TY;
  __shared__ u32x4 sink[NT];
  const int ntp = gridDim.x;
  int tile = (blockIdx.x & 7) * (ntp >> 3) + (blockIdx.x >> 3);
  if (tile >= NTILES) return;
  const int tx = tile % TX, ty = tile / TX;
  const int f0 = blockIdx.y * fpb, f1 = min(nframes, f0 + fpb);
  const int tid = threadIdx.x;
  const int cx = (int)(640.f + (tx * TW + TW / 2 - 320) * 1.345f), cy = (int)(512.f + (ty * TH + TH / 2 - 240) * 1.52f);
  const int x0 = max(0, (cx - winw / 2)) & ~15, y0 = max(0, cy - winh / 2);
  const int cpr = winw / 16, nch = cpr * winh;
  const int lane_x = tid % TW, row0 = (tid / TW) * 4;
  const long long obase = (long long)(ty * TH + row0) * OW + tx * TW + lane_x;
  long long so[2];
  bool has[2];
  for (int k = 0; k < 2; k++) {
    const int c = tid + k * NT;
    has[k] = c < nch;
    const int r = has[k] ? c / cpr : 0;
    so[k] = (long long)(y0 + r) * IW + x0 + (c - r * cpr) * 16;
  }
  u32x4 cur[2] = {}, nxt[2] = {};
  for (int k = 0; k < 2; k++)
    if (has[k]) cur[k] = *reinterpret_cast<const u32x4*>(in + (long long)f0 * sin + so[k]);
  for (int f = f0; f < f1; f++) {
    if (f + 1 < f1)
      for (int k = 0; k < 2; k++)
        if (has[k]) nxt[k] = *reinterpret_cast<const u32x4*>(in + (long long)(f + 1) * sin + so[k]);
    uint32_t acc = 0;
    for (int k = 0; k < 2; k++) acc ^= cur[k].x ^ cur[k].y ^ cur[k].z ^ cur[k].w;
    sink[tid] = cur[0];
    __syncthreads();
    float* dst = out + (long long)f * sout + obase;
    const float x = __uint_as_float(acc & 0x3fffffffu);
#pragma unroll
    for (int j = 0; j < 4; j++) st<SMODE>(x, dst + j * OW);
    for (int k = 0; k < 2; k++) cur[k] = nxt[k];
  }
}

// read-only "pieces": a workgroup reads 7168 bytes of each of its frames as rows of L contiguous bytes
// (stride = one image row); the 104 workgroups of a frame group tile a 1024 x 728 byte region without
// overlap.  Same bytes for every L -- isolates how the piece length affects the achieved read rate.
template <int L>
__global__ __launch_bounds__(512) void read_pieces(const uint8_t* __restrict__ in, float* __restrict__ out, int nframes,
                                                   int fpb) {
  constexpr int ROWS = 7168 / L, NCOL = 1024 / L, CPR = L / 16;
  __shared__ u32x4 sink[512];
  const int col = blockIdx.x % NCOL, rb = blockIdx.x / NCOL;
  const int tid = threadIdx.x;
  const int f0 = blockIdx.y * fpb, f1 = min(nframes, f0 + fpb);
  const bool active = tid < ROWS * CPR;
  const int r = tid / CPR, c = tid % CPR;
  const long long off = (long long)(146 + rb * ROWS + r) * IW + 128 + col * L + c * 16;
  uint32_t acc = 0;
  for (int f = f0; f < f1; f++) {
    if (active) {
      u32x4 v = *reinterpret_cast<const u32x4*>(in + (long long)f * NIN + off);
      acc ^= v.x ^ v.y ^ v.z ^ v.w;
      sink[tid] = v;
    }
    __syncthreads();
  }
  if (acc == 0x12345678u) out[0] = 1.f;
}

// full-row writer: a workgroup writes ROWS whole output rows of each of its frames (contiguous
// ROWS*2560 bytes), 4 B per lane, wave-contiguous
template <int SMODE, int ROWS>
__global__ __launch_bounds__(256) void w_rows(float* __restrict__ out, int nframes, int fpb) {
  const int band = blockIdx.x;  // OH/ROWS bands
  const int f0 = blockIdx.y * fpb, f1 = min(nframes, f0 + fpb);
  for (int f = f0; f < f1; f++) {
    float* dst = out + (long long)f * NOUT + (long long)band * ROWS * OW;
    for (int i = threadIdx.x; i < ROWS * OW; i += 256) st<SMODE>((float)i, dst + i);
  }
}

static hipEvent_t e0, e1;
template <typename F>
static float time_ms(F launch, int reps = 5) {
  launch();
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < reps; r++) {
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = std::min(best, ms);
  }
  return best;
}
static void report(const char* name, double bytes, float ms) {
  printf("%-44s %9.1f MB  %8.4f ms  %8.1f GB/s\n", name, bytes / 1e6, ms, bytes / ms / 1e6);
  fflush(stdout);
}

int main() {
  const int F = 1024;
  const size_t in_bytes = (size_t)F * NIN, out_bytes = (size_t)F * NOUT * 4;
  uint8_t* d_in;
  float* d_out;
  uint32_t* d_flag;
  hipMalloc(&d_in, in_bytes);
  hipMalloc(&d_out, out_bytes);
  hipMalloc(&d_flag, 4);
  hipMemset(d_in, 1, in_bytes);
  hipMemset(d_out, 0, out_bytes);
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const size_t n4 = out_bytes / 4, n16 = out_bytes / 16;
  for (int G : {2048, 8192}) {
    char nm[96];
    snprintf(nm, sizeof nm, "write lin 4B nt      grid %d", G);
    report(nm, out_bytes, time_ms([&] { w_lin4<ST_NT><<<G, 256>>>(d_out, n4); }));
    snprintf(nm, sizeof nm, "write lin 4B plain   grid %d", G);
    report(nm, out_bytes, time_ms([&] { w_lin4<ST_PLAIN><<<G, 256>>>(d_out, n4); }));
    snprintf(nm, sizeof nm, "write lin 8B nt      grid %d", G);
    report(nm, out_bytes, time_ms([&] { w_lin8<ST_NT><<<G, 256>>>((f32x2*)d_out, out_bytes / 8); }));
    snprintf(nm, sizeof nm, "write lin 8B plain   grid %d", G);
    report(nm, out_bytes, time_ms([&] { w_lin8<ST_PLAIN><<<G, 256>>>((f32x2*)d_out, out_bytes / 8); }));
    snprintf(nm, sizeof nm, "write lin 16B nt     grid %d", G);
    report(nm, out_bytes, time_ms([&] { w_lin16<ST_NT><<<G, 256>>>((f32x4*)d_out, n16); }));
    snprintf(nm, sizeof nm, "write lin 16B plain  grid %d", G);
    report(nm, out_bytes, time_ms([&] { w_lin16<ST_PLAIN><<<G, 256>>>((f32x4*)d_out, n16); }));
    snprintf(nm, sizeof nm, "write span 4B nt     grid %d", G);
    report(nm, out_bytes, time_ms([&] { w_span4<ST_NT><<<G, 256>>>(d_out, n4, (n4 + G - 1) / G); }));
    snprintf(nm, sizeof nm, "read lin 16B plain   grid %d", G);
    report(nm, in_bytes, time_ms([&] { r_lin16<ST_PLAIN><<<G, 256>>>((const u32x4*)d_in, d_flag, in_bytes / 16); }));
    snprintf(nm, sizeof nm, "read lin 16B nt      grid %d", G);
    report(nm, in_bytes, time_ms([&] { r_lin16<ST_NT><<<G, 256>>>((const u32x4*)d_in, d_flag, in_bytes / 16); }));
  }
  // linear mix at the fused kernel's ideal ratio: 16 B read : 32 B written (0.61 MB : 1.23 MB)
  {
    const size_t r16 = out_bytes / 32;  // chunks read so that 8 dwords per chunk fill the output
    const double bytes = (double)r16 * 16 + (double)out_bytes;
    report("mix lin 16B plain-ld : 8x4B nt-st   g4096", bytes, time_ms([&] { mix_lin<ST_PLAIN, ST_NT, 8><<<4096, 256>>>((const u32x4*)d_in, d_out, r16); }));
    report("mix lin 16B nt-ld    : 8x4B nt-st   g4096", bytes, time_ms([&] { mix_lin<ST_NT, ST_NT, 8><<<4096, 256>>>((const u32x4*)d_in, d_out, r16); }));
    report("mix lin 16B plain-ld : 8x4B plain   g4096", bytes, time_ms([&] { mix_lin<ST_PLAIN, ST_PLAIN, 8><<<4096, 256>>>((const u32x4*)d_in, d_out, r16); }));
    report("mix lin 16B plain-ld : 8x4B nt-st   g16384", bytes, time_ms([&] { mix_lin<ST_PLAIN, ST_NT, 8><<<16384, 256>>>((const u32x4*)d_in, d_out, r16); }));
  }
  // tile patterns: 64x32 tiles (150), band-mapped over XCDs like the real kernel, fpb 32
  {
    const int fpb = 32, groups = F / fpb;
    dim3 g64(152, groups), g128(80, groups);
    const double wb = (double)out_bytes;
    // window 128 B x 56 rows -> 150*7168 = 1.075 MB requested per frame
    const double rb = 150.0 * 128 * 56 * F;
    report("tile64x32 write-only nt", wb, time_ms([&] { tile_rw<64, 32, ST_PLAIN, ST_NT, false, true><<<g64, 512>>>(d_in, d_out, F, fpb, 128, 56, 1); }));
    report("tile64x32 write-only plain", wb, time_ms([&] { tile_rw<64, 32, ST_PLAIN, ST_PLAIN, false, true><<<g64, 512>>>(d_in, d_out, F, fpb, 128, 56, 1); }));
    report("tile64x32 write-only nt  (no band map)", wb, time_ms([&] { tile_rw<64, 32, ST_PLAIN, ST_NT, false, true><<<g64, 512>>>(d_in, d_out, F, fpb, 128, 56, 0); }));
    report("tile128x16 write-only nt", wb, time_ms([&] { tile_rw<128, 16, ST_PLAIN, ST_NT, false, true><<<dim3(152, groups), 512>>>(d_in, d_out, F, fpb, 192, 40, 1); }));
    report("tile64x32 read-only plain (requested B)", rb, time_ms([&] { tile_rw<64, 32, ST_PLAIN, ST_NT, true, false><<<g64, 512>>>(d_in, d_out, F, fpb, 128, 56, 1); }));
    report("tile64x32 read-only nt    (requested B)", rb, time_ms([&] { tile_rw<64, 32, ST_NT, ST_NT, true, false><<<g64, 512>>>(d_in, d_out, F, fpb, 128, 56, 1); }));
    report("tile64x32 read+write plain-ld nt-st", rb + wb, time_ms([&] { tile_rw<64, 32, ST_PLAIN, ST_NT, true, true><<<g64, 512>>>(d_in, d_out, F, fpb, 128, 56, 1); }));
    report("tile64x32 read+write nt-ld nt-st", rb + wb, time_ms([&] { tile_rw<64, 32, ST_NT, ST_NT, true, true><<<g64, 512>>>(d_in, d_out, F, fpb, 128, 56, 1); }));
    report("tile64x32 read+write plain-ld plain-st", rb + wb, time_ms([&] { tile_rw<64, 32, ST_PLAIN, ST_PLAIN, true, true><<<g64, 512>>>(d_in, d_out, F, fpb, 128, 56, 1); }));
    // structure experiments (round 2): what do the per-frame barrier and a prefetch cost / buy in the skeleton?
    const double rbs = 150.0 * 8 * 1024 * F;  // strips: 8 waves x 1 KiB per tile and frame
    report("tile64x32 r+w prefetch, barrier", rb + wb, time_ms([&] { tile_rw_pf<64, 32, ST_NT, true><<<g64, 512>>>(d_in, d_out, F, fpb, 128, 56, 1); }));
    report("tile64x32 r+w prefetch, NO barrier", rb + wb, time_ms([&] { tile_rw_pf<64, 32, ST_NT, false><<<g64, 512>>>(d_in, d_out, F, fpb, 128, 56, 1); }));
    report("tile128x16 r+w prefetch, barrier (192x40 win)", 150.0 * 192 * 40 * F + wb, time_ms([&] { tile_rw_pf<128, 16, ST_NT, true><<<dim3(152, groups), 512>>>(d_in, d_out, F, fpb, 192, 40, 1); }));
    report("tile128x32 r+w prefetch, barrier (192x56 win)", 75.0 * 192 * 56 * F + wb, time_ms([&] { tile_rw_pf<128, 32, ST_NT, true><<<dim3(80, groups), 1024>>>(d_in, d_out, F, fpb, 192, 56, 1); }));
    report("strips 64x4 per wave, no prefetch, no barrier", rbs + wb, time_ms([&] { strip_rw<0, false, ST_NT><<<g64, 512>>>(d_in, d_out, F, fpb, 1); }));
    report("strips 64x4 per wave, prefetch, no barrier", rbs + wb, time_ms([&] { strip_rw<1, false, ST_NT><<<g64, 512>>>(d_in, d_out, F, fpb, 1); }));
    report("strips 64x4 per wave, prefetch, barrier", rbs + wb, time_ms([&] { strip_rw<1, true, ST_NT><<<g64, 512>>>(d_in, d_out, F, fpb, 1); }));
    // frame strides: dense (as the API has them) vs padded by odd multiples of 128 B / a few KB
    {
      const int Fs = 960;  // fewer frames so that padded strides still fit the buffers
      for (long long pin : {0LL, 128LL, 1152LL, 4096LL + 128, 65536LL + 128})
        for (long long pout : {0LL, 32LL, 288LL, 1024LL + 32}) {
          if ((pin == 0) != (pout == 0) && !(pin == 0 || pout == 0)) continue;
          char nm[96];
          snprintf(nm, sizeof nm, "tile64x32 pf strided: in +%lld B, out +%lld floats", pin, pout);
          report(nm, (rb + wb) * Fs / F, time_ms([&] { tile_rw_strided<64, 32, ST_NT><<<dim3(152, Fs / 32), 512>>>(d_in, d_out, Fs, 32, 128, 56, NIN + pin, NOUT + pout); }));
        }
    }
    // compact in-flight frame set without a short workgroup life: chunks of c frames, groups interleaved
    for (int lds : {0, 52 * 1024})
      for (int fp : {32, 64, 128})
        for (int ck : {1, 4, 8, 16, 32}) {
          if (ck > fp) continue;
          char nm[96];
          snprintf(nm, sizeof nm, "tile64x32 chunked fpb %d chunk %d %s", fp, ck, lds ? "3WG/CU" : "4WG/CU");
          report(nm, rb + wb, time_ms([&] { tile_rw_chunked<64, 32, ST_NT><<<dim3(152, F / fp), 512, lds>>>(d_in, d_out, F, fp, 128, 56, ck); }));
        }
    // occupancy: the real kernel runs 3 workgroups of 512 per CU (LDS), this skeleton 4
    report("tile64x32 pf barrier, 3 WG/CU (52 KiB dummy LDS)", rb + wb, time_ms([&] { tile_rw_pf<64, 32, ST_NT, true><<<g64, 512, 52 * 1024>>>(d_in, d_out, F, fpb, 128, 56, 1); }));
    report("tile64x32 pf barrier, 2 WG/CU (70 KiB dummy LDS)", rb + wb, time_ms([&] { tile_rw_pf<64, 32, ST_NT, true><<<g64, 512, 64 * 1024>>>(d_in, d_out, F, fpb, 128, 56, 1); }));
    // one round of persistent workgroups sweeping the batch side by side (G groups, frames g, g+G, ...)
    for (int G : {5, 4, 10, 32}) {
      char nm[96];
      snprintf(nm, sizeof nm, "tile64x32 pf barrier, interleaved G=%d", G);
      report(nm, rb + wb, time_ms([&] { tile_rw_pf<64, 32, ST_NT, true><<<dim3(152, G), 512>>>(d_in, d_out, F, 0, 128, 56, 1, 1); }));
      snprintf(nm, sizeof nm, "tile64x32 pf barrier, interleaved G=%d 3WG/CU", G);
      report(nm, rb + wb, time_ms([&] { tile_rw_pf<64, 32, ST_NT, true><<<dim3(152, G), 512, 52 * 1024>>>(d_in, d_out, F, 0, 128, 56, 1, 1); }));
    }
    for (int fp : {8, 64}) {
      char nm[96];
      snprintf(nm, sizeof nm, "strips prefetch no barrier fpb %d", fp);
      report(nm, rbs + wb, time_ms([&] { strip_rw<1, false, ST_NT><<<dim3(152, (F + fp - 1) / fp), 512>>>(d_in, d_out, F, fp, 1); }));
      snprintf(nm, sizeof nm, "tile64x32 prefetch barrier fpb %d", fp);
      report(nm, rb + wb, time_ms([&] { tile_rw_pf<64, 32, ST_NT, true><<<dim3(152, (F + fp - 1) / fp), 512>>>(d_in, d_out, F, fp, 128, 56, 1); }));
    }
    for (int fp : {8, 16, 64, 205}) {
      char nm[96];
      snprintf(nm, sizeof nm, "tile64x32 read+write plain-ld nt-st fpb %d", fp);
      report(nm, rb + wb, time_ms([&] { tile_rw<64, 32, ST_PLAIN, ST_NT, true, true><<<dim3(152, (F + fp - 1) / fp), 512>>>(d_in, d_out, F, fp, 128, 56, 1); }));
    }
    {
      const double pb = 104.0 * 7168 * F;
      dim3 gp(104, groups);
      report("read pieces L=128  (56 rows)", pb, time_ms([&] { read_pieces<128><<<gp, 512>>>(d_in, d_out, F, fpb); }));
      report("read pieces L=256  (28 rows)", pb, time_ms([&] { read_pieces<256><<<gp, 512>>>(d_in, d_out, F, fpb); }));
      report("read pieces L=512  (14 rows)", pb, time_ms([&] { read_pieces<512><<<gp, 512>>>(d_in, d_out, F, fpb); }));
      report("read pieces L=1024 (7 rows)", pb, time_ms([&] { read_pieces<1024><<<gp, 512>>>(d_in, d_out, F, fpb); }));
      dim3 gp8(104, F / 8);
      report("read pieces L=128  fpb 8", pb, time_ms([&] { read_pieces<128><<<gp8, 512>>>(d_in, d_out, F, 8); }));
      report("read pieces L=1024 fpb 8", pb, time_ms([&] { read_pieces<1024><<<gp8, 512>>>(d_in, d_out, F, 8); }));
    }
    report("rows8 write-only nt (20 KB contiguous)", wb, time_ms([&] { w_rows<ST_NT, 8><<<dim3(OH / 8, groups), 256>>>(d_out, F, fpb); }));
    report("rows32 write-only nt (80 KB contiguous)", wb, time_ms([&] { w_rows<ST_NT, 32><<<dim3(OH / 32, F / 4), 256>>>(d_out, F, 4); }));
  }
  return 0;
}
