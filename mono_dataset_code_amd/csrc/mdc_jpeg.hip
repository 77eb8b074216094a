// JPEG ingest, device half (SURVEY.md section 8 row f2 "later GPU JPEG"): the host keeps the serial part of JPEG
// decoding -- Huffman decoding of the entropy-coded segments, csrc/host/image_codecs*.cpp -- and ships the quantised luma
// coefficients (64 int16 per 8x8 block, natural order) + the quantisation table; this kernel dequantises, runs libjpeg's
// "islow" integer inverse DCT (jidctint.c: the same constants, the same two passes, the same rounding as the host decoder,
// which is itself pinned bit for bit to libjpeg-turbo in tests/test_reader_cpu.py) and writes the 8-bit frame straight
// into the buffer the fused photometric + remap kernel reads.  Integer arithmetic only: bytes identical to the host path.
//
// Eight threads per block (a 1280 x 1024 frame has 20480 blocks): a block's 128 bytes are one contiguous load of its group, the
// two 1-D passes run on a column / a row per thread with the block turned through LDS in between, every thread stores the eight
// samples of its row.  In the reader the stage is bound by the PCIe transfer of the coefficients (2 bytes per pixel), which is what
// it trades for the host's IDCT time; with the coefficients made on the device (Huffman kernels below) it is bound by HBM.
#include "../../include/mdc_hip.h"
#include "mdc_internal.h"

#include <algorithm>

namespace mdc {
namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

template <class T>
__device__ __forceinline__ int descale(T x, int n) { return (int)((x + ((T)1 << (n - 1))) >> n); }
__device__ __forceinline__ unsigned clamp_sample(int x) {
  x += 128;
  return (unsigned)(x < 0 ? 0 : (x > 255 ? 255 : x));
}

// One 1-D pass of jidctint.c on (c0..c7) -> the eight outputs, descaled by `shift`; even part / odd part as the reference names
// them.  T = long long is the host decoder's arithmetic (libjpeg-turbo's JLONG on LP64).  T = int gives the same values whenever no
// intermediate leaves 32 bits: every intermediate is a fixed integer combination of the inputs, the largest sum of absolute weights
// is 61214 (the output o2), so inputs of magnitude <= kIdctSafe32 cannot overflow (61214 * 35079 + 2^17 < 2^31) -- which is every
// coefficient and workspace value of a real image; a thread whose inputs are larger takes the 64-bit form.
constexpr int kIdctSafe32 = 35079;
template <class T>
__device__ __forceinline__ void idct_1d(const int* c, int shift, int* o) {
  const T F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299, F1_847 = 15137,
          F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
  T z2 = c[2], z3 = c[6];
  T z1 = (z2 + z3) * F0_541;
  T tmp2 = z1 + z3 * (-F1_847), tmp3 = z1 + z2 * F0_765;
  T tmp0 = ((T)c[0] + (T)c[4]) * ((T)1 << 13), tmp1 = ((T)c[0] - (T)c[4]) * ((T)1 << 13);
  const T tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  tmp0 = c[7];
  tmp1 = c[5];
  tmp2 = c[3];
  tmp3 = c[1];
  z1 = tmp0 + tmp3;
  z2 = tmp1 + tmp2;
  z3 = tmp0 + tmp2;
  T z4 = tmp1 + tmp3;
  const T z5 = (z3 + z4) * F1_175;
  tmp0 *= F0_298;
  tmp1 *= F2_053;
  tmp2 *= F3_072;
  tmp3 *= F1_501;
  z1 *= -F0_899;
  z2 *= -F2_562;
  z3 *= -F1_961;
  z4 *= -F0_390;
  z3 += z5;
  z4 += z5;
  tmp0 = tmp0 + z1 + z3;
  tmp1 = tmp1 + z2 + z4;
  tmp2 = tmp2 + z2 + z3;
  tmp3 = tmp3 + z1 + z4;
  o[0] = descale<T>(tmp10 + tmp3, shift);
  o[7] = descale<T>(tmp10 - tmp3, shift);
  o[1] = descale<T>(tmp11 + tmp2, shift);
  o[6] = descale<T>(tmp11 - tmp2, shift);
  o[2] = descale<T>(tmp12 + tmp1, shift);
  o[5] = descale<T>(tmp12 - tmp1, shift);
  o[3] = descale<T>(tmp13 + tmp0, shift);
  o[4] = descale<T>(tmp13 - tmp0, shift);
}
__device__ __forceinline__ void idct_pass(const int* c, int shift, int* o) {
  unsigned big = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) big |= (unsigned)(c[k] + kIdctSafe32) > 2u * kIdctSafe32 ? 1u : 0u;
  if (big) idct_1d<long long>(c, shift, o);
  else idct_1d<int>(c, shift, o);
}

// records: per frame [64 x u16 quantisation table, natural order][blocks_rows x blocks_w x 64 int16], rec_i16 int16 apart
//
// EIGHT threads per block (one thread per block held 64 coefficients + 64 workspace values + 64-bit temporaries: 280 registers, one
// wave per SIMD, 2 TB/s): thread j of a block's group loads row j (16 bytes: a block is 128 contiguous bytes for its eight threads),
// dequantises, and the group turns the block through LDS -- rows in, columns out; pass 1 on a column per thread; columns in, rows out;
// pass 2 on a row per thread, which packs its eight samples into one 8-byte store.  A wave holds 8 blocks of a block row; nothing
// crosses waves, the barriers only order the LDS traffic.  Same integer arithmetic, term by term.
constexpr int kIdctThreads = 256, kIdctBlocks = kIdctThreads / 8, kIdctPitch = 9, kIdctTile = 8 * kIdctPitch;  // (rows 9 words apart, blocks 72: bank = 8 block + 9 row + column -- every access pattern below is conflict-free per half-wave)
__global__ __launch_bounds__(kIdctThreads) void jpeg_idct_kernel(const int16_t* __restrict__ records, long long rec_i16, uint8_t* __restrict__ frames,
                                                                 int W, int H, int blocks_w, int bw_used, int bh_used, long long nframes) {
  __shared__ int s_tile[kIdctBlocks * kIdctTile];
  const int j = threadIdx.x & 7, slot = threadIdx.x >> 3;
  const long long per_frame = (long long)bw_used * bh_used;
  const long long g = (long long)blockIdx.x * kIdctBlocks + slot;  // the block, over all frames
  const bool live = g < per_frame * nframes;
  const long long gg = live ? g : 0;
  const long long f = gg / per_frame;
  const int r = (int)(gg - f * per_frame);
  const int by = r / bw_used, bx = r - by * bw_used;
  const int16_t* rec = records + f * rec_i16;
  const int16_t* blk = rec + 64 + ((long long)by * blocks_w + bx) * 64;
  int* tile = s_tile + slot * kIdctTile;
  {  // row j of the block, dequantised (int * int as on the host) -> tile[j][0..7]
    const i32x4 v = *reinterpret_cast<const i32x4*>(blk + 8 * j);
    const i32x4 q = *reinterpret_cast<const i32x4*>(rec + 8 * j);  // (the table's u16 entries, two per word)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      tile[j * kIdctPitch + 2 * k] = (int)(short)(v[k] & 0xffff) * (int)((unsigned)q[k] & 0xffffu);
      tile[j * kIdctPitch + 2 * k + 1] = (int)(short)((unsigned)v[k] >> 16) * (int)((unsigned)q[k] >> 16);
    }
  }
  __syncthreads();
  int c[8];
#pragma unroll
  for (int k = 0; k < 8; k++) c[k] = tile[k * kIdctPitch + j];  // column j
  __syncthreads();
  int o[8];
  idct_pass(c, 11, o);  // pass 1: the column -> workspace column (scaled by 2^PASS1_BITS)
#pragma unroll
  for (int k = 0; k < 8; k++) tile[k * kIdctPitch + j] = o[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; k++) c[k] = tile[j * kIdctPitch + k];  // workspace row j
  idct_pass(c, 18, o);  // pass 2: the row -> samples
#pragma unroll
  for (int k = 0; k < 8; k++) o[k] = (int)clamp_sample(o[k]);
  if (!live || by * 8 + j >= H) return;
  uint8_t* line = frames + f * (long long)W * H + (long long)(by * 8 + j) * W + bx * 8;
  const bool whole = bx * 8 + 8 <= W && (W & 7) == 0 && ((reinterpret_cast<uintptr_t>(frames) | (uintptr_t)((long long)W * H)) & 7) == 0;
  if (whole) {
    const unsigned lo = (unsigned)(o[0] | o[1] << 8 | o[2] << 16 | o[3] << 24), hi = (unsigned)(o[4] | o[5] << 8 | o[6] << 16 | o[7] << 24);
    *reinterpret_cast<uint2*>(line) = make_uint2(lo, hi);
  } else {
    for (int x = 0; x < 8 && bx * 8 + x < W; x++) line[x] = (uint8_t)o[x];
  }
}

// ---------------------------------------------------------------------------------------------------------
// Huffman decoding on the device (the serial half of JPEG decoding, made parallel).  Input: a STREAM per frame
// (mdc_jpeg_stream_header, include/mdc_hip.h): the host has parsed the file's markers, built the decode tables of the scan
// and copied the entropy-coded segment with its FF00 byte stuffing (and its restart markers) removed -- 0.1 ms per frame,
// against 2.5 ms for the Huffman decoding itself.  Output: the LUMA coefficient record the inverse-DCT kernel above reads
// (what cv::imread(..., GRAYSCALE) keeps of a colour file: Y).  Three kernels, one per kind of stream, each leaves the
// frames of the other kinds alone; launches of up to 128 frames spread a frame over several workgroups (further down):
//
//  * jpeg_huffman_kernel<false>: one component, no restart markers (what the TUM mono dataset ships).  One workgroup of
//    1024 threads per frame.  The bit stream is cut into 1024 subsequences of S bits; thread i decodes [i S, (i+1) S) from an
//    entry state (bit position of a symbol boundary, coefficient index z inside the current block; z == 0: a DC symbol comes
//    next).  The true entry state of subsequence i is the exit state of subsequence i-1 -- unknown at first, so every thread
//    starts from a guess (its left neighbour decodes the last quarter of its own subsequence from an arbitrary state), and the
//    states are RELAXED: decode, hand the exit state to the right neighbour, decode again where the entry state changed, until
//    nothing changes.  Subsequence 0's entry state is exact, so after k rounds the first k are exact: the fixed point is the
//    sequential decoder's state sequence; and because Huffman streams resynchronise after a few symbols (bit position) and
//    every end-of-block resets z, wrong guesses heal inside one subsequence: 2-3 rounds in practice (the bound, 1024, only
//    costs time).  Then prefix sums over the blocks each subsequence completed and over the DC differences it met give every
//    thread its first block and its DC predictor, and a last pass decodes once more and WRITES the coefficients, DC terms as
//    final values: a block a thread begins and finishes leaves LDS as its whole 128-byte line, a block two subsequences share
//    is cleared by the thread that begins it and filled coefficient by coefficient (no zero-fill of the record).
//  * jpeg_huffman_kernel<true>: three components interleaved in one scan (YCbCr baseline), no restart markers.  The same
//    relaxation over a larger state: (bit, z, u), u = the block's position inside its MCU (hY x vY luma blocks, then Cb, then
//    Cr), which picks the table pair; chroma symbols are decoded and dropped, luma blocks are counted and written (into a
//    zero-filled record).  4:2:0 files need 6-8 rounds: a wrong u only heals when it happens to become right.
//  * jpeg_huffman_intervals_kernel<C>: files with restart markers, one or three components.  Every restart interval starts at a
//    byte the host has recorded, with z = 0 and all predictors 0: exact entry states -- a thread decodes whole short intervals
//    front to back and writes final DC values directly; long intervals are cut into parts that relax inside a wave.
//
// Integer logic only: the record equals the host decoder's bit for bit; a code no table holds, too few blocks or an interval
// that runs into the next one set the frame's status (the caller falls back to the host decoder).
// ---------------------------------------------------------------------------------------------------------
__device__ __constant__ unsigned char c_zigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                                      41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                                      30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// the decode tables as the kernels use them (copied from the stream): [0] DC luma, [1] AC luma, ([2] DC chroma, [3] AC chroma)
template <int NT>
struct HuffLds {
  uint32_t t1[NT][2048];
  uint32_t t2[NT][MDC_JPEG_HUFF_SUBTABLES][32];
};

// Geometry of a scan: luma blocks of an MCU (hY x vY), blocks per MCU (luma + 2 chroma for three components), MCUs per row,
// luma blocks in all, record pitch in blocks.
struct ScanGeo {
  int hY, vY, hv, nb, mx, nluma, pitch;
};
__device__ __forceinline__ ScanGeo scan_geo(const mdc_jpeg_stream_header* hd, int W, int H, int pitch) {
  ScanGeo g;
  const int ncomp = (int)(hd->comp_info & 255u);
  g.hY = ncomp == 3 ? (int)((hd->comp_info >> 8) & 15u) : 1;
  g.vY = ncomp == 3 ? (int)((hd->comp_info >> 12) & 15u) : 1;
  g.hv = g.hY * g.vY;
  g.nb = ncomp == 3 ? g.hv + 2 : 1;
  g.mx = (W + 8 * g.hY - 1) / (8 * g.hY);
  const int my = (H + 8 * g.vY - 1) / (8 * g.vY);
  g.nluma = g.mx * my * g.hv;
  g.pitch = pitch;
  return g;
}
// the q-th luma block of the scan (MCU by MCU, inside an MCU row by row) -> its place in the record
__device__ __forceinline__ int16_t* luma_block(int16_t* coef, int q, const ScanGeo& g) {
  const int m = q / g.hv, uu = q - m * g.hv;
  const int mcu_y = m / g.mx, mcu_x = m - mcu_y * g.mx;
  const int bx = mcu_x * g.hY + uu % g.hY, by = mcu_y * g.vY + uu / g.hY;
  return coef + ((long long)by * g.pitch + bx) * 64;
}
// which kernel a stream is for: 0 = one component, relaxed; 1 = three components, relaxed; 2 = restart intervals; -1 = none (bad header)
__device__ __forceinline__ int stream_kind(const mdc_jpeg_stream_header* hd, int W, int H, int pitch, int rows, long long stream_stride) {
  const uint32_t ecs_bytes = hd->ecs_bytes, ncomp = hd->comp_info & 255u, hY = (hd->comp_info >> 8) & 15u, vY = (hd->comp_info >> 12) & 15u;
  const uint32_t base = (uint32_t)sizeof(mdc_jpeg_stream_header) + (ncomp == 3 ? 2u * (uint32_t)sizeof(mdc_jpeg_huff) : 0u);
  const bool ok = hd->magic == MDC_JPEG_STREAM_MAGIC && (int)hd->w == W && (int)hd->h == H && ecs_bytes > 0 && ecs_bytes < (1u << 28) &&
                  (ncomp == 1 || (ncomp == 3 && hY >= 1 && hY <= 4 && vY >= 1 && vY <= 4)) && hd->n_intervals >= 1 && hd->n_intervals < (1u << 24) &&
                  hd->ecs_offset % 16 == 0 && hd->ecs_offset >= base + ((hd->restart_interval ? hd->n_intervals * 4u : 0u)) &&
                  (long long)hd->ecs_offset + ecs_bytes + 16 <= stream_stride && (hd->restart_interval != 0 || hd->n_intervals == 1);
  if (!ok) return -1;
  // the luma grid of the MCU-padded scan must fit the record
  const int h = ncomp == 3 ? (int)hY : 1, v = ncomp == 3 ? (int)vY : 1;
  if ((W + 8 * h - 1) / (8 * h) * h > pitch || (H + 8 * v - 1) / (8 * v) * v > rows) return -1;
  if (hd->restart_interval) return 2;
  return ncomp == 3 ? 1 : 0;
}

struct BitReader {
  const uint32_t* base;  // entropy-coded bytes, 4-byte aligned, zero-padded past the end
  uint64_t acc;          // next bits, MSB first
  int cnt;               // valid bits in acc
  uint32_t widx;         // next word to fetch
  uint32_t next;         // that word as loaded, bytes not yet swapped: the swap at the point of USE lets the load stay in flight until
                         // the next refill (~5 symbols) -- swapped at once, every refill waited for its own load
  uint32_t last;         // index of the last word that may be read
  // Where the words are read: `base` itself, or -- the split kernel: a segment's words [lo, ...) copied to LDS once with coalesced
  // loads (every refill is otherwise a 64-way divergent 4-byte load over a working set beyond the L1) -- that copy, as a generic
  // pointer.  One or the other for the whole workgroup: no choice per access.
  const uint32_t* src = nullptr;
  uint32_t lo = 0;
#ifdef MDC_EXP_HUFF_FAKE_STREAM  // diagnosis (wrong results): every refill reads one of 64 words -- what do the stream loads cost?
  __device__ __forceinline__ uint32_t raw(uint32_t i) const { return src[(max(min(i, last), lo) - lo) & 63u]; }
#else
  __device__ __forceinline__ uint32_t raw(uint32_t i) const { return src[max(min(i, last), lo) - lo]; }  // (clamped both ways: an entry state may come from anywhere)
#endif
  __device__ __forceinline__ void start(uint32_t bit) {
    widx = bit >> 5;
    const uint64_t w0 = __builtin_bswap32(raw(widx)), w1 = __builtin_bswap32(raw(widx + 1));
    const int sh = (int)(bit & 31);
    acc = (w0 << 32 | w1) << sh;
    cnt = 64 - sh;
    widx += 2;
    next = raw(widx);
  }
  __device__ __forceinline__ void refill() {  // afterwards cnt > 32
    if (cnt <= 32) {
      acc |= (uint64_t)__builtin_bswap32(next) << (32 - cnt);
      cnt += 32;
      widx++;
      next = raw(widx);
    }
  }
  __device__ __forceinline__ uint32_t pos() const { return widx * 32u - (uint32_t)cnt; }
  __device__ __forceinline__ void skip(int n) {
    acc <<= n;
    cnt -= n;
  }
};

// One symbol at the reader's position with table pair `tab` (0 luma, 2 chroma), DC when z == 0 else AC: -> coefficient index
// the symbol writes (0: the DC difference; -1: none) and its value; z advanced (>= 64: the block is complete); *wrong set for a
// code no table holds, a DC category above 11 (the host decoder refuses those too) or a run past the block.
// One code path for every symbol, selects instead of branches: the table is picked by address, one lookup of the next 11 bits gives
// length, run, size and -- mostly -- the value; lanes of a wave are at different symbols anyway, so every branch that some lane takes
// costs all of them (the branchy form of this function was ~130 instructions per symbol).  Every symbol consumes at least one bit.
template <int NT>
__device__ __forceinline__ int huff_symbol(BitReader& b, const HuffLds<NT>& T, int tab, int& z, int* value, bool* wrong_out) {
  const bool ac = z != 0;
  const uint32_t hi = (uint32_t)(b.acc >> 32);  // the next 32 bits (the reader keeps more than 32 valid): all a symbol can need
  uint32_t e = T.t1[tab + (ac ? 1 : 0)][hi >> 21];
  if ((e & 31u) == 31u) e = T.t2[tab + (ac ? 1 : 0)][(e >> 16) & (MDC_JPEG_HUFF_SUBTABLES - 1)][(hi >> 16) & 31u];  // bits 11..15 of the window
  const int len = (int)(e & 31u), run = (int)((e >> 5) & 15u), size = (int)((e >> 9) & 15u);
  const bool wrong = (unsigned)(len - 1) > 15u || (!ac && size > 11);
  // the value: in the entry (short code + short value), or the `size` bits behind the code, extended (len + size <= 31)
  const uint32_t behind = hi << len;
  const int rawv = (int)((behind >> 1) >> (31 - size));       // size 0 -> 0
  const int half = (1 << size) >> 1;
  const int ext = rawv < half ? rawv - (1 << size) + 1 : rawv;  // size 0 -> 0
  int v = (e & (1u << 13)) ? (int)(int16_t)(e >> 16) : ext;
  v = wrong ? 0 : v;
  b.skip(wrong ? 1 : len + size);  // (speculative rounds run through garbage: keep moving)
  // z: DC -> 1; a wrong AC code leaves it; EOB -> 64, ZRL -> z + 16; a coefficient at z + run -> z + run + 1
  const int zr = z + run;
  const bool coef = ac && !wrong && size != 0;
  const bool over = coef && zr > 63;
  const int at = !ac ? 0 : (coef && !over) ? zr : -1;
  const int z_nocoef = run == 15 ? z + 16 : 64;
  int zn = size != 0 ? zr + 1 : z_nocoef;
  zn = wrong ? z : zn;
  z = ac ? zn : 1;
  *value = v;
  *wrong_out = wrong || over;
  return at;
}

// Decodes from the state (bit, z, u) until a symbol boundary at or past `end`.  WRITE: coefficients of luma blocks [q, nluma)
// go to the record.  Returns the exit state; *nblk += luma blocks completed; *bad set when a code is in no table.
// *dc: the luma DC predictor.  A counting pass adds the DC differences it meets (the sum over a subsequence; an exclusive prefix sum
// over the subsequences is then every thread's predictor at its entry -- DC symbols come in block order); the write pass starts from
// that and writes final DC values, as the sequential decoder does (no pass over the record afterwards).
// Block staging (WRITE, `stage` != nullptr): the coefficients of a block this thread starts AND finishes are collected in LDS --
// dword p of the block at stage[p * kHuffThreads] (the pointer is already offset by the thread: a wave's lanes sit on 64 different
// banks) -- and leave as eight 16-byte rows when the block is complete instead of ~12 scattered 2-byte stores (the L2's partial-write
// rate bounded the whole decoder at ~110 k frames/s; profiles/r04_experiments/05_*, 08_*).  A block that straddles two subsequences
// is written coefficient by coefficient by both sides -- their coefficient sets are disjoint -- into a block the thread that BEGINS
// it has cleared before anybody's write pass (clear_shared_block): with that, the record needs no zero-fill.
constexpr int kHuffThreads = 1024;
// WHOLE: ALL eight rows -- the block's 128-byte line is written whole (the L2 merges the eight stores; a line written in part is
// read and modified in memory: non-zero rows only were 5 % slower at 256 frames per launch although 2.3 x fewer stores), and nobody has
// to clear it beforehand.  Otherwise the non-zero rows only, into a record that was cleared (small launches: 7 % faster at 64 frames).
template <bool WHOLE>
__device__ __forceinline__ void stage_flush_rows(uint32_t* stage, int16_t* cur, unsigned long long cmask) {
#pragma unroll 1
  for (int r = 0; r < 8; r++) {
    if (!WHOLE && !((cmask >> (8 * r)) & 0xffull)) continue;
    i32x4 v;
    v.x = (int)stage[(4 * r + 0) * kHuffThreads];
    v.y = (int)stage[(4 * r + 1) * kHuffThreads];
    v.z = (int)stage[(4 * r + 2) * kHuffThreads];
    v.w = (int)stage[(4 * r + 3) * kHuffThreads];
    *reinterpret_cast<i32x4*>(cur + 8 * r) = v;
    stage[(4 * r + 0) * kHuffThreads] = 0u;
    stage[(4 * r + 1) * kHuffThreads] = 0u;
    stage[(4 * r + 2) * kHuffThreads] = 0u;
    stage[(4 * r + 3) * kHuffThreads] = 0u;
  }
}
__device__ __forceinline__ void stage_flush_scattered(uint32_t* stage, int16_t* cur, unsigned long long cmask) {
  while (cmask) {  // an unfinished block at the end of the subsequence: its other coefficients are the next thread's
    const int n = __ffsll((long long)cmask) - 1;
    cmask &= cmask - 1;
    int16_t* h = reinterpret_cast<int16_t*>(stage + (n >> 1) * kHuffThreads) + (n & 1);
    cur[n] = *h;
    *h = 0;
  }
}

template <bool WRITE, bool COLOR, bool WHOLE = false, int NT = 2>
__device__ __forceinline__ void huff_run(BitReader& b, const HuffLds<NT>& T, uint32_t bit, int z, int u, uint32_t end, uint32_t* out_bit, int* out_z,
                                         int* out_u, int* nblk, int* bad, int* dc, int16_t* coef, int q, const ScanGeo& g, uint32_t* stage = nullptr) {
  b.start(bit);
  int done = 0;
  uint32_t p = bit;
  int16_t* cur = nullptr;
  if (WRITE && (!COLOR || u < g.hv) && q < g.nluma) cur = luma_block(coef, q, g);
  bool staged = WRITE && stage && z == 0;  // (z != 0: the block was begun by the left neighbour)
  unsigned long long cmask = 0;
  while (p < end) {
    b.refill();
    // Luma or chroma is read off u at every symbol.  (Carried from block to block as a per-lane flag it came out wrong in some builds
    // of the three-component split kernel -- DC terms of colour frames off from some block on, depending on unrelated lines of this
    // function; profiles/r04_experiments/08_*.  tests/test_reader.py runs the colour cases on the product and the fault-injection build.)
    const bool luma = !COLOR || u < g.hv;
    int v;
    bool wrong;
    const int at = huff_symbol(b, T, luma ? 0 : 2, z, &v, &wrong);
    // (past the last luma block: the padding bits, not an error; chroma symbols count like luma ones)
    if (wrong && (!WRITE || cur || (COLOR && !luma && q < g.nluma))) *bad = 1;
    if (at == 0 && luma) {
      *dc += v;
      v = *dc;
    }
#ifdef MDC_EXP_HUFF_NOSTORE  // diagnosis (wrong results): the write pass decodes but stores only DC terms
    if (WRITE && cur && at == 0) cur[0] = (int16_t)v;
#else
    if (WRITE && cur && at >= 0) {
      const int n = c_zigzag[at];
      if (staged) {
        reinterpret_cast<int16_t*>(stage + (n >> 1) * kHuffThreads)[n & 1] = (int16_t)v;
        cmask |= 1ull << n;
      } else {
        cur[n] = (int16_t)v;
      }
    }
#endif
    if (z >= 64) {
      if (WRITE && staged && cur && cmask) stage_flush_rows<WHOLE>(stage, cur, cmask);
      staged = WRITE && stage;
      cmask = 0;
      z = 0;
      if (luma) {
        done++;
        if (WRITE) q++;
      }
      if (COLOR) u = u + 1 == g.nb ? 0 : u + 1;
      if (WRITE) cur = ((!COLOR || u < g.hv) && q < g.nluma) ? luma_block(coef, q, g) : nullptr;
    }
    p = b.pos();  // (every symbol consumes at least one bit: the loop ends whatever the tables hold)
  }
  if (WRITE && staged && cur && cmask) stage_flush_scattered(stage, cur, cmask);
  *out_bit = p;
  *out_z = z;
  *out_u = u;
  *nblk += done;
}

// tables -> LDS (NT of them: luma pair in the header, chroma pair right behind it)
template <int NT, int THREADS>
__device__ __forceinline__ void load_tables(HuffLds<NT>& T, const mdc_jpeg_stream_header* hd, int tid) {
  const mdc_jpeg_huff* chroma = reinterpret_cast<const mdc_jpeg_huff*>(hd + 1);
  for (int k = 0; k < NT; k++) {
    const mdc_jpeg_huff* h = k == 0 ? &hd->dc : k == 1 ? &hd->ac : &chroma[k - 2];
    for (int i = tid; i < 2048; i += THREADS) T.t1[k][i] = h->t1[i];
    for (int i = tid; i < MDC_JPEG_HUFF_SUBTABLES * 32; i += THREADS) (&T.t2[k][0][0])[i] = (&h->t2[0][0])[i];
  }
}
// The block a subsequence begins and leaves to its right neighbours, cleared by the thread that begins it (block index = blocks
// completed before and in the subsequence; an inherited block -- entered at z != 0, none completed -- is its beginner's).
__device__ __forceinline__ void clear_shared_block(int16_t* coef, int first, int nblk, int in_z, int out_z, const ScanGeo& g) {
  const int tb = first + nblk;
  if (out_z != 0 && (nblk > 0 || in_z == 0) && tb < g.nluma) {
    i32x4* blk = reinterpret_cast<i32x4*>(luma_block(coef, tb, g));
#pragma unroll
    for (int r = 0; r < 8; r++) blk[r] = i32x4{0, 0, 0, 0};
  }
}
// quantisation table -> record; record body zero-filled
template <int THREADS>
__device__ __forceinline__ void init_record(int16_t* rec, const mdc_jpeg_stream_header* hd, int pitch, int rows, int tid) {
  if (tid < 64) reinterpret_cast<uint16_t*>(rec)[tid] = hd->quant[tid];
  i32x4* body = reinterpret_cast<i32x4*>(rec + 64);
  const long long n16 = (long long)pitch * rows * 8;  // 16-byte pieces
  for (long long i = tid; i < n16; i += THREADS) body[i] = i32x4{0, 0, 0, 0};
}

// Exclusive prefix sums of two per-thread counts over the workgroup's kHuffThreads threads, and their totals (s_scan: 2 x 17 ints)
__device__ __forceinline__ void scan_pair(int a, int c, int* s_scan, int tid, int* ex_a, int* ex_c, int* tot_a, int* tot_c) {
  constexpr int NW = kHuffThreads / 64;
  const int lane = tid & 63, wave = tid >> 6;
  int ia = a, ic = c;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int ua = __shfl_up(ia, d, 64), uc = __shfl_up(ic, d, 64);
    if (lane >= d) {
      ia += ua;
      ic += uc;
    }
  }
  __syncthreads();  // (s_scan may still be read from an earlier call)
  if (lane == 63) {
    s_scan[wave] = ia;
    s_scan[NW + 1 + wave] = ic;
  }
  __syncthreads();
  if (tid < 2) {
    int* t = s_scan + tid * (NW + 1);
    int acc = 0;
    for (int w = 0; w < NW; w++) {
      const int v = t[w];
      t[w] = acc;
      acc += v;
    }
    t[NW] = acc;
  }
  __syncthreads();
  *ex_a = s_scan[wave] + ia - a;
  *ex_c = s_scan[NW + 1 + wave] + ic - c;
  *tot_a = s_scan[NW];
  *tot_c = s_scan[2 * NW + 1];
}

template <bool COLOR>
__global__ __launch_bounds__(kHuffThreads) void jpeg_huffman_kernel(const unsigned char* __restrict__ streams, long long stream_stride,
                                                                    int16_t* __restrict__ records, long long rec_i16, int W, int H, int pitch,
                                                                    int rows, int* __restrict__ status, unsigned kinds, uint32_t stage_bytes) {
  constexpr int NT = COLOR ? 4 : 2;
  __shared__ HuffLds<NT> s_t;
  __shared__ uint32_t s_bit[kHuffThreads];
  __shared__ unsigned short s_zu[kHuffThreads];  // z | u << 8
  __shared__ int s_scan[2 * (kHuffThreads / 64 + 1)];
  __shared__ int s_flag;
  extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];  // block staging of the write pass (stage_bytes of it, or none)
  const int tid = threadIdx.x;
  const long long f = blockIdx.x;
  const unsigned char* st = streams + f * stream_stride;
  const mdc_jpeg_stream_header* hd = reinterpret_cast<const mdc_jpeg_stream_header*>(st);
  int16_t* rec = records + f * rec_i16;
  // header checks (uniform): anything off -> status 2, record untouched; a stream of another kind: not this kernel's
  const int kind = stream_kind(hd, W, H, pitch, rows, stream_stride);
  if (kind != (COLOR ? 1 : 0)) {
    // (this kernel -- the one that always runs -- reports the streams nobody decodes: a header no kernel takes, or a kind whose
    // kernel the caller did not ask for)
    if (!COLOR && tid == 0 && (kind < 0 || !((kinds >> kind) & 1u))) status[f] = 2;
    return;
  }
  const ScanGeo g = scan_geo(hd, W, H, pitch);
  const uint32_t ecs_bytes = hd->ecs_bytes;
  load_tables<NT, kHuffThreads>(s_t, hd, tid);
  uint32_t* stage = nullptr;
  if (stage_bytes >= 32u * kHuffThreads * 4u) {
    stage = s_dyn + tid;
    for (int k = 0; k < 32; k++) stage[k * kHuffThreads] = 0u;
  }
  // with block staging every block leaves as a whole line or is cleared by the thread that begins it: no zero-fill of the record
  const bool sparse_fill = !COLOR && stage != nullptr;
  if (!sparse_fill) init_record<kHuffThreads>(rec, hd, pitch, rows, tid);
  else if (tid < 64) reinterpret_cast<uint16_t*>(rec)[tid] = hd->quant[tid];
  if (tid == 0) s_flag = 0;
  __syncthreads();
  const uint32_t nbits = ecs_bytes * 8u;
  uint32_t S = (nbits + kHuffThreads - 1) / kHuffThreads;
  S = max(256u, (S + 31u) & ~31u);
  const uint32_t my0 = min(nbits, (uint32_t)tid * S), my1 = min(nbits, my0 + S);
  BitReader b;
  b.base = reinterpret_cast<const uint32_t*>(st + hd->ecs_offset);
  b.src = b.base;
  b.last = (ecs_bytes + 3) / 4 + 2;  // the host pads 16 zero bytes
  int16_t* coef = rec + 64;
  // ---- first guesses: only the exit state of a subsequence matters to its right neighbour, and a decoder started anywhere
  // is on the true path after a few hundred bits -- so the guess comes from the LAST quarter (at least 512 bits) of the subsequence alone
  // (a quarter of a full pass; where it is wrong, the relaxation below finds out)
  const uint32_t guess_bits = max((uint32_t)MDC_EXP_GUESS_MIN_BITS, S / MDC_EXP_GUESS_DIV);  // (high qualities: ~250 bits per block, 512 bits are two blocks)
  uint32_t in_bit = my0, out_bit = my0;
  int in_z = 0, out_z = 0, in_u = 0, out_u = 0, nblk = 0, bad = 0, dcsum = 0;
  if (my0 < my1) {
    const uint32_t from = my1 - my0 > guess_bits ? my1 - guess_bits : my0;
    huff_run<false, COLOR>(b, s_t, from, 0, 0, my1, &out_bit, &out_z, &out_u, &nblk, &bad, &dcsum, nullptr, 0, g);
  }
  s_bit[tid] = out_bit;
  s_zu[tid] = (unsigned short)(out_z | out_u << 8);
  __syncthreads();
  in_bit = tid ? s_bit[tid - 1] : 0u;
  in_z = tid ? (s_zu[tid - 1] & 255) : 0;
  in_u = tid ? (s_zu[tid - 1] >> 8) : 0;
  __syncthreads();
  // ---- relaxation
  bool dirty = true;
  int rounds = 0;
  for (int round = 0; round <= kHuffThreads; round++) {
    rounds++;
    if (dirty) {
      nblk = 0;
      bad = 0;
      dcsum = 0;
      out_bit = in_bit;
      out_z = in_z;
      out_u = in_u;
      if (in_bit < my1) huff_run<false, COLOR>(b, s_t, in_bit, in_z, in_u, my1, &out_bit, &out_z, &out_u, &nblk, &bad, &dcsum, nullptr, 0, g);
    }
    s_bit[tid] = out_bit;
    s_zu[tid] = (unsigned short)(out_z | out_u << 8);
    __syncthreads();
    const uint32_t nb = tid ? s_bit[tid - 1] : 0u;
    const int nz = tid ? (s_zu[tid - 1] & 255) : 0, nu = tid ? (s_zu[tid - 1] >> 8) : 0;
    dirty = (nb != in_bit || nz != in_z || nu != in_u) && my0 < nbits;  // (threads past the end of the stream decode nothing: they just follow)
    in_bit = nb;
    in_z = nz;
    in_u = nu;
    if (!__syncthreads_or(dirty ? 1 : 0)) break;
  }
  // ---- first luma block and DC predictor of every subsequence: exclusive prefix sums of the blocks completed / the DC differences met
  int first, pred, total, total_dc;
  scan_pair(nblk, dcsum, s_scan, tid, &first, &pred, &total, &total_dc);
  (void)total_dc;
  if (sparse_fill) {
    clear_shared_block(coef, first, nblk, in_z, out_z, g);
    __syncthreads();
  }
  // ---- write pass (the true states)
  int bad_w = 0;
  if (in_bit < my1) {
    int dummy = 0, oz, ou;
    uint32_t ob;
    if (sparse_fill) huff_run<true, COLOR, true>(b, s_t, in_bit, in_z, in_u, my1, &ob, &oz, &ou, &dummy, &bad_w, &pred, coef, first, g, stage);
    else huff_run<true, COLOR>(b, s_t, in_bit, in_z, in_u, my1, &ob, &oz, &ou, &dummy, &bad_w, &pred, coef, first, g, stage);
  }
  // fewer blocks than the frame has: truncated or damaged.  More: the 1..7 padding bits after the last block can parse as
  // another (short-coded) block; those are never written.
  if (bad_w || (tid == 0 && total < g.nluma)) s_flag = 1;  // (benign race: every writer writes 1)
  __syncthreads();
#ifdef MDC_EXP_HUFF_ROUNDS  // experiment build: relaxation rounds in the status word's upper bits
  if (tid == 0) status[f] = (s_flag ? 1 : 0) | rounds << 8;
#else
  (void)rounds;
  if (tid == 0) status[f] = s_flag ? 1 : 0;
#endif
}

// ---------------------------------------------------------------------------------------------------------
// The relaxed decoder spread over G workgroups per frame (small batches: a 64-frame chunk of the reader's pipeline would keep
// 64 of the chip's 256 CUs busy with one workgroup per frame).  Segment g of a frame = subsequences [1024 g, 1024 (g+1)) of
// 1024 G; a workgroup relaxes its segment exactly as the one-workgroup kernel does, with a GUESSED entry state for its first
// subsequence, then takes the true one from its left neighbour's published exit state (a decoupled look-back through global
// memory: block f G + g only ever waits for block f G + g - 1, which was dispatched before it; the wait is bounded -- if the
// neighbour does not show up the frame is reported as not decoded and the caller's host decoder takes it) and relaxes again
// where that changed something -- a few subsequences, Huffman streams resynchronise.  Block counts and DC sums travel with the
// state, so the write pass needs no further exchange.  Three launches: jpeg_record_init_kernel (quantisation table, zero-fill
// -- the segments' write passes must find the whole record cleared), jpeg_huffman_split_kernel, jpeg_split_status_kernel (the
// frame's status from its segments' flags).
// ---------------------------------------------------------------------------------------------------------
struct SegState {  // one per (frame, segment), zeroed before the launch
  uint32_t bit;     // FINAL exit state of the segment's last subsequence (flag bit 0)
  uint32_t zu;
  int blocks_incl;  // luma blocks completed in segments 0..g
  int flag;         // bit 0: final; bit 1: the segment met a bad code (set after the write pass); bit 2: no decode here; bit 3: provisional
  uint32_t pbit;    // PROVISIONAL exit state (flag bit 3): the segment relaxed from a guessed entry state -- almost always the final one
  uint32_t pzu;
  int dc_incl;      // sum of the luma DC differences met in segments 0..g: the right neighbour's predictor at its entry (final, like blocks_incl)
  uint32_t pad;
};
constexpr int kHuffMaxSegments = MDC_EXP_HUFF_MAX_SEGMENTS;
static_assert(kHuffMaxSegments == 1 || kHuffMaxSegments == 2 || kHuffMaxSegments == 4 || kHuffMaxSegments == 8, "segments per frame: a power of two up to 8");

__global__ __launch_bounds__(256) void jpeg_record_init_kernel(const unsigned char* __restrict__ streams, long long stream_stride,
                                                               int16_t* __restrict__ records, long long rec_i16, int W, int H, int pitch, int rows,
                                                               int parts, SegState* __restrict__ seg, int G, int stage_blocks) {
  const long long f = blockIdx.x / parts;
  const int part = blockIdx.x % parts;
  if (part == 0 && (int)threadIdx.x < G) seg[f * G + threadIdx.x] = SegState{0u, 0u, 0, 0, 0u, 0u, 0, 0u};  // (the split kernel is the next launch on the stream)
  const mdc_jpeg_stream_header* hd = reinterpret_cast<const mdc_jpeg_stream_header*>(streams + f * stream_stride);
  int16_t* rec = records + f * rec_i16;
  if (part == 0 && threadIdx.x < 64) reinterpret_cast<uint16_t*>(rec)[threadIdx.x] = hd->quant[threadIdx.x];
  // (one-component frames with block staging clear what they must themselves; frames of the other kinds are left to their own kernels,
  // but the three-component split kernel, which has no block staging, wants its record cleared here)
  const int kind = stream_kind(hd, W, H, pitch, rows, stream_stride);
  if (kind != 1 && !(kind == 0 && stage_blocks != 2)) return;
  i32x4* body = reinterpret_cast<i32x4*>(rec + 64);
  const long long n16 = (long long)pitch * rows * 8, per = (n16 + parts - 1) / parts;
  const long long i0 = part * per, i1 = min(n16, i0 + per);
  for (long long i = i0 + threadIdx.x; i < i1; i += 256) body[i] = i32x4{0, 0, 0, 0};
}

template <bool COLOR>
__global__ __launch_bounds__(kHuffThreads) void jpeg_huffman_split_kernel(const unsigned char* __restrict__ streams, long long stream_stride,
                                                                          int16_t* __restrict__ records, long long rec_i16, int W, int H, int pitch,
                                                                          int rows, SegState* __restrict__ seg, int G, uint32_t lds_stream_bytes,
                                                                          int stage_blocks) {
  constexpr int NT = COLOR ? 4 : 2;
  __shared__ HuffLds<NT> s_t;
  __shared__ uint32_t s_bit[kHuffThreads];
  __shared__ unsigned short s_zu[kHuffThreads];
  __shared__ int s_scan[2 * (kHuffThreads / 64 + 1)];
  __shared__ uint32_t s_entry[5];  // the left segment's published state: bit, zu, blocks_incl, ok | final << 1, dc_incl
  // dynamic LDS: block staging of the write pass (one component: 128 KB), or the segment's stream words (three components: the
  // four tables leave no room for the blocks) -- one workgroup per CU either way
  extern __shared__ __attribute__((aligned(16))) uint32_t s_stream[];
  const int tid = threadIdx.x;
  const long long f = blockIdx.x / G;
  const int sg = blockIdx.x % G;
  const unsigned char* st = streams + f * stream_stride;
  const mdc_jpeg_stream_header* hd = reinterpret_cast<const mdc_jpeg_stream_header*>(st);
  SegState* my_seg = seg + f * G + sg;
  const int kind = stream_kind(hd, W, H, pitch, rows, stream_stride);
  if (kind != (COLOR ? 1 : 0)) {  // not this kernel's frame: say so to whoever waits (a final state with no blocks), leave
    if (tid == 0) {
      my_seg->blocks_incl = 0;
      __threadfence();
      atomicExch(&my_seg->flag, 1 | 4);  // bit 2: "no decode here" (the finish kernel leaves such frames to the other kernels' status)
    }
    return;
  }
  int16_t* rec = records + f * rec_i16;
  const ScanGeo g = scan_geo(hd, W, H, pitch);
  const uint32_t ecs_bytes = hd->ecs_bytes;
  load_tables<NT, kHuffThreads>(s_t, hd, tid);
  // (the dynamic LDS holds EITHER the staged blocks of the write pass -- large grids, where the L2's partial-write rate binds -- OR
  // the segment's stream words -- small grids, where the divergent stream loads weigh more: the launcher decides)
  const bool block_staging = !COLOR && stage_blocks != 0 && lds_stream_bytes >= 32u * kHuffThreads * 4u;
  const bool sparse_fill = block_staging && stage_blocks == 2;  // (whole blocks, no zero-fill: see stage_flush_rows)
  uint32_t* stage = nullptr;
  if (block_staging) {
    stage = s_stream + tid;
    for (int k = 0; k < 32; k++) stage[k * kHuffThreads] = 0u;
  }
  __syncthreads();
  const uint32_t nbits = ecs_bytes * 8u;
  const uint32_t nsub = (uint32_t)G * kHuffThreads;
  // subsequences of an ODD number of 32-bit words: thread t starts at word t * sw, and an odd stride puts the 64 lanes of a
  // wave on 64 different LDS banks
  uint32_t sw = ((nbits + nsub - 1) / nsub + 31u) / 32u;
  sw = max(9u, sw | 1u);
  const uint32_t S = sw * 32u;
  const uint32_t gi = (uint32_t)sg * kHuffThreads + (uint32_t)tid;  // this thread's subsequence of the frame
  const uint32_t my0 = (uint32_t)min((unsigned long long)nbits, (unsigned long long)gi * S), my1 = min(nbits, my0 + S);
  BitReader b;
  b.base = reinterpret_cast<const uint32_t*>(st + hd->ecs_offset);
  b.src = b.base;
  b.last = (ecs_bytes + 3) / 4 + 2;
  {  // the segment's words (+ a margin: a subsequence's last symbol and the reader's look-ahead run past its end) -> LDS
    const uint32_t w0 = (uint32_t)sg * kHuffThreads * sw, want = kHuffThreads * sw + 64u;
    const uint32_t have = min(want, b.last + 1u > w0 ? b.last + 1u - w0 : 0u);
    if (!block_staging && have * 4u <= lds_stream_bytes) {
      const i32x4* src = reinterpret_cast<const i32x4*>(b.base + w0);  // (w0 * 4 is a multiple of 16: sw * 1024 words per segment)
      for (uint32_t i = tid; i < have / 4u; i += kHuffThreads) reinterpret_cast<i32x4*>(s_stream)[i] = src[i];
      if ((uint32_t)tid < (have & 3u)) s_stream[(have & ~3u) + tid] = b.base[w0 + (have & ~3u) + tid];  // (never past the stream's last word)
      b.src = s_stream;  // (every word a thread of this segment reads lies in [w0, w0 + have): its subsequence, the reader's look-ahead)
      b.lo = w0;
    }
  }
  __syncthreads();
  int16_t* coef = rec + 64;
  const uint32_t guess_bits = max((uint32_t)MDC_EXP_GUESS_MIN_BITS, S / MDC_EXP_GUESS_DIV);
  uint32_t in_bit = my0, out_bit = my0;
  int in_z = 0, out_z = 0, in_u = 0, out_u = 0, nblk = 0, bad = 0, dcsum = 0;
  if (my0 < my1) {
    const uint32_t from = my1 - my0 > guess_bits ? my1 - guess_bits : my0;
    huff_run<false, COLOR>(b, s_t, from, 0, 0, my1, &out_bit, &out_z, &out_u, &nblk, &bad, &dcsum, nullptr, 0, g);
  }
  // entry of the segment's first subsequence: exact for segment 0, a guess (its own start, z = 0) otherwise
  uint32_t e_bit = my0;
  int e_z = 0, e_u = 0;
  int base_blocks = 0, base_dc = 0;
  bool entry_ok = true;
  // Segment g > 0 relaxes up to three times: from the guess; from its left neighbour's PROVISIONAL exit state (what that one
  // reached from ITS guess -- published by all segments at about the same time, so these second relaxations run side by side
  // instead of one after the other down the chain); and, only if the neighbour's final state turns out to differ from the
  // provisional one, from that.  What travels down the chain serially is then a comparison and the block count.
  bool have_final = sg == 0;
  for (int pass = 0; pass < 3; pass++) {
    s_bit[tid] = out_bit;
    s_zu[tid] = (unsigned short)(out_z | out_u << 8);
    __syncthreads();
    bool dirty = pass == 0;
    {
      const uint32_t nb = tid ? s_bit[tid - 1] : e_bit;
      const int nz = tid ? (s_zu[tid - 1] & 255) : e_z, nu = tid ? (s_zu[tid - 1] >> 8) : e_u;
      if (pass >= 1) dirty = (nb != in_bit || nz != in_z || nu != in_u) && my0 < nbits;
      in_bit = nb;
      in_z = nz;
      in_u = nu;
    }
    __syncthreads();
    for (int round = 0; round <= kHuffThreads; round++) {
      if (dirty) {
        nblk = 0;
        bad = 0;
        dcsum = 0;
        out_bit = in_bit;
        out_z = in_z;
        out_u = in_u;
        if (in_bit < my1) huff_run<false, COLOR>(b, s_t, in_bit, in_z, in_u, my1, &out_bit, &out_z, &out_u, &nblk, &bad, &dcsum, nullptr, 0, g);
      }
      s_bit[tid] = out_bit;
      s_zu[tid] = (unsigned short)(out_z | out_u << 8);
      __syncthreads();
      const uint32_t nb = tid ? s_bit[tid - 1] : e_bit;
      const int nz = tid ? (s_zu[tid - 1] & 255) : e_z, nu = tid ? (s_zu[tid - 1] >> 8) : e_u;
      dirty = (nb != in_bit || nz != in_z || nu != in_u) && my0 < nbits;
      in_bit = nb;
      in_z = nz;
      in_u = nu;
      if (!__syncthreads_or(dirty ? 1 : 0)) break;
    }
    if (have_final) break;
    if (MDC_EXP_HUFF_PROVISIONAL && pass == 0 && tid == kHuffThreads - 1) {  // provisional exit state for the right neighbour
#ifdef MDC_EXP_HUFF_BAD_PROVISIONAL  // fault injection: every provisional state is wrong -- the right neighbour must relax a third time
      my_seg->pbit = out_bit + 9u;
#else
      my_seg->pbit = out_bit;
#endif
      my_seg->pzu = (uint32_t)(out_z | out_u << 8);
      __threadfence();
      __hip_atomic_fetch_or(&my_seg->flag, 8, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- the left segment's exit state: provisional or final after the first relaxation, final after the second (bounded waits)
    if (tid == 0) {
      const SegState* left = my_seg - 1;
      const int want = (MDC_EXP_HUFF_PROVISIONAL && pass == 0) ? (1 | 8) : 1;
      int fl = 0;
      for (int spin = 0; spin < (1 << 20); spin++) {
        fl = __hip_atomic_load(&left->flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if (fl & want) break;
        __builtin_amdgcn_s_sleep(8);
      }
      const bool fin = (fl & 1) != 0;
      s_entry[3] = (uint32_t)(((fl & want) != 0 && !(fl & 4)) ? 1 : 0) | (fin ? 2u : 0u);
      s_entry[0] = __hip_atomic_load(fin ? &left->bit : &left->pbit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_entry[1] = __hip_atomic_load(fin ? &left->zu : &left->pzu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_entry[2] = fin ? (uint32_t)__hip_atomic_load(&left->blocks_incl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
      s_entry[4] = fin ? (uint32_t)__hip_atomic_load(&left->dc_incl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    }
    __syncthreads();
    entry_ok = (s_entry[3] & 1u) != 0;
    have_final = (s_entry[3] & 2u) != 0 || !entry_ok;  // (a neighbour that never showed up ends the exchange: the frame goes to the host decoder)
    const uint32_t n_bit = s_entry[0];
    const int n_z = (int)(s_entry[1] & 255u), n_u = (int)(s_entry[1] >> 8);
    base_blocks = (int)s_entry[2];
    base_dc = (int)s_entry[4];
    __syncthreads();
    const bool same = pass >= 1 && n_bit == e_bit && n_z == e_z && n_u == e_u;
    e_bit = n_bit;
    e_z = n_z;
    e_u = n_u;
    if (same && have_final) break;  // the provisional state was the final one: nothing to relax again
  }
#if defined(MDC_EXP_HUFF_VERIFY) && MDC_EXP_HUFF_VERIFY == 1  // diagnosis: is every thread's (blocks, DC sum, exit state) what a decode from its entry state gives?
  {
    int nb2 = 0, bad2 = 0, dc2 = 0, oz2 = in_z, ou2 = in_u;
    uint32_t ob2 = in_bit;
    if (in_bit < my1) huff_run<false, COLOR>(b, s_t, in_bit, in_z, in_u, my1, &ob2, &oz2, &ou2, &nb2, &bad2, &dc2, nullptr, 0, g);
    if (my0 < nbits) {
      unsigned m = 0;
      if (dc2 != dcsum) m += 1u;
      if (nb2 != nblk) m += 1u << 10;
      if (ob2 != out_bit || oz2 != out_z || ou2 != out_u) m += 1u << 20;
      if (m) atomicAdd(&my_seg->pad, m);
    }
  }
#endif
  // ---- first luma block and DC predictor of every subsequence: the left segments' totals + exclusive prefix sums over this one
  int first, pred, total, total_dc;
  scan_pair(nblk, dcsum, s_scan, tid, &first, &pred, &total, &total_dc);
  first += base_blocks;
  pred += base_dc;
  if (sparse_fill) {  // (before the publish: the right neighbour's write pass comes after its acquire of our final state)
    if (entry_ok) clear_shared_block(coef, first, nblk, in_z, out_z, g);
    __syncthreads();
  }
  // ---- publish: the exit state of the last subsequence + blocks so far (the right neighbour waits for this)
  if (tid == kHuffThreads - 1) {
    my_seg->bit = out_bit;
    my_seg->zu = (uint32_t)(out_z | out_u << 8);
    my_seg->blocks_incl = base_blocks + total;
    my_seg->dc_incl = base_dc + total_dc;
    __threadfence();
    __hip_atomic_store(&my_seg->flag, entry_ok ? 1 : (1 | 4), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- write pass (the true states)
  int bad_w = entry_ok ? 0 : 1;
#if defined(MDC_EXP_HUFF_VERIFY) && MDC_EXP_HUFF_VERIFY == 2
  const int pred_in = pred;
#endif
  if (entry_ok && in_bit < my1) {
    int dummy = 0, oz, ou;
    uint32_t ob;
    if (sparse_fill) huff_run<true, COLOR, true>(b, s_t, in_bit, in_z, in_u, my1, &ob, &oz, &ou, &dummy, &bad_w, &pred, coef, first, g, stage);
    else huff_run<true, COLOR>(b, s_t, in_bit, in_z, in_u, my1, &ob, &oz, &ou, &dummy, &bad_w, &pred, coef, first, g, stage);
#if defined(MDC_EXP_HUFF_VERIFY) && MDC_EXP_HUFF_VERIFY == 2  // diagnosis: the write pass against the counting pass of the same subsequence
    unsigned m = 0;
    if (pred - pred_in != dcsum) m += 1u;
    if (dummy != nblk) m += 1u << 10;
    if (ob != out_bit || oz != out_z || ou != out_u) m += 1u << 20;
    if (m) atomicAdd(&my_seg->pad, m);
#endif
  }
  if (__syncthreads_or(bad_w) && tid == 0) atomicOr(&my_seg->flag, 2);
}

// After the split kernel: the frame's status from its segments' flags (one thread per frame).
template <bool COLOR>
__global__ __launch_bounds__(64) void jpeg_split_status_kernel(const unsigned char* __restrict__ streams, long long stream_stride, int W, int H, int pitch,
                                                               int rows, const SegState* __restrict__ seg, int G, int* __restrict__ status, unsigned kinds,
                                                               int nframes) {
  const long long f = (long long)blockIdx.x * 64 + threadIdx.x;
  if (f >= nframes) return;
  const mdc_jpeg_stream_header* hd = reinterpret_cast<const mdc_jpeg_stream_header*>(streams + f * stream_stride);
  const int kind = stream_kind(hd, W, H, pitch, rows, stream_stride);
  if (kind != (COLOR ? 1 : 0)) {
    if (!COLOR && (kind < 0 || !((kinds >> kind) & 1u))) status[f] = 2;
    return;
  }
  const ScanGeo g = scan_geo(hd, W, H, pitch);
  int flags = 0;
  for (int k = 0; k < G; k++) flags |= seg[f * G + k].flag;
  const bool failed = (flags & (2 | 4)) != 0 || !(flags & 1) || seg[f * G + G - 1].blocks_incl < g.nluma;
#ifdef MDC_EXP_HUFF_VERIFY
  unsigned mism = 0;
  for (int k = 0; k < G; k++) mism += seg[f * G + k].pad;
  status[f] = (failed ? 1 : 0) | (int)((mism & 0xff) | ((mism >> 10) & 0xff) << 8 | ((mism >> 20) & 0xff) << 16) << 8;
#else
  status[f] = failed ? 1 : 0;
#endif
}

// Restart intervals: interval i holds MCUs [i Ri, (i+1) Ri) and begins, byte aligned, at the offset the host recorded, with every
// predictor 0 -- an exact entry state per interval, no exchange between intervals.  Short intervals (a few MCUs) are decoded by one
// thread each, front to back, straight into the record.  Long ones (one per MCU row is what hardware encoders write: 128 intervals
// of 2 KB in a 1280 x 1024 frame) would leave most of a workgroup idle: they are cut into T parts (a power of two <= 64, parts of at
// least ~400 bits), the parts of an interval sit in neighbouring lanes of one wave and relax their entry states as the subsequences
// of the other kernels do -- part 0 exact, the others from a guess, states handed on by lane shuffles, no barrier --, a prefix sum
// over the group gives first block and DC predictor, a last pass writes.  The padding bits at an interval's end may parse as more
// symbols: blocks beyond the interval's last are neither written nor an error; too few blocks are.
// (launched before it: the records of the restart-interval frames cleared, their status 0 -- the decoding workgroups only ever set it)
__global__ __launch_bounds__(256) void jpeg_intervals_init_kernel(const unsigned char* __restrict__ streams, long long stream_stride,
                                                                  int16_t* __restrict__ records, long long rec_i16, int W, int H, int pitch, int rows,
                                                                  int parts, int* __restrict__ status) {
  const long long f = blockIdx.x / parts;
  const int part = blockIdx.x % parts;
  const mdc_jpeg_stream_header* hd = reinterpret_cast<const mdc_jpeg_stream_header*>(streams + f * stream_stride);
  if (stream_kind(hd, W, H, pitch, rows, stream_stride) != 2) return;
  int16_t* rec = records + f * rec_i16;
  if (part == 0 && threadIdx.x < 64) reinterpret_cast<uint16_t*>(rec)[threadIdx.x] = hd->quant[threadIdx.x];
  if (part == 0 && threadIdx.x == 64) status[f] = 0;
  i32x4* body = reinterpret_cast<i32x4*>(rec + 64);
  const long long n16 = (long long)pitch * rows * 8, per = (n16 + parts - 1) / parts;
  const long long i0 = part * per, i1 = min(n16, i0 + per);
  for (long long i = i0 + threadIdx.x; i < i1; i += 256) body[i] = i32x4{0, 0, 0, 0};
}

template <bool COLOR>
__global__ __launch_bounds__(kHuffThreads) void jpeg_huffman_intervals_kernel(const unsigned char* __restrict__ streams, long long stream_stride,
                                                                              int16_t* __restrict__ records, long long rec_i16, int W, int H, int pitch,
                                                                              int rows, int* __restrict__ status, int slices) {
  constexpr int NT = COLOR ? 4 : 2;
  __shared__ HuffLds<NT> s_t;
  __shared__ int s_flag;
  const int tid = threadIdx.x;
  const long long f = blockIdx.x / slices;  // `slices` workgroups per frame take the chunks of 1024 parts in turn: intervals are independent
  const int slice = blockIdx.x % slices;
  const unsigned char* st = streams + f * stream_stride;
  const mdc_jpeg_stream_header* hd = reinterpret_cast<const mdc_jpeg_stream_header*>(st);
  if (stream_kind(hd, W, H, pitch, rows, stream_stride) != 2) return;
  const ScanGeo g = scan_geo(hd, W, H, pitch);
  if ((g.nb > 1) != COLOR) return;  // (the other instantiation's frame)
  int16_t* rec = records + f * rec_i16;
  const int ri = (int)hd->restart_interval, n_iv = (int)hd->n_intervals;
  int lT = 0;  // log2 of the parts per interval
  while (lT < 6 && (hd->ecs_bytes * 8u / (uint32_t)n_iv) >> (lT + 1) >= 384u) lT++;
  if (slice && (long long)slice * kHuffThreads >= ((long long)n_iv << lT)) return;  // (more workgroups than chunks)
  load_tables<NT, kHuffThreads>(s_t, hd, tid);
  if (tid == 0) s_flag = 0;
  __syncthreads();
  const int mcus = g.nluma / g.hv;
  const uint32_t* starts = reinterpret_cast<const uint32_t*>(st + sizeof(mdc_jpeg_stream_header) + (COLOR ? 2 * sizeof(mdc_jpeg_huff) : 0));
  const uint32_t ecs_bytes = hd->ecs_bytes;
  BitReader b;
  b.base = reinterpret_cast<const uint32_t*>(st + hd->ecs_offset);
  b.src = b.base;
  b.last = (ecs_bytes + 3) / 4 + 2;
  int16_t* coef = rec + 64;
  int bad = 0;
  const bool table_ok = !((long long)(n_iv - 1) * ri >= mcus || (long long)n_iv * ri < mcus);  // the interval table describes this frame
  if (!table_ok) bad = 1;
  const int T = 1 << lT;
  const long long nparts = table_ok ? (long long)n_iv << lT : 0;
  for (long long base = (long long)slice * kHuffThreads; base < nparts; base += (long long)slices * kHuffThreads) {  // (uniform over the workgroup)
    const long long id = base + tid;
    bool live = id < nparts;
    const int iv = live ? (int)(id >> lT) : 0, j = (int)(id & (T - 1));
    const uint32_t b0 = starts[iv], b1 = iv + 1 < n_iv ? starts[iv + 1] : ecs_bytes;
    if (live && (b0 > b1 || b1 > ecs_bytes)) {
      bad = 1;
      live = false;
    }
    const uint32_t lo = b0 * 8u, hi = live ? b1 * 8u : lo;
    const int m0 = iv * ri, m1 = min(mcus, m0 + ri);
    ScanGeo gi = g;
    gi.nluma = m1 * g.hv;  // blocks past the interval's last are nobody's
    const int first_block = m0 * g.hv, expected = (m1 - m0) * g.hv;
    if (lT == 0) {  // one thread per interval: exact entry state, one pass
      int nb = 0, bw = 0, pred = 0, oz, ou;
      uint32_t ob = lo;
      oz = 0;
      if (lo < hi) huff_run<true, COLOR>(b, s_t, lo, 0, 0, hi, &ob, &oz, &ou, &nb, &bw, &pred, coef, first_block, gi);
      // too few blocks, or the last one completed by a symbol that reaches into the next interval: damaged
      if (live && (bw || nb < expected || (nb == expected && ob > hi && oz == 0))) bad = 1;
      continue;
    }
    uint32_t S = (((hi - lo) + (uint32_t)T - 1u) >> lT) + 31u & ~31u;
    S = max(S, 32u);
    const uint32_t my0 = (uint32_t)min((unsigned long long)hi, (unsigned long long)lo + (unsigned long long)j * S), my1 = min(hi, my0 + S);
    uint32_t in_bit = my0, out_bit = my0;
    int in_z = 0, out_z = 0, in_u = 0, out_u = 0, nblk = 0, badc = 0, dcsum = 0;
    if (my0 < my1) {  // the guess: what the last quarter of the part gives its right neighbour
      const uint32_t guess_bits = max((uint32_t)MDC_EXP_GUESS_MIN_BITS, S / MDC_EXP_GUESS_DIV);
      const uint32_t from = my1 - my0 > guess_bits ? my1 - guess_bits : my0;
      huff_run<false, COLOR>(b, s_t, from, 0, 0, my1, &out_bit, &out_z, &out_u, &nblk, &badc, &dcsum, nullptr, 0, gi);
    }
    for (int round = 0; round <= T; round++) {  // (every lane of the wave walks the same rounds: the shuffles need them all)
      uint32_t nb = __shfl_up(out_bit, 1, T);
      int nz = __shfl_up(out_z, 1, T), nu = __shfl_up(out_u, 1, T);
      if (j == 0) {
        nb = lo;
        nz = 0;
        nu = 0;
      }
      const bool changed = (round == 0 || nb != in_bit || nz != in_z || nu != in_u) && my0 < hi;
      in_bit = nb;
      in_z = nz;
      in_u = nu;
      if (!__any(changed ? 1 : 0)) break;
      if (changed) {
        nblk = 0;
        badc = 0;
        dcsum = 0;
        out_bit = in_bit;
        out_z = in_z;
        out_u = in_u;
        if (in_bit < my1) huff_run<false, COLOR>(b, s_t, in_bit, in_z, in_u, my1, &out_bit, &out_z, &out_u, &nblk, &badc, &dcsum, nullptr, 0, gi);
      }
    }
    // first block and DC predictor of every part: exclusive prefix sums over the interval's parts
    int ia = nblk, ic = dcsum;
    for (int d = 1; d < T; d <<= 1) {
      const int ua = __shfl_up(ia, d, T), uc = __shfl_up(ic, d, T);
      if (j >= d) {
        ia += ua;
        ic += uc;
      }
    }
    const int total = __shfl(ia, T - 1, T);
    const uint32_t end_bit = __shfl(out_bit, T - 1, T);
    const int end_z = __shfl(out_z, T - 1, T);
    int bw = 0, pred = ic - dcsum, dummy = 0, oz, ou;
    uint32_t ob;
    if (in_bit < my1) huff_run<true, COLOR>(b, s_t, in_bit, in_z, in_u, my1, &ob, &oz, &ou, &dummy, &bw, &pred, coef, first_block + ia - nblk, gi);
    if (live && (bw || total < expected || (total == expected && end_bit > hi && end_z == 0))) bad = 1;  // (as above)
  }
  if (bad) s_flag = 1;  // (benign race: every writer writes 1)
  __syncthreads();
  if (tid == 0 && s_flag) atomicOr(&status[f], 1);
}

}  // namespace

size_t jpeg_huffman_scratch_bytes(int64_t nframes) { return (size_t)nframes * kHuffMaxSegments * sizeof(SegState); }
static_assert(sizeof(SegState) == 32, "SegState: two per 64-byte line");
int jpeg_huffman_segments(int64_t nframes) {
  // 8 workgroups per frame up to 16 frames, 4 up to 64 (the reader's chunk: one workgroup per CU), 2 up to 128; beyond, one
  // workgroup per frame fills the chip (profiles/r04_experiments/04_*, 05_*, 08_*: 8 x 32 workgroups were slower than 4 x 32)
  const int g = nframes <= 16 ? 8 : nframes <= 64 ? 4 : nframes <= 128 ? 2 : 1;
  return g < kHuffMaxSegments ? g : kHuffMaxSegments;
}

hipError_t launch_jpeg_huffman(const void* d_streams, int64_t stream_stride, void* d_records, int64_t record_bytes, int w, int h, int blocks_w,
                               int blocks_rows, int64_t nframes, int* d_status, hipStream_t s, unsigned kinds, void* d_scratch) {
  if (nframes <= 0) return hipSuccess;
  const int bw_used = (w + 7) / 8, bh_used = (h + 7) / 8;
  if (w <= 0 || h <= 0 || blocks_w < bw_used || blocks_rows < bh_used || record_bytes % 16 != 0 || stream_stride % 16 != 0 ||
      stream_stride < (int64_t)sizeof(mdc_jpeg_stream_header) + 32 || record_bytes < 128 + (int64_t)blocks_w * blocks_rows * 128 ||
      ((reinterpret_cast<uintptr_t>(d_records) | reinterpret_cast<uintptr_t>(d_streams)) & 15) != 0 || nframes > (1ll << 30))
    return hipErrorInvalidValue;
  // One launch per kind of stream in the batch (`kinds`: bit 0 one component, bit 1 three components, bit 2 restart intervals;
  // a caller that cannot look into the streams passes all three): a workgroup whose frame is of another kind leaves at once.
  // The one-component kernel always runs: it is the one that reports headers no kernel takes.
  const unsigned char* st = static_cast<const unsigned char*>(d_streams);
  int16_t* rec = static_cast<int16_t*>(d_records);
  kinds |= 1u;
  const int G = d_scratch ? jpeg_huffman_segments(nframes) : 1;
  const long long rec_i16 = record_bytes / 2;
  if (G > 1) {  // small batch: several workgroups per frame (d_scratch: jpeg_huffman_scratch_bytes(nframes), any content)
    SegState* seg = static_cast<SegState*>(d_scratch);
    // dynamic LDS: one component, >= 48 frames -- 128 KB of block staging for the write pass (0.66 -> 0.55 ms per 64 frames; below
    // that it costs more than it saves: 0.46 -> 0.52 ms for 32); else the segment's stream bytes (96 KB; 72 with four tables; a
    // stream too long for it is read from global memory)
    const int stage_blocks = nframes > 64 ? 2 : nframes >= 48 ? 1 : 0;  // (2: whole blocks, no zero-fill of one-component records)
    const size_t pad = stage_blocks ? 128 * 1024 : 96 * 1024;
    hipError_t e = hipSuccess;
    {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&jpeg_huffman_split_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
      if (e == hipSuccess && (kinds & 2u))
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&jpeg_huffman_split_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
      if (e != hipSuccess) return e;
    }
    jpeg_record_init_kernel<<<(unsigned)(nframes * 8), 256, 0, s>>>(st, stream_stride, rec, rec_i16, w, h, blocks_w, blocks_rows, 8, seg, G, stage_blocks);
    jpeg_huffman_split_kernel<false><<<(unsigned)(nframes * G), kHuffThreads, pad, s>>>(st, stream_stride, rec, rec_i16, w, h, blocks_w, blocks_rows, seg, G,
                                                                                        (uint32_t)pad, stage_blocks);
    jpeg_split_status_kernel<false><<<(unsigned)((nframes + 63) / 64), 64, 0, s>>>(st, stream_stride, w, h, blocks_w, blocks_rows, seg, G, d_status, kinds, (int)nframes);
    if (kinds & 2u) {
      e = hipMemsetAsync(seg, 0, (size_t)nframes * G * sizeof(SegState), s);
      if (e != hipSuccess) return e;
      jpeg_huffman_split_kernel<true><<<(unsigned)(nframes * G), kHuffThreads, 72 * 1024, s>>>(st, stream_stride, rec, rec_i16, w, h, blocks_w, blocks_rows, seg, G,
                                                                                                 (uint32_t)(72 * 1024), 0);
      jpeg_split_status_kernel<true><<<(unsigned)((nframes + 63) / 64), 64, 0, s>>>(st, stream_stride, w, h, blocks_w, blocks_rows, seg, G, d_status, kinds, (int)nframes);
    }
  } else {
    const size_t stage = 128 * 1024;  // block staging of the one-component kernel's write pass (one workgroup per CU)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&jpeg_huffman_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)stage);
    if (e != hipSuccess) return e;
    jpeg_huffman_kernel<false><<<(unsigned)nframes, kHuffThreads, stage, s>>>(st, stream_stride, rec, rec_i16, w, h, blocks_w, blocks_rows, d_status, kinds,
                                                                              (uint32_t)stage);
    if (kinds & 2u)
      jpeg_huffman_kernel<true><<<(unsigned)nframes, kHuffThreads, 0, s>>>(st, stream_stride, rec, rec_i16, w, h, blocks_w, blocks_rows, d_status, kinds, 0u);
  }
  if (kinds & 4u) {  // (a frame is taken by the instantiation for its component count; small batches: several workgroups per frame)
    const int slices = (int)std::max<int64_t>(1, std::min<int64_t>(8, 256 / nframes));
    jpeg_intervals_init_kernel<<<(unsigned)(nframes * 8), 256, 0, s>>>(st, stream_stride, rec, rec_i16, w, h, blocks_w, blocks_rows, 8, d_status);
    jpeg_huffman_intervals_kernel<false><<<(unsigned)(nframes * slices), kHuffThreads, 0, s>>>(st, stream_stride, rec, rec_i16, w, h, blocks_w, blocks_rows, d_status, slices);
    jpeg_huffman_intervals_kernel<true><<<(unsigned)(nframes * slices), kHuffThreads, 0, s>>>(st, stream_stride, rec, rec_i16, w, h, blocks_w, blocks_rows, d_status, slices);
  }
  return hipGetLastError();
}

hipError_t launch_jpeg_idct(const void* d_records, int64_t record_bytes, uint8_t* d_frames, int w, int h, int blocks_w, int blocks_rows,
                            int64_t nframes, hipStream_t s) {
  if (nframes <= 0) return hipSuccess;
  const int bw_used = (w + 7) / 8, bh_used = (h + 7) / 8;
  if (w <= 0 || h <= 0 || blocks_w < bw_used || blocks_rows < bh_used || record_bytes % 16 != 0 ||
      record_bytes < 128 + (int64_t)blocks_w * blocks_rows * 128 || (reinterpret_cast<uintptr_t>(d_records) & 15) != 0)
    return hipErrorInvalidValue;
  const long long n = (long long)bw_used * bh_used * nframes;  // blocks
  jpeg_idct_kernel<<<(unsigned)((n + kIdctBlocks - 1) / kIdctBlocks), kIdctThreads, 0, s>>>(static_cast<const int16_t*>(d_records), record_bytes / 2, d_frames, w, h, blocks_w,
                                                            bw_used, bh_used, nframes);
  return hipGetLastError();
}

}  // namespace mdc
