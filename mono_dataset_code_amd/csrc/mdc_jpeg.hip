// JPEG ingest, device half (SURVEY.md section 8 row f2 "later GPU JPEG"): the host keeps the serial part of JPEG
// decoding -- Huffman decoding of the entropy-coded segments, csrc/host/image_codecs*.cpp -- and ships the quantised luma
// coefficients (64 int16 per 8x8 block, natural order) + the quantisation table; this kernel dequantises, runs libjpeg's
// "islow" integer inverse DCT (jidctint.c: the same constants, the same two passes, the same rounding as the host decoder,
// which is itself pinned bit for bit to libjpeg-turbo in tests/test_reader_cpu.py) and writes the 8-bit frame straight
// into the buffer the fused photometric + remap kernel reads.  Integer arithmetic only: bytes identical to the host path.
//
// One thread per block (a 1280 x 1024 frame has 20480): a lane reads its block's 128 bytes as eight 16-byte loads and
// writes eight 8-byte row pieces; neighbouring lanes own neighbouring blocks of a block row, so a wave's stores of one
// row are 512 contiguous bytes.  The arithmetic is a few microseconds per frame on the whole chip -- the stage is bound
// by the PCIe transfer of the coefficients (2 bytes per pixel), which is what it trades for the host's IDCT time.
#include "mdc_internal.h"

namespace mdc {
namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int descale(long long x, int n) { return (int)((x + (1LL << (n - 1))) >> n); }
__device__ __forceinline__ unsigned clamp_sample(int x) {
  x += 128;
  return (unsigned)(x < 0 ? 0 : (x > 255 ? 255 : x));
}

// one 1-D pass of jidctint.c on (c0..c7) -> (o0..o7) before descaling; even part / odd part as the reference names them
struct Pass {
  long long tmp10, tmp11, tmp12, tmp13, tmp0, tmp1, tmp2, tmp3;
};
__device__ __forceinline__ Pass idct_1d(long long c0, long long c1, long long c2, long long c3, long long c4, long long c5, long long c6,
                                        long long c7) {
  const long long F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299, F1_847 = 15137,
                  F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
  Pass p;
  long long z2 = c2, z3 = c6;
  long long z1 = (z2 + z3) * F0_541;
  long long tmp2 = z1 + z3 * (-F1_847), tmp3 = z1 + z2 * F0_765;
  long long tmp0 = (c0 + c4) * (1LL << 13), tmp1 = (c0 - c4) * (1LL << 13);
  p.tmp10 = tmp0 + tmp3;
  p.tmp13 = tmp0 - tmp3;
  p.tmp11 = tmp1 + tmp2;
  p.tmp12 = tmp1 - tmp2;
  tmp0 = c7;
  tmp1 = c5;
  tmp2 = c3;
  tmp3 = c1;
  z1 = tmp0 + tmp3;
  z2 = tmp1 + tmp2;
  z3 = tmp0 + tmp2;
  long long z4 = tmp1 + tmp3;
  const long long z5 = (z3 + z4) * F1_175;
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
  p.tmp0 = tmp0 + z1 + z3;
  p.tmp1 = tmp1 + z2 + z4;
  p.tmp2 = tmp2 + z2 + z3;
  p.tmp3 = tmp3 + z1 + z4;
  return p;
}

// records: per frame [64 x u16 quantisation table, natural order][blocks_rows x blocks_w x 64 int16], rec_i16 int16 apart
__global__ __launch_bounds__(256) void jpeg_idct_kernel(const int16_t* __restrict__ records, long long rec_i16, uint8_t* __restrict__ frames,
                                                        int W, int H, int blocks_w, int bw_used, int bh_used, long long nframes) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long per_frame = (long long)bw_used * bh_used;
  if (t >= per_frame * nframes) return;
  const long long f = t / per_frame;
  const int r = (int)(t - f * per_frame);
  const int by = r / bw_used, bx = r - by * bw_used;
  const int16_t* rec = records + f * rec_i16;
  const int16_t* blk = rec + 64 + ((long long)by * blocks_w + bx) * 64;
  const uint16_t* q = reinterpret_cast<const uint16_t*>(rec);
  int c[64];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const i32x4 v = *reinterpret_cast<const i32x4*>(blk + 8 * k);  // 8 coefficients
#pragma unroll
    for (int j = 0; j < 4; j++) {
      c[8 * k + 2 * j] = (int)(short)(v[j] & 0xffff) * (int)q[8 * k + 2 * j];  // int * int as on the host
      c[8 * k + 2 * j + 1] = (int)(short)((unsigned)v[j] >> 16) * (int)q[8 * k + 2 * j + 1];
    }
  }
  // pass 1: columns -> workspace (scaled by 2^PASS1_BITS)
  int ws[64];
#pragma unroll
  for (int col = 0; col < 8; col++) {
    const Pass p = idct_1d(c[col], c[8 + col], c[16 + col], c[24 + col], c[32 + col], c[40 + col], c[48 + col], c[56 + col]);
    ws[0 * 8 + col] = descale(p.tmp10 + p.tmp3, 11);
    ws[7 * 8 + col] = descale(p.tmp10 - p.tmp3, 11);
    ws[1 * 8 + col] = descale(p.tmp11 + p.tmp2, 11);
    ws[6 * 8 + col] = descale(p.tmp11 - p.tmp2, 11);
    ws[2 * 8 + col] = descale(p.tmp12 + p.tmp1, 11);
    ws[5 * 8 + col] = descale(p.tmp12 - p.tmp1, 11);
    ws[3 * 8 + col] = descale(p.tmp13 + p.tmp0, 11);
    ws[4 * 8 + col] = descale(p.tmp13 - p.tmp0, 11);
  }
  // pass 2: rows -> samples
  uint8_t* dst = frames + f * (long long)W * H + (long long)(by * 8) * W + bx * 8;
  const bool whole = bx * 8 + 8 <= W && (W & 7) == 0 && ((reinterpret_cast<uintptr_t>(frames) | (uintptr_t)((long long)W * H)) & 7) == 0;
#pragma unroll
  for (int row = 0; row < 8; row++) {
    const int* w = ws + row * 8;
    const Pass p = idct_1d(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]);
    unsigned o[8];
    o[0] = clamp_sample(descale(p.tmp10 + p.tmp3, 18));
    o[7] = clamp_sample(descale(p.tmp10 - p.tmp3, 18));
    o[1] = clamp_sample(descale(p.tmp11 + p.tmp2, 18));
    o[6] = clamp_sample(descale(p.tmp11 - p.tmp2, 18));
    o[2] = clamp_sample(descale(p.tmp12 + p.tmp1, 18));
    o[5] = clamp_sample(descale(p.tmp12 - p.tmp1, 18));
    o[3] = clamp_sample(descale(p.tmp13 + p.tmp0, 18));
    o[4] = clamp_sample(descale(p.tmp13 - p.tmp0, 18));
    if (by * 8 + row >= H) break;
    uint8_t* line = dst + (long long)row * W;
    if (whole) {
      const unsigned lo = o[0] | o[1] << 8 | o[2] << 16 | o[3] << 24, hi = o[4] | o[5] << 8 | o[6] << 16 | o[7] << 24;
      *reinterpret_cast<uint2*>(line) = make_uint2(lo, hi);
    } else {
      for (int x = 0; x < 8 && bx * 8 + x < W; x++) line[x] = (uint8_t)o[x];
    }
  }
}

}  // namespace

hipError_t launch_jpeg_idct(const void* d_records, int64_t record_bytes, uint8_t* d_frames, int w, int h, int blocks_w, int blocks_rows,
                            int64_t nframes, hipStream_t s) {
  if (nframes <= 0) return hipSuccess;
  const int bw_used = (w + 7) / 8, bh_used = (h + 7) / 8;
  if (w <= 0 || h <= 0 || blocks_w < bw_used || blocks_rows < bh_used || record_bytes % 16 != 0 ||
      record_bytes < 128 + (int64_t)blocks_w * blocks_rows * 128 || (reinterpret_cast<uintptr_t>(d_records) & 15) != 0)
    return hipErrorInvalidValue;
  const long long n = (long long)bw_used * bh_used * nframes;
  jpeg_idct_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(static_cast<const int16_t*>(d_records), record_bytes / 2, d_frames, w, h, blocks_w,
                                                            bw_used, bh_used, nframes);
  return hipGetLastError();
}

}  // namespace mdc
