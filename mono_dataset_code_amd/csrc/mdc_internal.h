// Internal interface between the C-ABI layer (mdc_capi.hip, mdc_plan.hip, mdc_host_calls.hip, mdc_pipeline.hip) and the gfx950
// kernels (mdc_kernels.hip).  Not installed; see include/mdc_hip.h for the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fov_point_model.h"
#include "mdc_build_config.h"

namespace mdc {

// Geometry of the tiled kernel: tile_w x tile_h outputs per workgroup, one lane per output column
// (tile_w / 64 waves side by side), `rpt` vertically consecutive output rows per thread  ->  tile_w * tile_h / rpt threads.
// Legal shapes: 64 x {16, 32, 60, 64} (256 / 512 / 960 / 1024 threads), 128 x {16, 32} (512 / 1024), 4 rows per thread.
// (The kernel is written for `rpt` rows per thread; 320 x 16 and 640 x 8 tiles with 8 rows per thread were built, are bit-identical
// and 15-20 % slower -- profiles/r04_experiments/01_* -- and are not instantiated.)
struct TileShape {
  int w, h;
};
constexpr TileShape kTileShapes[] = {{64, 16}, {64, 32}, {64, 60}, {64, 64}, {128, 16}, {128, 32}};
constexpr bool tile_shape_ok(int w, int h) {
  for (const TileShape& t : kTileShapes)
    if (t.w == w && t.h == h) return true;
  return false;
}
constexpr int tile_rpt(int w, int h) { return w >= 320 ? 8 : 4; }             // output rows per thread
constexpr int tile_threads(int w, int h) { return w * h / tile_rpt(w, h); }  // workgroup size
constexpr int kTileMaxChunks = 3;     // 16-byte chunks a thread may stage per frame (raw u8 frames)
constexpr int kTileMaxChunksF32 = 4;  // same for float frames (a window holds 4x the bytes)
constexpr int kLutRep = MDC_EXP_LUT_REP;  // LDS replicas of the 256-entry response LUT (32 = one per bank, conflict-free)
constexpr uint32_t kOutside = 0xfffffff0u;  // buffer offset beyond any frame: the access is dropped by the range check

struct RemapArgs {
  const float* lut;    // 256 floats: response LUT variant (identity or GInv; [255] = NaN when killing overexposed)
  const float* vinv;   // vignetteMapInv (in_w*in_h) or nullptr
  const float* rx;     // remapX (out_w*out_h)
  const float* ry;     // remapY
  int in_w, in_h, out_w, out_h;
};

// Plan of the tiled kernel for one remap (built on the host by plan_tiles, mdc_plan.hip).
// The source window of an output tile is the exact set of raw-frame bytes its bilinear taps
// touch, row by row: for every source row the run [x0, x0 + 16 n) of aligned 16-byte chunks
// covering the taps of that row (x0 % 16 == 0).  Chunks are numbered row by row; chunk c is
// staged at LDS byte 16 c of the window buffer.
struct TilePlan {
  const uint32_t* d_chunks;  // [n_tiles][chunk_cap] byte offset of chunk c inside a frame, kOutside for c >= nch
  const int* d_nch;          // [n_tiles] chunks of the tile's window (0 = every output black / outside)
  const uint32_t* d_taps;    // [out_w*out_h] LDS byte offset of tap (xi,yi) | offset of tap (xi,yi+1) << 16
  const int* d_order;        // block -> tile (or -1), n_blocks entries, n_blocks % 8 == 0: block b runs on XCD b % 8
  int n_blocks;
  int n_tiles, tiles_x;
  int tile_w;      // output columns per tile: 64 or 128
  int tile_h;      // output rows per tile: 16, 32, 60 or 64 (see kTileShapes)
  int chunk_cap;   // row length of d_chunks: kTileMaxChunks (F32: kTileMaxChunksF32) x threads, padded with kOutside
  int win_bytes;   // LDS bytes of one window buffer: 16 * max nch, rounded up to 1 KiB
  int nbuf;        // window buffers per workgroup (2..4): nbuf-1 frames are staged ahead
  bool has_black;  // some output carries the (-1,-1) sentinel
  bool interleave; // frame groups take every G-th frame instead of fpb consecutive ones
  bool taper = true;  // large launches end on frame groups of fpb/2, fpb/4, fpb/8 (MDC_OPT_TAIL_TAPER)
};

// Plan of the wave-private strip kernel (remap_strip_kernel): a WAVE owns a 128 x 8 output tile (lane l: columns l and
// l + 64, 8 rows), stages its own source window (<= kStripChunkCap chunks, dense list, kOutside-padded), converts it to
// lut * vignette floats and samples those.  Offsets in d_taps are bytes inside the wave's FLOAT window (4 x the u8 offset).
constexpr int kStripTileW = 128, kStripTileH = 8;
constexpr int kStripChunkCap = 128;  // two LDS-DMA instructions per wave and frame at most
constexpr int kStripWaves = MDC_EXP_STRIP_WAVES;  // waves (tiles) per workgroup: they share the LUT replicas
struct StripPlan {
  const uint32_t* d_chunks;  // [n_tiles][kStripChunkCap]
  const int* d_nch;          // [n_tiles]
  const uint32_t* d_taps;    // [out_w*out_h] float-window byte offset of tap (xi,yi) | of tap (xi,yi+1) << 16
  const int* d_order;        // block -> group of kStripWaves consecutive tiles (or -1); block b runs on XCD b % 8
  int n_blocks, n_tiles, tiles_x;
  int win_bytes;             // u8 window bytes of a wave: 16 * max nch, rounded up to 64
  int passes;                // convert passes per frame (64 dwords each): 2, 3, 4, 5 or 8
  int nbuf;                  // u8 windows per wave: 1 or 2
  bool interleave;
};
#if MDC_EXP_STRIP_FAKE_GRAD
extern float* g_fake_grad_dI;
extern float* g_fake_grad_abs;
#endif
hipError_t launch_remap_strip_u8(const uint8_t* d_in, float* d_out, const RemapArgs& a, const StripPlan& p, int64_t nframes, int fpb,
                                 hipStream_t s, float* d_l1 = nullptr, float* d_l2 = nullptr, float* d_l3 = nullptr);
size_t strip_lds_bytes(int win_bytes, int nbuf, int waves);

// out[f][i] = lut[in[f][i]] (* vinv[i]) over nframes frames of npix pixels.
hipError_t launch_unmap(const uint8_t* d_in, float* d_out, const float* d_lut, const float* d_vinv, int64_t npix,
                        int64_t nframes, int fpb, hipStream_t s);

// Fused LUT (* vignette) + bilinear remap, u8 frames, direct global gather.
hipError_t launch_remap_gather_u8(const uint8_t* d_in, float* d_out, const RemapArgs& a, int64_t nframes, int fpb,
                                  hipStream_t s);
// Bilinear remap of float frames (UndistorterFOV::undistort<float>).
hipError_t launch_remap_gather_f32(const float* d_in, float* d_out, const RemapArgs& a, int64_t nframes, int fpb,
                                   hipStream_t s);
// Fused LUT (* vignette) + bilinear remap, u8 frames, source windows staged in LDS.
// d_l1..d_l3 (optional): levels 1..3 of the 2x2 box pyramid of every output frame, written by the same
// launch (needs whole tiles: out_w % tile_w == 0, out_h % tile_h == 0, tile_h % 8 == 0).
hipError_t launch_remap_tiled_u8(const uint8_t* d_in, float* d_out, const RemapArgs& a, const TilePlan& p,
                                 int64_t nframes, int fpb, hipStream_t s, float* d_l1 = nullptr, float* d_l2 = nullptr,
                                 float* d_l3 = nullptr);
// undistort<float>: the same tiled kernel on float frames (16-byte chunks of 4 pixels, no LUT).
hipError_t launch_remap_tiled_f32(const float* d_in, float* d_out, const RemapArgs& a, const TilePlan& p,
                                  int64_t nframes, int fpb, hipStream_t s);
size_t tiled_lds_bytes(int win_bytes, int nbuf, bool lut = true);  // (LUT replicas +) nbuf window buffers
size_t tiled_pyramid_lds_bytes(int tile_w, int tile_h);  // + level-2 hand-over rows of the fused pyramid
constexpr size_t kLdsPerCU = 160 * 1024;

// Linear read of rows [y0, y1] x byte columns [x0, x1] (widened to whole 128-byte lines) of each of nframes u8 frames
// (frame_bytes apart, row_pitch bytes a row): a software prefetch into the Infinity Cache ahead of a remap launch over
// those frames (d_sink: any device word, never written in practice)
hipError_t launch_prefetch_rows(const uint8_t* d_frames, int64_t frame_bytes, int row_pitch, int x0, int x1, int y0, int y1, int64_t nframes,
                                uint32_t* d_sink, hipStream_t s);

// One 2x2 box level: dst (w/2 x h/2) from src (w x h), nframes images each.
hipError_t launch_pyramid_level(const float* d_src, float* d_dst, int w, int h, int64_t nframes, hipStream_t s);

// FOV lens model (struct DistortModel) and its per-point arithmetic: fov_point_model.h
// (x, y) rectified pixel -> raw pixel, in place (UndistorterFOV::distortCoordinates).
hipError_t launch_distort_points(float* d_x, float* d_y, int64_t n, const DistortModel& m, hipStream_t s);

// vignetteCalib solver half-iterations (src/main_vignetteCalib.cpp:400-448, :455-527); d_er = {E, R}
hipError_t launch_vcal_plane_step(const float* d_images, const float* d_p2x, const float* d_p2y, int n, int wI, int hI, int np,
                                  float* d_plane_color, const float* d_vig, int oth2, float* d_ff, float* d_fc, double* d_er,
                                  hipStream_t s);
hipError_t launch_vcal_vignette_step(const float* d_images, const float* d_p2x, const float* d_p2y, int n, int wI, int hI, int np,
                                     const float* d_plane_color, float* d_vig, int oth2, float* d_tt, float* d_ct, double* d_er,
                                     unsigned* d_max_bits, hipStream_t s);

// The vignette half-iteration as an ordered gather over a prebuilt per-pixel contribution index (mdc_vcal.hip):
// bit-identical to the reference's sequential scatter, no atomics.
struct VcalIndex;
hipError_t vcal_index_build(const float* d_images, const float* d_p2x, const float* d_p2y, int n, int wI, int hI, int np,
                            hipStream_t s, VcalIndex** out);
void vcal_index_free(VcalIndex* ix);
long long vcal_index_bytes(const VcalIndex* ix);
long long vcal_index_entries(const VcalIndex* ix);
hipError_t launch_vcal_vignette_step_indexed(const VcalIndex* ix, const float* d_plane_color, float* d_vig, int oth2, float* d_tt,
                                             float* d_ct, double* d_er, unsigned* d_max_bits, hipStream_t s);

// vignetteCalib: image = meanExposure * image / exposure_time[image] (:286-291), in place
hipError_t launch_vcal_scale_images(float* d_images, int n, int64_t npix, float mean_exposure, const float* d_exposure, hipStream_t s);
// vignetteCalib: gradient mask of n stacked w x h images, in place (:293-301)
hipError_t launch_vcal_gradient_mask(float* d_images, int n, int wI, int hI, int max_abs_grad, hipStream_t s);
// vignetteCalib: NaN coordinates for plane points outside the image (:345-357)
hipError_t launch_vcal_mask_coords(float* d_x, float* d_y, int64_t n, int wI, int hI, hipStream_t s);
// vignetteCalib's output smoothing (:541-566): four NaN-aware 3 x 3 mean passes; d_tt = result, d_ct = scratch
hipError_t launch_vcal_smooth(const float* d_vig, int wI, int hI, float* d_tt, float* d_ct, hipStream_t s);

// JPEG ingest, device half: coefficient records (per frame: 64 x u16 luma quantisation table, then blocks_rows x blocks_w blocks
// of 64 quantised int16 coefficients, natural order; record_bytes apart) -> 8-bit frames (nframes * w * h), libjpeg's islow
// inverse DCT (mdc_jpeg.hip)
// Huffman decoding of baseline JPEG streams -- one or three components, with or without restart intervals
// (mdc_jpeg_stream_header + tables + unstuffed entropy-coded bytes, one per frame, stream_stride bytes apart) -- into LUMA
// coefficient records; d_status[f]: 0 decoded, 1 bad code / block count, 2 bad header.
// kinds: which kinds of stream the batch may hold (bit 0: one component, bit 1: three components, bit 2: restart intervals) -- one launch each
hipError_t launch_jpeg_huffman(const void* d_streams, int64_t stream_stride, void* d_records, int64_t record_bytes, int w, int h, int blocks_w,
                               int blocks_rows, int64_t nframes, int* d_status, hipStream_t s, unsigned kinds = 7u, void* d_scratch = nullptr);
// d_scratch (optional, jpeg_huffman_scratch_bytes(nframes) bytes): lets a small batch spread every frame's stream over
// jpeg_huffman_segments(nframes) workgroups that hand their exit states on through it
size_t jpeg_huffman_scratch_bytes(int64_t nframes);
int jpeg_huffman_segments(int64_t nframes);
hipError_t launch_jpeg_idct(const void* d_records, int64_t record_bytes, uint8_t* d_frames, int w, int h, int blocks_w, int blocks_rows,
                            int64_t nframes, hipStream_t s);

// DSO hand-off of one pyramid level: (I, dx, dy) triples + absSquaredGrad (see mdc_vcal.hip)
hipError_t launch_gradients(const float* d_level, float* d_dI, float* d_abs2, int w, int h, int64_t nframes, hipStream_t s);
// ... of up to four levels in one launch
hipError_t launch_gradients_levels(int n_levels, const float* const* d_src, float* const* d_dI, float* const* d_abs2, const int* w,
                                   const int* h, int64_t nframes, hipStream_t s);

}  // namespace mdc
