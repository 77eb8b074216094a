// Internal interface between the C-ABI layer (mdc_capi.hip) and the gfx950
// kernels (mdc_kernels.hip).  Not installed; see include/mdc_hip.h for the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mdc {

// Source window of one output tile of the LDS-tiled kernel.  The window is the
// set of raw-frame bytes [y0, y0+rows) x [x0, x0 + 16*cpr): x0 is 16-byte
// aligned so every row is a whole number of 16-byte chunks of the frame.
struct TileDesc {
  int x0, y0;  // top-left source pixel of the window
  int rows;    // window height (0 = every output of the tile is black)
  int cpr;     // 16-byte chunks per window row; LDS pitch = 16*cpr bytes
};

// Geometry of the tiled kernel: kTileW x tile_h outputs per workgroup, one lane per
// output column, 4 output rows per thread  ->  16*tile_h threads
// (tile_h = 16: 256 threads; tile_h = 32: 512 threads).
constexpr int kTileW = 64;
constexpr int kTileMaxChunks = 3;  // 16-byte chunks a thread may stage per frame
constexpr int kLutRep = 32;        // LDS replicas of the 256-entry response LUT (one per bank)

struct RemapArgs {
  const float* lut;    // 256 floats: response LUT variant (identity or GInv; [255] = NaN when killing overexposed)
  const float* vinv;   // vignetteMapInv (in_w*in_h) or nullptr
  const float* rx;     // remapX (out_w*out_h)
  const float* ry;     // remapY
  int in_w, in_h, out_w, out_h;
};

struct TilePlan {
  const TileDesc* d_tiles;
  const int* d_order;  // block -> tile (or -1), n_blocks entries, n_blocks % 8 == 0: block b runs on XCD b % 8
  int n_blocks;
  int n_tiles, tiles_x;
  int tile_h;     // 16 or 32 output rows per tile
  int win_bytes;  // LDS bytes of one staging buffer (max over tiles of rows*cpr*16)
};

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
// lut_rep = LDS replicas of the response LUT (32 = conflict-free, 16/8 = smaller LDS footprint).
hipError_t launch_remap_tiled_u8(const uint8_t* d_in, float* d_out, const RemapArgs& a, const TilePlan& p,
                                 int64_t nframes, int fpb, int lut_rep, int taps, hipStream_t s);
size_t tiled_lds_bytes(int win_bytes, int lut_rep);

// One 2x2 box level: dst (w/2 x h/2) from src (w x h), nframes images each.
hipError_t launch_pyramid_level(const float* d_src, float* d_dst, int w, int h, int64_t nframes, hipStream_t s);

hipError_t launch_synth(uint8_t* d_out, int64_t first_frame, int64_t nframes, int npix, uint32_t seed, hipStream_t s);

}  // namespace mdc
