// libmdc_hip.so: host-side plans of the remap kernels (TilePlan / StripPlan, mdc_internal.h) -- which 16-byte chunks of a
// raw frame an output tile stages, where each output's taps lie inside the staged window, which XCD runs which tile.
#include "mdc_ctx.h"

namespace mdc {
// Placement table of the tiled kernel: entry b = tile run by block b of a frame group, -1 = none.
// The dispatcher deals blocks round-robin over the 8 XCDs (block b -> XCD b % 8, slot b / 8), so
// XCD k runs the tiles of entries k, k+8, k+16, ...  Neighbouring tiles share source lines (halo
// rows, 128-byte lines straddling a tile border); they should meet in ONE XCD's L2.
//   MDC_ORDER_BANDS     row-major runs of ceil(n/8) tiles per XCD
//   MDC_ORDER_ROWS      whole tile rows per XCD, as even as the row count allows (no horizontal
//                       neighbours split; XCDs with a row less idle at the end of a frame group)
//   MDC_ORDER_IDENTITY  block b = tile b: neighbours land on different XCDs (diagnosis: worst case)
//   MDC_ORDER_BLOCKS2D  the tile grid cut into 8 rectangles by recursive bisection of the longer side
//                       (least shared halo perimeter between XCDs; the rectangles differ in size by up to
//                       one row / column, XCDs with fewer tiles get padding slots)
static void bisect(int x0, int y0, int x1, int y1, int parts, int tx, std::vector<std::vector<int>>& out) {
  if (parts == 1) {
    std::vector<int> v;
    for (int y = y0; y < y1; y++)
      for (int x = x0; x < x1; x++) v.push_back(y * tx + x);
    out.push_back(v);
    return;
  }
  if (x1 - x0 > y1 - y0) {
    const int xm = x0 + (x1 - x0 + 1) / 2;
    bisect(x0, y0, xm, y1, parts / 2, tx, out);
    bisect(xm, y0, x1, y1, parts / 2, tx, out);
  } else {
    const int ym = y0 + (y1 - y0 + 1) / 2;
    bisect(x0, y0, x1, ym, parts / 2, tx, out);
    bisect(x0, ym, x1, y1, parts / 2, tx, out);
  }
}

std::vector<int> tile_order(int tx, int ty, int mode) {
  const int n = tx * ty;
  std::vector<std::vector<int>> per_xcd(8);
  if (mode == MDC_ORDER_BLOCKS2D && tx * ty >= 8) {
    per_xcd.clear();
    bisect(0, 0, tx, ty, 8, tx, per_xcd);
  } else if (mode == MDC_ORDER_IDENTITY) {
    for (int t = 0; t < n; t++) per_xcd[t % 8].push_back(t);
  } else if (mode == MDC_ORDER_ROWS && ty >= 8) {
    int r = 0;
    for (int k = 0; k < 8; k++) {
      const int rows = ty / 8 + (k < ty % 8 ? 1 : 0);
      for (int y = r; y < r + rows; y++)
        for (int x = 0; x < tx; x++) per_xcd[k].push_back(y * tx + x);
      r += rows;
    }
  } else {
    const int per = (n + 7) / 8;
    for (int t = 0; t < n; t++) per_xcd[t / per].push_back(t);
  }
  size_t slots = 0;
  for (const auto& v : per_xcd) slots = std::max(slots, v.size());
  std::vector<int> order(slots * 8, -1);
  for (int k = 0; k < 8; k++)
    for (size_t j = 0; j < per_xcd[k].size(); j++) order[j * 8 + k] = per_xcd[k][j];
  return order;
}

// Plan of the tiled kernel (see TilePlan): per tile the exact source window as a list of
// 16-byte chunks, per output the LDS offsets of its two tap rows.  Fails (tiled = false)
// when rows of the frame are not whole 16-byte chunks or a window is too large for LDS.
void free_src_plan(mdc_ctx::SrcPlan& pl) {
  for (void** p : {(void**)&pl.d_chunks, (void**)&pl.d_nch, (void**)&pl.d_taps, (void**)&pl.d_order})
    if (*p) {
      (void)hipFree(*p);
      *p = nullptr;
    }
  pl.tiled = false;
  pl.staged_bytes = 0;
  pl.n_tiles = pl.tiles_x = pl.n_blocks = 0;
}

void free_strip_plan(mdc_ctx::Strip& st) {
  for (void** p : {(void**)&st.d_chunks, (void**)&st.d_nch, (void**)&st.d_taps, (void**)&st.d_order})
    if (*p) {
      (void)hipFree(*p);
      *p = nullptr;
    }
  st.planned = false;
  st.staged_bytes = 0;
  st.n_blocks = st.n_tiles = st.tiles_x = 0;
}

void free_plan(mdc_ctx* c) {
  for (auto& pl : c->plan) {
    free_src_plan(pl);
  }
  free_strip_plan(c->strip);
}

template <typename T>
int upload(mdc_ctx* c, T** dst, const std::vector<T>& v) {
  MDC_HIP(c, hipMalloc(dst, std::max<size_t>(v.size(), 1) * sizeof(T)));
  if (!v.empty()) MDC_HIP(c, hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return MDC_OK;
}

// The plan for source pixels of `es` bytes (1 = raw u8 frames with the LUT replicas in LDS,
// 4 = float frames, no LUT): a 16-byte chunk holds 16 / es pixels.  Leaves pl.tiled = false when
// frame rows are not whole chunks or a window is too large.
int plan_source(mdc_ctx* c, int es, int kTileW, int kTileH, mdc_ctx::SrcPlan& pl) {
  const int ow = c->out_w, oh = c->out_h, iw = c->rm_in_w;
  const int kTileThreads = tile_threads(kTileW, kTileH);
  pl.staged_bytes = 0;
  const int tx = (ow + kTileW - 1) / kTileW, ty = (oh + kTileH - 1) / kTileH;
  const int n_tiles = tx * ty;
  const int ppc = 16 / es;  // pixels per chunk
  const bool lut = es == 1;
  // whole 16-byte chunks per frame row; one frame within the 32-bit lane offsets of the buffer descriptors
  const char* why = "frame rows are not whole chunks / frame too large";
  bool ok = (iw % ppc == 0) && (int64_t)iw * c->rm_in_h * es < (int64_t)kOutside && (int64_t)ow * oh * 4 < (int64_t)kOutside;
  if (tile_rpt(kTileW, kTileH) != 4 && es != 1) ok = false;  // the 8-rows-per-thread tiles exist for raw u8 frames only
  // the 960-/1024-thread tiles derive the output offsets of rows 1..3 from row 0 (kOutsideLean, mdc_kernels.hip)
  if ((kTileThreads >= 960 || tile_rpt(kTileW, kTileH) > 4) && (int64_t)ow * (oh + kTileH) * 4 >= 0xc0000000ll) ok = false;
  std::vector<std::vector<uint32_t>> chunks(n_tiles);
  std::vector<int> nch(n_tiles, 0);
  std::vector<uint32_t> taps((size_t)ow * oh, 0u);
  for (int t = 0; t < n_tiles && ok; t++) {
    const int bx = (t % tx) * kTileW, by = (t / tx) * kTileH;
    const int x1 = std::min(bx + kTileW, ow), y1 = std::min(by + kTileH, oh);
    int y_lo = std::numeric_limits<int>::max(), y_hi = -1;
    for (int y = by; y < y1; y++)
      for (int x = bx; x < x1; x++) {
        const float xx = c->h_rx[(size_t)y * ow + x], yy = c->h_ry[(size_t)y * ow + x];
        if (xx < 0) continue;
        y_lo = std::min(y_lo, (int)yy);
        y_hi = std::max(y_hi, (int)yy + 1);
      }
    if (y_hi < 0) continue;  // every output black: no window
    // Exact chunk SET per source row (not one run from the leftmost to the rightmost tap: the source footprint of a wide,
    // flat tile is a bowed band that touches a row in two separate places).  pos[row][chunk] = index of the chunk in the
    // tile's list, -1 = not staged.  A tap pair (xi, xi+1) marks both bytes' chunks, so chunks that are neighbours in a
    // frame row and both used are neighbours in the list too: the pair stays contiguous in LDS.
    const int cpr = iw / ppc;  // chunks per frame row
    const int nrows = y_hi - y_lo + 1;
    std::vector<int> pos((size_t)nrows * cpr, -1);
    for (int y = by; y < y1; y++)
      for (int x = bx; x < x1; x++) {
        const float xx = c->h_rx[(size_t)y * ow + x], yy = c->h_ry[(size_t)y * ow + x];
        if (xx < 0) continue;
        const int xi = (int)xx, yi = (int)yy;
        for (int dy = 0; dy < 2; dy++) {
          int* pr = &pos[(size_t)(yi + dy - y_lo) * cpr];
          pr[xi / ppc] = 0;
          pr[(xi + 1) / ppc] = 0;
        }
      }
    for (int k = 0; k < nrows; k++)
      for (int ch = 0; ch < cpr; ch++) {
        int& q = pos[(size_t)k * cpr + ch];
        if (q < 0) continue;
        q = (int)chunks[t].size();
        chunks[t].push_back((uint32_t)(((y_lo + k) * iw + ch * ppc) * es));
      }
    nch[t] = (int)chunks[t].size();
    if (nch[t] > (lut ? kTileMaxChunks : kTileMaxChunksF32) * kTileThreads || nch[t] * 16 > 65535) {
      ok = false;
      why = "a window has too many chunks";
    }
    pl.staged_bytes += (int64_t)nch[t] * 16;
    for (int y = by; y < y1 && ok; y++)
      for (int x = bx; x < x1; x++) {
        const float xx = c->h_rx[(size_t)y * ow + x], yy = c->h_ry[(size_t)y * ow + x];
        if (xx < 0) continue;
        const int xi = (int)xx, yi = (int)yy;
        const int p0 = pos[(size_t)(yi - y_lo) * cpr + xi / ppc], p1 = pos[(size_t)(yi + 1 - y_lo) * cpr + xi / ppc];
        taps[(size_t)y * ow + x] = (uint32_t)(p0 * 16 + (xi % ppc) * es) | ((uint32_t)(p1 * 16 + (xi % ppc) * es) << 16);
      }
  }
  // every tile's chunk list is padded (kOutside) to the kernel's maximum of staging rounds: the kernel loads
  // all of them unconditionally, first thing, before it knows the tile's chunk count
  const int cap = (lut ? kTileMaxChunks : kTileMaxChunksF32) * kTileThreads;
  int nch_max = 1;
  for (int t = 0; t < n_tiles; t++) nch_max = std::max(nch_max, nch[t]);
  const int win_bytes = (nch_max * 16 + 1023) & ~1023;  // a wave's DMA destination is 1 KiB aligned
  // Window buffers: as many frames staged ahead as LDS allows WITHOUT lowering the number of
  // workgroups per CU that two buffers permit (occupancy first, then depth), at most 4.
  const int wg_per_cu =
      std::max<int>(1, std::min<size_t>(kLdsPerCU / tiled_lds_bytes(win_bytes, 2, lut), 2048 / kTileThreads));
  // (the 960- / 1024-thread tiles stay at two buffers unless asked otherwise: their two-buffer instantiation keeps the bilinear weights
  // in registers -- exactly 64 VGPRs --, the three-buffer one has to redo them per frame (LEAN): 3.8 % slower, profiles/r04_experiments/06_*)
  const int nbuf_max = kTileThreads > 512 ? 3 : 4;
  int nbuf = 2;
  while (kTileThreads <= 512 && nbuf < nbuf_max && tiled_lds_bytes(win_bytes, nbuf + 1, lut) * wg_per_cu <= kLdsPerCU) nbuf++;
  if (c->opt_nbuf >= 2) nbuf = std::min(c->opt_nbuf, nbuf_max);
  if (tile_rpt(kTileW, kTileH) > 4) nbuf = 3;  // the only instantiation of the 8-rows-per-thread tiles (mdc_kernels.hip: launch_tiled_buf)
  if (tiled_lds_bytes(win_bytes, nbuf, lut) > kLdsPerCU) {
    ok = false;
    why = "windows do not fit LDS";
  }
  if (!ok) {
    if (getenv("MDC_DEBUG_PLAN")) fprintf(stderr, "mdc plan %dx%d (element size %d): not plannable: %s\n", kTileW, kTileH, es, why);
    return MDC_OK;
  }
  std::vector<uint32_t> flat((size_t)n_tiles * cap, kOutside);
  for (int t = 0; t < n_tiles; t++) std::copy(chunks[t].begin(), chunks[t].end(), flat.begin() + (size_t)t * cap);
  int rc;
  if ((rc = upload(c, &pl.d_chunks, flat)) != MDC_OK || (rc = upload(c, &pl.d_nch, nch)) != MDC_OK ||
      (rc = upload(c, &pl.d_taps, taps)) != MDC_OK)
    return rc;
  const std::vector<int> order = tile_order(tx, ty, c->opt_order);
  if ((rc = upload(c, &pl.d_order, order)) != MDC_OK) return rc;
  pl.n_blocks = (int)order.size();
  pl.n_tiles = n_tiles;
  pl.tiles_x = tx;
  pl.tile_w = kTileW;
  pl.tile_h = kTileH;
  pl.chunk_cap = cap;
  pl.win_bytes = win_bytes;
  pl.nbuf = nbuf;
  pl.tiled = true;
  return MDC_OK;
}

// Plan of the wave-private strip kernel (StripPlan): per 128 x 8 output tile the exact source window as a dense list of
// 16-byte chunks (<= kStripChunkCap), per output the byte offsets of its two tap rows inside the wave's FLOAT window.
// Planned when the remap stages fewer source pixels than it has outputs (config 5's scale-1 rectification, magnifying
// remaps) or on request (MDC_OPT_TWO_STAGE = 1); leaves st.planned = false when a window is too large, frame rows are not
// whole chunks, or the output height is not a multiple of 8 (rows are addressed through the store's scalar offset,
// which the hardware's range check does not cover).
int plan_strip(mdc_ctx* c) {
  mdc_ctx::Strip& st = c->strip;
  free_strip_plan(st);
  if (c->opt_two_stage == 2) return MDC_OK;
  const int ow = c->out_w, oh = c->out_h, iw = c->rm_in_w;
  constexpr int TW = kStripTileW, TH = kStripTileH;
  if (iw % 16 != 0 || oh % TH != 0 || (int64_t)iw * c->rm_in_h >= (int64_t)kOutside || (int64_t)ow * (oh + TH) * 4 >= 0xc0000000ll) return MDC_OK;
  const int tx = (ow + TW - 1) / TW, ty = oh / TH, n_tiles = tx * ty;
  std::vector<uint32_t> flat((size_t)n_tiles * kStripChunkCap, kOutside);
  std::vector<int> nch(n_tiles, 0);
  std::vector<uint32_t> taps((size_t)ow * oh, 0u);
  struct Row {
    int lo = std::numeric_limits<int>::max(), hi = -1, x0 = 0, lds = 0;
  };
  int nch_max = 1;
  int64_t staged = 0;
  for (int t = 0; t < n_tiles; t++) {
    const int bx = (t % tx) * TW, by = (t / tx) * TH;
    const int x1 = std::min(bx + TW, ow), y1 = by + TH;
    int y_lo = std::numeric_limits<int>::max(), y_hi = -1;
    for (int y = by; y < y1; y++)
      for (int x = bx; x < x1; x++) {
        const float xx = c->h_rx[(size_t)y * ow + x], yy = c->h_ry[(size_t)y * ow + x];
        if (xx < 0) continue;
        y_lo = std::min(y_lo, (int)yy);
        y_hi = std::max(y_hi, (int)yy + 1);
      }
    if (y_hi < 0) continue;  // every output black
    std::vector<Row> rows(y_hi - y_lo + 1);
    for (int y = by; y < y1; y++)
      for (int x = bx; x < x1; x++) {
        const float xx = c->h_rx[(size_t)y * ow + x], yy = c->h_ry[(size_t)y * ow + x];
        if (xx < 0) continue;
        const int xi = (int)xx, yi = (int)yy;
        for (int dy = 0; dy < 2; dy++) {
          Row& r = rows[yi + dy - y_lo];
          r.lo = std::min(r.lo, xi);
          r.hi = std::max(r.hi, xi + 1);
        }
      }
    int n = 0;
    for (size_t k = 0; k < rows.size(); k++) {
      Row& r = rows[k];
      if (r.hi < 0) continue;
      r.x0 = r.lo - r.lo % 16;
      r.lds = n * 16;
      const int cnt = (r.hi - r.x0) / 16 + 1;
      if (r.x0 + cnt * 16 > iw || n + cnt > kStripChunkCap) return MDC_OK;  // not plannable: the workgroup kernels keep the job
      for (int j = 0; j < cnt; j++) flat[(size_t)t * kStripChunkCap + n + j] = (uint32_t)((y_lo + (int)k) * iw + r.x0 + j * 16);
      n += cnt;
    }
    nch[t] = n;
    nch_max = std::max(nch_max, n);
    staged += (int64_t)n * 16;
    for (int y = by; y < y1; y++)
      for (int x = bx; x < x1; x++) {
        const float xx = c->h_rx[(size_t)y * ow + x], yy = c->h_ry[(size_t)y * ow + x];
        if (xx < 0) continue;
        const int xi = (int)xx, yi = (int)yy;
        const Row &r0 = rows[yi - y_lo], &r1 = rows[yi + 1 - y_lo];
        taps[(size_t)y * ow + x] = (uint32_t)(4 * (r0.lds + xi - r0.x0)) | ((uint32_t)(4 * (r1.lds + xi - r1.x0)) << 16);
      }
  }
  const double src_per_out = (double)staged / std::max<double>(1.0, (double)ow * oh);
  if (c->opt_two_stage != 1 && src_per_out >= 1.0) return MDC_OK;
  const int win = (nch_max * 16 + 63) & ~63;
  const int need = (4 * nch_max + 63) / 64;  // convert passes
  const int passes = need <= 2 ? 2 : need <= 3 ? 3 : need <= 4 ? 4 : need <= 5 ? 5 : 8;
  const int nbuf = c->opt_nbuf >= 1 && c->opt_nbuf <= 4 ? c->opt_nbuf : 2;
  if (strip_lds_bytes(win, nbuf, kStripWaves) > kLdsPerCU) return MDC_OK;
  int rc;
  if ((rc = upload(c, &st.d_chunks, flat)) != MDC_OK || (rc = upload(c, &st.d_nch, nch)) != MDC_OK || (rc = upload(c, &st.d_taps, taps)) != MDC_OK)
    return rc;
  // groups of kStripWaves consecutive tiles (row-major: neighbours along a tile row); XCD placement as for the workgroup tiles
  const int n_groups = (n_tiles + kStripWaves - 1) / kStripWaves;
  const int gx = std::max(1, tx / kStripWaves);
  const std::vector<int> order = (tx % kStripWaves == 0) ? tile_order(gx, n_groups / gx, c->opt_order) : tile_order(n_groups, 1, MDC_ORDER_BANDS);
  if ((rc = upload(c, &st.d_order, order)) != MDC_OK) return rc;
  st.n_blocks = (int)order.size();
  st.n_tiles = n_tiles;
  st.tiles_x = tx;
  st.win_bytes = win;
  st.passes = passes;
  st.nbuf = nbuf;
  st.staged_bytes = staged;
  st.planned = true;
  return MDC_OK;
}

// Plans of the tiled kernels for the current remap: tile grid, XCD placement, source bounding
// box, one SrcPlan per source pixel type.
int plan_tiles(mdc_ctx* c) {
  c->n_black = 0;
  c->bbox[0] = c->bbox[1] = std::numeric_limits<int>::max();
  c->bbox[2] = c->bbox[3] = -1;
  free_plan(c);
  const int ow = c->out_w, oh = c->out_h;
  for (size_t i = 0; i < (size_t)ow * oh; i++) {
    const float xx = c->h_rx[i], yy = c->h_ry[i];
    if (xx < 0) {
      c->n_black++;
      continue;
    }
    c->bbox[0] = std::min(c->bbox[0], (int)xx);
    c->bbox[2] = std::max(c->bbox[2], (int)xx + 1);
    c->bbox[1] = std::min(c->bbox[1], (int)yy);
    c->bbox[3] = std::max(c->bbox[3], (int)yy + 1);
  }
  if (c->bbox[2] < 0) c->bbox[0] = c->bbox[1] = 0;
  // Tile shape per source type: the requested one, or the first candidate whose windows fit (strongly
  // distorting cameras need the taller tiles: their windows are too wide for the staging rounds of the
  // smaller workgroups).  Both lists are in order of measured speed on the bench camera (tools/sweep.py,
  // tools/rate_undistort_f32.py).
  // (128 x 16 first: measured 5-7 % faster than 64 x 32 on the bench camera -- a 64-wide tile spans ~86 source
  // bytes, less than one 128-byte line, so nearly every line is fetched by two workgroups; at 128 columns far
  // fewer are.  profiles/r02_experiments/)
  static const TileShape cand_u8[] = {{128, 16}, {64, 32}, {128, 32}, {64, 64}, {64, 60}, {64, 16}};
  static const TileShape cand_f32[] = {{128, 16}, {64, 32}, {64, 16}, {128, 32}, {64, 64}, {64, 60}};  // 0.66 / 0.63 / 0.60 / 0.60 / 0.54 of 8 TB/s
  for (int which = 0; which < 2; which++) {
    const TileShape* cand = which == 0 ? cand_u8 : cand_f32;
    const bool forced = c->opt_tile_h != 0 || c->opt_tile_w != 0;
    for (int k = 0; k < 6; k++) {
      const int tw = c->opt_tile_w ? c->opt_tile_w : cand[k].w, th = c->opt_tile_h ? c->opt_tile_h : cand[k].h;
      if (forced && (tw != cand[k].w || th != cand[k].h)) continue;  // a forced dimension filters the list
      free_src_plan(c->plan[which]);
      const int rc = plan_source(c, which == 0 ? 1 : 4, tw, th, c->plan[which]);
      if (rc != MDC_OK) return rc;
      if (c->plan[which].tiled) break;
    }
  }
  return plan_strip(c);
}

}  // namespace mdc
