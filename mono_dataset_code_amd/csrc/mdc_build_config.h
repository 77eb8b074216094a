// Build-time switches of libmdc_hip, in one place.
//
// Three kinds:
//   tuning     the shipped value is the measured best; other values give the SAME results (bit for bit) at another speed
//   debug      MDC_DEBUG_BOUNDS: every tap / staging chunk / gather index is checked in the kernel, a violation traps
//   diagnosis  WRONG RESULTS BY DESIGN (a stream compiled out, arithmetic replaced): for taking a kernel apart only
//
// A diagnosis switch compiles only under MDC_DIAGNOSIS_BUILD, which mono_dataset_code_amd/build.py:build_variant sets and
// which puts the library under mono_dataset_code_amd/variants/ -- never next to the product.  mdc_build_flags() (include/mdc_hip.h)
// returns every switch that is not at its shipped value; the product library must return "" (tests/test_abi.py), and
// bench.py prints it in its line.
#pragma once
#include <string>

// ---- tuning ------------------------------------------------------------------------------------------------------
#ifndef MDC_EXP_LUT_REP
#define MDC_EXP_LUT_REP 32  // LDS replicas of the 256-entry response LUT (32 = one per bank, conflict-free)
#endif
#ifndef MDC_EXP_STRIP_WAVES
#define MDC_EXP_STRIP_WAVES 4  // strip kernel: waves (tiles) per workgroup, sharing the LUT replicas
#endif
#ifndef MDC_EXP_LOAD_NT
#define MDC_EXP_LOAD_NT 0   // staging loads: plain (L2-allocating) -- neighbouring tiles re-use halo lines; nt measured slower
#endif
#ifndef MDC_EXP_STORE_NT
#define MDC_EXP_STORE_NT 1  // output stores carry the nontemporal hint
#endif
#ifndef MDC_EXP_STRIP_LUT_REP
#define MDC_EXP_STRIP_LUT_REP 8  // LUT replicas of the strip kernel (8 KiB): its LUT reads are per SOURCE pixel, a few bank conflicts cost little
#endif
#ifndef MDC_EXP_STRIP_WAVES_PER_EU
#define MDC_EXP_STRIP_WAVES_PER_EU 5  // register budget of the strip kernel: 512 / 5 -> 100 VGPRs
#endif
#ifndef MDC_EXP_GUESS_DIV
#define MDC_EXP_GUESS_DIV 4  // device Huffman decoder: the first guess decodes the last 1/4 of the left neighbour's subsequence
#endif
#ifndef MDC_EXP_GRAD_ROWS
#define MDC_EXP_GRAD_ROWS 8  // gradient kernel: rows per wave (its two halo rows are read again by the neighbouring bands)
#endif
#ifndef MDC_EXP_GUESS_MIN_BITS
#define MDC_EXP_GUESS_MIN_BITS 512  // ... but at least this many bits (high qualities: ~250 bits per block, 512 bits are two blocks)
#endif
#ifndef MDC_EXP_HUFF_MAX_SEGMENTS
#define MDC_EXP_HUFF_MAX_SEGMENTS 8  // device Huffman decoder, small batches: workgroups per frame at most (8 up to 32 frames, 4 up to 64, 2 up to 128)
#endif
#ifndef MDC_EXP_HUFF_PROVISIONAL
#define MDC_EXP_HUFF_PROVISIONAL 1  // ... segments hand on a provisional exit state first (0: the final one only, one relaxation after the other down the chain)
#endif
// MDC_EXP_STORE_AUX (undefined = follow MDC_EXP_STORE_NT): raw cache-policy bits of the output stores

// ---- debug -------------------------------------------------------------------------------------------------------
#ifndef MDC_DEBUG_BOUNDS
#define MDC_DEBUG_BOUNDS 0
#endif

// ---- diagnosis (wrong results) -------------------------------------------------------------------------------------
#ifndef MDC_EXP_SKIP_STORE
#define MDC_EXP_SKIP_STORE 0  // outputs are computed but (practically) never stored -> read side alone
#endif
#ifndef MDC_EXP_SKIP_LOAD
#define MDC_EXP_SKIP_LOAD 0   // every frame re-stages frame 0 (L2 hits) -> write side alone
#endif
#ifndef MDC_EXP_FAKE_COMPUTE
#define MDC_EXP_FAKE_COMPUTE 0  // 1 = one tap + one LUT read per output instead of 4 + 4, 2 = no LDS reads
#endif
#ifndef MDC_EXP_STRIP_NOCONVERT
#define MDC_EXP_STRIP_NOCONVERT 0  // the strip kernel skips its convert phase
#endif
#ifndef MDC_EXP_STRIP_NOSAMPLE
#define MDC_EXP_STRIP_NOSAMPLE 0   // the strip kernel stores a register instead of sampling
#endif
#ifndef MDC_EXP_STRIP_FAKE_GRAD
#define MDC_EXP_STRIP_FAKE_GRAD 0  // the strip kernel also writes level-0 gradient images -- the traffic, sample count and store shapes of a fused variant, not its values
#endif
#ifndef MDC_EXP_PAD_VALU
#define MDC_EXP_PAD_VALU 0    // N dummy VALU instructions per wave and frame in the tiled kernel's loop (right results): the cost of one instruction
#endif
#ifndef MDC_EXP_TIMING
#define MDC_EXP_TIMING 0      // some waves print the cycles their frame loop spent per phase (tools/phase_timing.sh): right results, device printf
#endif
// MDC_EXP_HUFF_ROUNDS (undefined): the Huffman kernel reports its relaxation rounds in the status word's upper bits
// MDC_EXP_HUFF_VERIFY (undefined): the split Huffman kernel re-decodes every subsequence before the write pass and counts disagreements into the status word
// MDC_EXP_HUFF_BAD_PROVISIONAL (undefined): fault injection -- the split Huffman kernel publishes wrong provisional states (right results, slower)
// MDC_EXP_GRAD_FAKE_READ (undefined): the gradient kernel reads the levels from a frame's first 16 KB (what does re-reading the levels cost the DSO path?)
// MDC_EXP_HUFF_FAKE_STREAM (undefined): the Huffman kernels' refills read one of 64 words (what do the divergent stream loads cost?)
// MDC_EXP_HUFF_NOSTORE (undefined): the Huffman kernels' write pass stores DC terms only (what do the scattered 2-byte stores cost?)

#if (MDC_EXP_SKIP_STORE || MDC_EXP_SKIP_LOAD || MDC_EXP_FAKE_COMPUTE || MDC_EXP_STRIP_FAKE_GRAD || MDC_EXP_STRIP_NOCONVERT || MDC_EXP_STRIP_NOSAMPLE || \
     MDC_EXP_TIMING || MDC_EXP_PAD_VALU || defined(MDC_EXP_HUFF_ROUNDS) || defined(MDC_EXP_HUFF_NOSTORE) || defined(MDC_EXP_GRAD_FAKE_READ) || defined(MDC_EXP_HUFF_FAKE_STREAM) || defined(MDC_EXP_HUFF_VERIFY) || defined(MDC_EXP_HUFF_BAD_PROVISIONAL)) && !defined(MDC_DIAGNOSIS_BUILD)
#error "a diagnosis switch (wrong results / device printf) is set: build through mono_dataset_code_amd/build.py:build_variant, which defines MDC_DIAGNOSIS_BUILD and writes to variants/"
#endif

// "NAME=value NAME=value ..." of everything that is not at its shipped value ("" for the product build)
#define MDC_CFG_STR2(x) #x
#define MDC_CFG_STR(x) MDC_CFG_STR2(x)
#define MDC_CFG_ITEM(name, shipped) ((name) != (shipped) ? " " #name "=" MDC_CFG_STR(name) : "")
namespace mdc {
inline const char* build_flags_string() {
  static const char* const parts[] = {
#ifdef MDC_DIAGNOSIS_BUILD
      " MDC_DIAGNOSIS_BUILD",
#endif
      MDC_CFG_ITEM(MDC_EXP_LUT_REP, 32),
      MDC_CFG_ITEM(MDC_EXP_STRIP_WAVES, 4),
      MDC_CFG_ITEM(MDC_EXP_LOAD_NT, 0),
      MDC_CFG_ITEM(MDC_EXP_STORE_NT, 1),
      MDC_CFG_ITEM(MDC_EXP_STRIP_LUT_REP, 8),
      MDC_CFG_ITEM(MDC_EXP_STRIP_WAVES_PER_EU, 5),
      MDC_CFG_ITEM(MDC_EXP_GUESS_DIV, 4),
      MDC_CFG_ITEM(MDC_EXP_GRAD_ROWS, 8),
      MDC_CFG_ITEM(MDC_EXP_GUESS_MIN_BITS, 512),
      MDC_CFG_ITEM(MDC_EXP_HUFF_MAX_SEGMENTS, 8),
      MDC_CFG_ITEM(MDC_EXP_HUFF_PROVISIONAL, 1),
#ifdef MDC_EXP_STORE_AUX
      " MDC_EXP_STORE_AUX=" MDC_CFG_STR(MDC_EXP_STORE_AUX),
#endif
      MDC_CFG_ITEM(MDC_DEBUG_BOUNDS, 0),
      MDC_CFG_ITEM(MDC_EXP_SKIP_STORE, 0),
      MDC_CFG_ITEM(MDC_EXP_SKIP_LOAD, 0),
      MDC_CFG_ITEM(MDC_EXP_FAKE_COMPUTE, 0),
      MDC_CFG_ITEM(MDC_EXP_STRIP_NOCONVERT, 0),
      MDC_CFG_ITEM(MDC_EXP_STRIP_NOSAMPLE, 0),
      MDC_CFG_ITEM(MDC_EXP_STRIP_FAKE_GRAD, 0),
      MDC_CFG_ITEM(MDC_EXP_TIMING, 0),
      MDC_CFG_ITEM(MDC_EXP_PAD_VALU, 0),
#ifdef MDC_EXP_HUFF_ROUNDS
      " MDC_EXP_HUFF_ROUNDS",
#endif
#ifdef MDC_EXP_HUFF_NOSTORE
      " MDC_EXP_HUFF_NOSTORE",
#endif
#ifdef MDC_EXP_GRAD_FAKE_READ
      " MDC_EXP_GRAD_FAKE_READ",
#endif
#ifdef MDC_EXP_HUFF_FAKE_STREAM
      " MDC_EXP_HUFF_FAKE_STREAM",
#endif
#ifdef MDC_EXP_HUFF_VERIFY
      " MDC_EXP_HUFF_VERIFY",
#endif
#ifdef MDC_EXP_HUFF_BAD_PROVISIONAL
      " MDC_EXP_HUFF_BAD_PROVISIONAL",
#endif
  };
  static const std::string joined = [] {  // thread-safe one-time initialisation
    std::string r;
    for (const char* p : parts) r += (r.empty() && *p == ' ') ? p + 1 : p;
    return r;
  }();
  return joined.c_str();
}
}  // namespace mdc
