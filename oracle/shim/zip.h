// TEST INFRASTRUCTURE ONLY -- stand-in for libzip's <zip.h>.
//
// The reference's src/BenchmarkDatasetReader.h includes "zip.h" unconditionally and
// falls back to images.zip only when the sequence folder has no images/ directory
// (:96-125).  The drop-in test always provides images/, so these entry points are
// never reached; they exist so the unmodified header compiles without libzip.
#pragma once
#include <cstddef>
typedef struct zip zip_t;
typedef struct zip_file zip_file_t;
#define ZIP_RDONLY 16
#define ZIP_FL_ENC_STRICT 128u
static inline zip_t* zip_open(const char*, int, int* err) { if (err) *err = 9; return 0; }
static inline long long zip_get_num_entries(zip_t*, unsigned) { return 0; }
static inline const char* zip_get_name(zip_t*, unsigned long long, unsigned) { return ""; }
static inline zip_file_t* zip_fopen(zip_t*, const char*, unsigned) { return 0; }
static inline long long zip_fread(zip_file_t*, void*, unsigned long long) { return -1; }
static inline int zip_close(zip_t*) { return 0; }
