// TEST INFRASTRUCTURE ONLY -- stand-in for libzip's <zip.h>, enough for the calls the reference's
// src/BenchmarkDatasetReader.h makes on images.zip (:107-125, :256-258): open, count, names, open entry, read.
// Implemented in oracle/shim_zip.cpp with zlib -- a reader written independently of the product's
// csrc/host/zip_reader.cpp, so that "the reference's unmodified reader on an archive" can serve as the oracle for the
// product's reader on the same archive (tests/test_reader.py).  No ZIP64, no encryption.
#pragma once
#include <cstddef>
typedef struct zip zip_t;
typedef struct zip_file zip_file_t;
#define ZIP_RDONLY 16
#define ZIP_FL_ENC_STRICT 128u
zip_t* zip_open(const char* path, int flags, int* err);
long long zip_get_num_entries(zip_t* a, unsigned flags);
const char* zip_get_name(zip_t* a, unsigned long long index, unsigned flags);
zip_file_t* zip_fopen(zip_t* a, const char* name, unsigned flags);
long long zip_fread(zip_file_t* f, void* buf, unsigned long long n);
int zip_close(zip_t* a);
