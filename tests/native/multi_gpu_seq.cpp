// BASELINE.json configs[3] as a C++ program: ONE process, every visible MI355X (or the first <ndev>), a synthetic
// sequence sharded round-robin (frame f -> device f % N), calibration built once on the host by the drop-in classes
// and handed to the other devices by the RCCL broadcast of include/mdc_multi.h.
//
//   multi_gpu_seq <calibration folder> <frames in the sequence> <timed passes> [<dump dir> [<ndev>]]
//
// Checks on every run: all devices hold bit-identical tables after the broadcast.  With <dump dir>: the first two
// results of every rank are written there (rank<r>_out.bin, floats) for tests/test_native_multi.py to compare with
// the CPU oracle for the global frame indices r and r + N.
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "MdcBind.h"
#include "mdc_bench.h"
#include "mdc_multi.h"

#define CHECK(x)                                                                      \
  do {                                                                                \
    const int rc_ = (x);                                                              \
    if (rc_ != MDC_OK) {                                                              \
      std::fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, mdc_multi_last_error(m)); \
      return 10;                                                                      \
    }                                                                                 \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s <calibration folder> <frames> <passes> [<dump dir> [<ndev>]]\n", argv[0]);
    return 2;
  }
  std::string folder = argv[1];
  if (folder[folder.size() - 1] != '/') folder += "/";
  const long long total = std::atoll(argv[2]);
  const int passes = std::atoi(argv[3]);
  const char* dump = argc > 4 && argv[4][0] ? argv[4] : 0;
  const int want = argc > 5 ? std::atoi(argv[5]) : 0;

  mdc_multi* m = 0;
  if (mdc_multi_create(0, want, &m) != MDC_OK) {
    std::fprintf(stderr, "mdc_multi_create: %s\n", mdc_multi_last_error(0));
    return 3;
  }
  const int n = mdc_multi_size(m);

  // calibration: built on the host once (rank 0's tables), as DatasetReader's constructor does (reference
  // src/BenchmarkDatasetReader.h:135-136), then ONE broadcast
  UndistorterFOV fov((folder + "camera.txt").c_str());
  PhotometricUndistorter photo(folder + "pcalib.txt", folder + "vignette.png", fov.getInputDims()[0], fov.getInputDims()[1]);
  if (!fov.isValid()) return 4;
  if (mdc_bind_objects(mdc_multi_ctx(m, 0), &fov, &photo) != MDC_OK) {
    std::fprintf(stderr, "bind: %s\n", mdc_last_error(mdc_multi_ctx(m, 0)));
    return 5;
  }
  CHECK(mdc_multi_bcast_tables(m, 0));
  std::vector<unsigned char> ref_blob;
  for (int r = 0; r < n; r++) {
    size_t bytes = 0;
    mdc_export_tables(mdc_multi_ctx(m, r), 0, 0, &bytes);
    std::vector<unsigned char> b(bytes);
    if (mdc_export_tables(mdc_multi_ctx(m, r), b.data(), b.size(), &bytes) != MDC_OK) return 6;
    if (r == 0) ref_blob = b;
    else if (b != ref_blob) {
      std::fprintf(stderr, "tables of rank %d differ from rank 0 after the broadcast\n", r);
      return 7;
    }
  }

  const int W = fov.getInputDims()[0], H = fov.getInputDims()[1], w = fov.getOutputDims()[0], h = fov.getOutputDims()[1];
  const size_t npi = (size_t)W * H, npo = (size_t)w * h;
  std::vector<uint8_t*> d_in((size_t)n, (uint8_t*)0);
  std::vector<float*> d_out((size_t)n, (float*)0);
  // every rank's frame / result buffers come from the product's allocator (mdc_alloc_placed_device, include/mdc_hip.h): a pair the pass
  // runs fast on, found by timing the pass on this rank's device; the tables are in place (the probe needs them)
  const unsigned flags = MDC_GAMMA | MDC_VIGNETTE | MDC_KILL_OVEREXPOSED | MDC_RECTIFY;
  std::vector<mdc_placed_buffers> placed((size_t)n);
  for (int r = 0; r < n; r++) {
    const long long mine = mdc_multi_frames_of_rank(m, total, r);
    if (mine <= 0) continue;
    if (mdc_alloc_placed_device(mdc_multi_ctx(m, r), 0, 0, mine, flags, MDC_PLACE_AUTO, mdc_multi_stream(m, r), &placed[(size_t)r]) != MDC_OK) {
      std::fprintf(stderr, "mdc_alloc_placed_device failed on rank %d (%lld frames): %s\n", r, mine, mdc_last_error(mdc_multi_ctx(m, r)));
      return 8;
    }
    d_in[(size_t)r] = placed[(size_t)r].d_in;
    d_out[(size_t)r] = placed[(size_t)r].d_out;
    std::printf("PLACEMENT rank %d device %d probe_ms_first %.4f probe_ms_chosen %.4f : %s\n", r, mdc_multi_device(m, r), placed[(size_t)r].ms_first,
                placed[(size_t)r].ms_chosen, placed[(size_t)r].note);
  }
  // the synthetic sequence of SURVEY.md 8(d), sharded: local frame i of rank r is global frame r + i*N
  // (libmdc_bench.so, a test utility: the product libraries do not generate frames)
  for (int r = 0; r < n; r++) {
    const long long mine = mdc_multi_frames_of_rank(m, total, r);
    for (long long i = 0; i < mine; i++)
      if (mdcb_synth_frames_device(mdc_multi_device(m, r), d_in[(size_t)r] + (size_t)i * npi, r + i * n, 1, (int)npi, 12345u, mdc_multi_stream(m, r)) != 0) {
        std::fprintf(stderr, "frame synthesis failed on rank %d\n", r);
        return 8;
      }
  }
  CHECK(mdc_multi_synchronize(m));
  for (int r = 0; r < n; r++)
    if (mdc_multi_comm_count(m, r) != n) {
      std::fprintf(stderr, "rank %d: ncclCommCount = %d, expected %d\n", r, mdc_multi_comm_count(m, r), n);
      return 11;
    }
  for (int k = 0; k < 3; k++) CHECK(mdc_multi_process_sequence_device(m, d_in.data(), d_out.data(), total, flags));
  CHECK(mdc_multi_synchronize(m));
  const auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < passes; k++) CHECK(mdc_multi_process_sequence_device(m, d_in.data(), d_out.data(), total, flags));
  CHECK(mdc_multi_synchronize(m));
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::printf("MULTI_GPU_SEQ devices %d frames %lld passes %d seconds %.6f frames_per_s %.1f mpix_per_s %.1f tables_bit_equal 1 rccl_ranks %d\n", n,
              total, passes, dt, total * (double)passes / dt, total * (double)passes * npi / dt / 1e6, mdc_multi_comm_count(m, 0));

  if (dump)
    for (int r = 0; r < n; r++) {
      const size_t mine = std::min<size_t>(2, (size_t)mdc_multi_frames_of_rank(m, total, r));
      std::vector<float> host(mine * npo);
      hipSetDevice(mdc_multi_device(m, r));
      if (mine && hipMemcpy(host.data(), d_out[(size_t)r], host.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return 9;
      const std::string path = std::string(dump) + "/rank" + std::to_string(r) + "_out.bin";
      FILE* f = std::fopen(path.c_str(), "wb");
      if (!f) return 9;
      std::fwrite(host.data(), sizeof(float), host.size(), f);
      std::fclose(f);
    }
  for (int r = 0; r < n; r++) mdc_free_placed_device(mdc_multi_ctx(m, r), &placed[(size_t)r]);
  mdc_multi_destroy(m);
  return 0;
}
