// Threading contract of the C ABI (SURVEY.md section 8b "Threading"; the reference's unMapImage / undistort are
// re-entrant on shared objects, reference src/FOVUndistorter.cpp:322-368 is const) proved WITHOUT torch:
//
//   thread_soak <calibration folder> <threads> <iterations per thread> [<frames per batch> [<prefetch chunk>]]
//
// <prefetch chunk> > 0 (a calibration the strip kernel takes, batches of >= 2 chunks): every rectifying launch walks its
// batch in prefetched chunks alternating between the caller's stream and a stream borrowed from the context's slot pool
// (MDC_OPT_PREFETCH_CHUNK / MDC_OPT_PREFETCH_STREAMS) -- from T threads at once, next to host calls that lease slots too.
//
// T host threads share ONE mdc_ctx.  Every thread owns a hipStreamNonBlocking stream, its own frames and its own
// device buffers and loops: hipMemsetAsync poison on ITS stream -> mdc_process_batch_device on that stream (flag
// combination rotating per iteration) -> hipMemcpyAsync back on that stream -> hipStreamSynchronize -> compare with the
// CPU oracle (oracle/liboracle.so: test infrastructure, linked by this test only), bit for bit, NaN as a mask.
// Every ordering the result depends on is a same-stream ordering, so a mismatch can only be a product race.
// Every 16th iteration the thread also goes through the blocking host entry points (mdc_process_host, mdc_unmap_host,
// mdc_undistort_host_u8) on the shared context.  Prints one line "THREAD_SOAK ... mismatches M"; exit code 0 iff M == 0.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "mdc_host.h"

extern "C" {
// oracle/mdc_oracle.c
void orc_get_image(const unsigned char* raw, float* out, float* tmp, int in_w, int in_h, int out_w, int out_h, const float* ginv,
                   const float* vinv, int valid_gamma, int valid_vignette, const float* remap_x, const float* remap_y, int have_remap,
                   int rectify, int g, int v, int o);
void orc_unmap(const unsigned char* in, float* out, int n, const float ginv[256], const float* vinv, int valid_gamma,
               int valid_vignette, int g, int v, int o);
void orc_undistort_u8(const unsigned char* input, float* output, const float* remap_x, const float* remap_y, int in_w, int n_out);
void orc_synth_frames(unsigned char* out, long long first_frame, long long nframes, int npix, unsigned seed);
}

static bool same_bits(const float* a, const float* b, size_t n) {
  for (size_t i = 0; i < n; i++) {
    uint32_t x, y;
    std::memcpy(&x, a + i, 4);
    std::memcpy(&y, b + i, 4);
    if (x != y && !(std::isnan(a[i]) && std::isnan(b[i]))) return false;
  }
  return true;
}

int main(int argc, char** argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s <calibration folder> <threads> <iterations> [<frames per batch> [<prefetch chunk>]]\n", argv[0]);
    return 2;
  }
  std::string folder = argv[1];
  if (folder[folder.size() - 1] != '/') folder += "/";
  const int T = std::atoi(argv[2]), iters = std::atoi(argv[3]), B = argc > 4 ? std::atoi(argv[4]) : 3;
  const int prefetch_chunk = argc > 5 ? std::atoi(argv[5]) : 0;

  mdch_fov* fov = mdch_fov_create((folder + "camera.txt").c_str());
  int d4[4];
  mdch_fov_dims(fov, d4);
  const int W = d4[0], H = d4[1], w = d4[2], h = d4[3];
  mdch_photo* photo = mdch_photo_create((folder + "pcalib.txt").c_str(), (folder + "vignette.png").c_str(), W, H);
  if (!mdch_fov_valid(fov) || mdch_photo_valid(photo) != 3) {
    std::fprintf(stderr, "calibration invalid\n");
    return 3;
  }
  const size_t npi = (size_t)W * H, npo = (size_t)w * h;
  std::vector<float> rx(npo), ry(npo), ginv(256), vinv(npi);
  mdch_fov_remap(fov, rx.data(), ry.data());
  mdch_photo_ginv(photo, ginv.data());
  mdch_photo_vignette(photo, 0, vinv.data());

  mdc_ctx* ctx = 0;
  if (mdc_create(0, &ctx) != MDC_OK) {
    std::fprintf(stderr, "mdc_create: %s\n", mdc_last_error(0));
    return 4;
  }
  if (mdch_bind(ctx, fov, photo) != MDC_OK) {
    std::fprintf(stderr, "bind: %s\n", mdc_last_error(ctx));
    return 5;
  }
  if (prefetch_chunk > 0) {
    mdc_info info;
    if (mdc_set_option(ctx, MDC_OPT_PREFETCH_CHUNK, prefetch_chunk) != MDC_OK || mdc_get_info(ctx, &info) != MDC_OK || !info.two_stage ||
        info.prefetch_streams != 2) {
      std::fprintf(stderr, "prefetch chunk %d: not on the strip path with two streams (%s)\n", prefetch_chunk, mdc_last_error(ctx));
      return 6;
    }
  }

  // frames and expected results, up front on one thread: thread k owns frames k*B .. k*B+B-1 of the counter-hash sequence;
  // flag words rotate over the 8 rectifying + 8 non-rectifying combinations
  std::vector<std::vector<unsigned char>> raw((size_t)T);
  std::vector<std::vector<std::vector<float>>> want((size_t)T);  // [thread][flags][B * npix]
  std::vector<float> tmp(npi);
  for (int k = 0; k < T; k++) {
    raw[(size_t)k].resize((size_t)B * npi);
    orc_synth_frames(raw[(size_t)k].data(), (long long)k * B, B, (int)npi, 777u);
    for (int j = 0; j < 50; j++) raw[(size_t)k][(size_t)((j * 7919 + k * 104729) % (int)((size_t)B * npi))] = 255;
    want[(size_t)k].resize(16);
    for (unsigned fl = 0; fl < 16; fl++) {
      const size_t no = (fl & MDC_RECTIFY) ? npo : npi;
      want[(size_t)k][fl].resize((size_t)B * no);
      for (int f = 0; f < B; f++)
        orc_get_image(raw[(size_t)k].data() + (size_t)f * npi, want[(size_t)k][fl].data() + (size_t)f * no, tmp.data(), W, H, w, h,
                      ginv.data(), vinv.data(), 1, 1, rx.data(), ry.data(), 1, (fl & MDC_RECTIFY) ? 1 : 0, (fl & MDC_GAMMA) ? 1 : 0,
                      (fl & MDC_VIGNETTE) ? 1 : 0, (fl & MDC_KILL_OVEREXPOSED) ? 1 : 0);
    }
  }

  std::atomic<long long> mismatches(0), launches(0), host_calls(0);
  std::atomic<int> failed(0);
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (int k = 0; k < T; k++)
    th.emplace_back([&, k]() {
      hipSetDevice(0);
      hipStream_t s;
      uint8_t* d_in = 0;
      float *d_out = 0, *h_out = 0;
      const size_t nmax = (size_t)B * (npi > npo ? npi : npo);
      if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess || hipMalloc((void**)&d_in, (size_t)B * npi) != hipSuccess ||
          hipMalloc((void**)&d_out, nmax * 4) != hipSuccess || hipHostMalloc((void**)&h_out, nmax * 4, 0) != hipSuccess ||
          hipMemcpyAsync(d_in, raw[(size_t)k].data(), (size_t)B * npi, hipMemcpyHostToDevice, s) != hipSuccess) {
        failed = 1;
        return;
      }
      std::vector<float> ho(npi > npo ? npi : npo), ht(npi);
      // page-locked twins of the host buffers: every other host leg goes through the zero-copy path (the kernels read the frame
      // and write the result over PCIe themselves, from this thread's leased stream, next to the other threads' launches)
      unsigned char* pin_raw = 0;
      float* pin_out = 0;
      if (hipHostMalloc((void**)&pin_raw, npi, 0) != hipSuccess || hipHostMalloc((void**)&pin_out, (npi > npo ? npi : npo) * 4, 0) != hipSuccess) {
        failed = 1;
        return;
      }
      std::memcpy(pin_raw, raw[(size_t)k].data(), npi);
      for (int it = 0; it < iters; it++) {
        const unsigned fl = (unsigned)((k * 5 + it) % 16);
        const size_t no = (fl & MDC_RECTIFY) ? npo : npi;
        // poison on THIS stream: ordered before the launch, which is ordered before the copy back
        if (hipMemsetAsync(d_out, 0xA5, (size_t)B * no * 4, s) != hipSuccess) failed = 1;
        if (mdc_process_batch_device(ctx, d_in, d_out, B, fl, s) != MDC_OK) failed = 1;
        if (hipMemcpyAsync(h_out, d_out, (size_t)B * no * 4, hipMemcpyDeviceToHost, s) != hipSuccess) failed = 1;
        if (hipStreamSynchronize(s) != hipSuccess) failed = 1;
        launches++;
        if (!same_bits(h_out, want[(size_t)k][fl].data(), (size_t)B * no)) {
          mismatches++;
          std::fprintf(stderr, "MISMATCH process_batch_device thread %d iteration %d flags %u\n", k, it, fl);
        }
        if (it % 16 == 0) {  // the blocking host entry points on the shared context
          const unsigned char* r0 = raw[(size_t)k].data();
          if (it % 32 == 0) {  // zero copy: page-locked frame -> page-locked result
            for (size_t q = 0; q < no; q++) pin_out[q] = -7.f;
            if (mdc_process_host(ctx, pin_raw, pin_out, fl) != MDC_OK) failed = 1;
            if (!same_bits(pin_out, want[(size_t)k][fl].data(), no)) {
              mismatches++;
              std::fprintf(stderr, "MISMATCH process_host (zero copy) thread %d iteration %d flags %u\n", k, it, fl);
            }
            host_calls++;
          }
          std::fill(ho.begin(), ho.end(), -7.f);
          if (mdc_process_host(ctx, r0, ho.data(), fl) != MDC_OK) failed = 1;
          if (!same_bits(ho.data(), want[(size_t)k][fl].data(), no)) {
            mismatches++;
            std::fprintf(stderr, "MISMATCH process_host thread %d iteration %d flags %u\n", k, it, fl);
          }
          std::fill(ht.begin(), ht.end(), -7.f);
          if (mdc_unmap_host(ctx, r0, ht.data(), (int)npi, fl & 7u) != MDC_OK) failed = 1;
          if (!same_bits(ht.data(), want[(size_t)k][fl & 7u].data(), npi)) {
            mismatches++;
            std::fprintf(stderr, "MISMATCH unmap_host thread %d iteration %d flags %u\n", k, it, fl & 7u);
          }
          std::fill(ho.begin(), ho.end(), -7.f);
          if (mdc_undistort_host_u8(ctx, r0, ho.data(), (int)npi, (int)npo) != MDC_OK) failed = 1;
          if (!same_bits(ho.data(), want[(size_t)k][MDC_RECTIFY].data(), npo)) {
            mismatches++;
            std::fprintf(stderr, "MISMATCH undistort_host_u8 thread %d iteration %d\n", k, it);
          }
          host_calls += 3;
        }
      }
      hipStreamSynchronize(s);
      hipFree(d_in);
      hipFree(d_out);
      hipHostFree(h_out);
      hipHostFree(pin_raw);
      hipHostFree(pin_out);
      hipStreamDestroy(s);
    });
  for (auto& t : th) t.join();
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (failed) std::fprintf(stderr, "a HIP / mdc call failed: %s\n", mdc_last_error(ctx));
  std::printf("THREAD_SOAK threads %d iterations %d frames_per_batch %d prefetch_chunk %d size %dx%d->%dx%d device_launches %lld host_calls %lld seconds %.2f "
              "call_failures %d mismatches %lld\n",
              T, iters, B, prefetch_chunk, W, H, w, h, (long long)launches, (long long)host_calls, sec, (int)failed, (long long)mismatches);
  mdc_destroy(ctx);
  mdch_photo_destroy(photo);
  mdch_fov_destroy(fov);
  return (mismatches == 0 && !failed) ? 0 : 1;
}
