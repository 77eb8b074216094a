// AddressSanitizer / UndefinedBehaviorSanitizer run of the host layer (no GPU needed): the calibration parsers and
// table builders, the frame decoders, the zip reader, the image pool and the reader's listing / prefetch pool are
// compiled INTO this program with -fsanitize=address,undefined (tests/test_sanitize.py builds and runs it) and fed
// valid, malformed and truncated inputs.  Any out-of-bounds access, use-after-free, signed overflow or misaligned
// access aborts the program.
//
//   host_sanitize <fixture dir>      fixtures are written by tests/test_sanitize.py
#include <dirent.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "BenchmarkDatasetReader.h"
#include "FOVUndistorter.h"
#include "PhotometricUndistorter.h"
#include "image_codecs.h"
#include "image_codecs_internal.h"
#include "mdc_hip.h"
#include "mdc_host.h"
#include "zip_reader.h"

static std::vector<std::string> list(const std::string& dir) {
  std::vector<std::string> out;
  if (DIR* dp = opendir(dir.c_str())) {
    while (struct dirent* e = readdir(dp))
      if (e->d_name[0] != '.') out.push_back(dir + "/" + e->d_name);
    closedir(dp);
  }
  return out;
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string root = argv[1];
  long decoded = 0, refused = 0, streams_built = 0;
  const size_t kHdr = sizeof(mdc_jpeg_stream_header);

  // 1. every file under images_any/ through the decoders, whole and truncated at many lengths
  for (const std::string& f : list(root + "/images_any")) {
    std::vector<unsigned char> bytes;
    if (!mdc_host::read_file(f, bytes)) return 3;
    std::vector<unsigned char> out(4u << 20);
    for (size_t cut : {bytes.size(), bytes.size() / 2, bytes.size() / 3, (size_t)100, (size_t)17, (size_t)4, (size_t)1, (size_t)0}) {
      if (cut > bytes.size()) continue;
      std::vector<unsigned char> part(bytes.begin(), bytes.begin() + (long)cut);  // exact-size heap block: overreads are caught
      int w = 0, h = 0;
      std::string err;
      if (mdc_host::decode_gray8(part.data(), part.size(), out.data(), out.size(), &w, &h, &err)) decoded++;
      else refused++;
      // a buffer that is too small must be refused, not overrun
      std::vector<unsigned char> tiny(16);
      mdc_host::decode_gray8(part.data(), part.size(), tiny.data(), tiny.size(), &w, &h, &err);
      // the stream builder of the device Huffman stage: exact-size stream buffers of several capacities (the copy of the
      // entropy-coded bytes must stop at the capacity), and the coefficient-record decoder
      for (size_t cap : {part.size() + kHdr + 64, kHdr + 40, kHdr + 16, kHdr - 56, (size_t)64}) {
        std::vector<uint32_t> stream((cap + 3) / 4);
        size_t used = 0;
        if (mdc_host::jpeg_stream(part.data(), part.size(), reinterpret_cast<unsigned char*>(stream.data()), cap, &used, &w, &h, &err)) {
          if (used > cap || used < kHdr + 17) return 9;
          streams_built++;
        }
      }
    }
    // single corrupted bytes inside the entropy-coded / compressed data
    for (size_t pos = bytes.size() / 2; pos < bytes.size() && pos < bytes.size() / 2 + 40; pos += 3) {
      std::vector<unsigned char> bad(bytes);
      bad[pos] ^= 0x5a;
      int w = 0, h = 0;
      std::string err;
      mdc_host::decode_gray8(bad.data(), bad.size(), out.data(), out.size(), &w, &h, &err);
      std::vector<uint32_t> stream((bad.size() + kHdr + 64) / 4);
      size_t used = 0;
      mdc_host::jpeg_stream(bad.data(), bad.size(), reinterpret_cast<unsigned char*>(stream.data()), stream.size() * 4, &used, &w, &h, &err);
    }
  }
  if (streams_built == 0) return 10;  // (the fixtures hold gray baseline JPEGs: some streams must come out)

  // 2. calibration parsers: every camera*.txt / pcalib*.txt under calib/ (valid, malformed, empty, truncated)
  for (const std::string& f : list(root + "/calib")) {
    if (f.find("camera") != std::string::npos) {
      UndistorterFOV u(f.c_str());
      if (u.isValid()) {
        std::vector<float> x(100), y(100);
        for (int i = 0; i < 100; i++) {
          x[(size_t)i] = (float)(i * 3 % 50);
          y[(size_t)i] = (float)(i * 7 % 40);
        }
        u.distortCoordinates(x.data(), y.data(), 100);
        (void)u.getK_rect();
        (void)u.getOriginalCalibration();
      }
    } else if (f.find("pcalib") != std::string::npos) {
      for (const std::string& v : list(root + "/vignettes")) {
        PhotometricUndistorter p(f, v, 48, 32);
        (void)p.getGInv();
      }
      PhotometricUndistorter q(f, "", 48, 32);
      PhotometricUndistorter r(f, root + "/does_not_exist.png", 48, 32);
    }
  }

  // 3. zip archives: valid (stored / deflated), truncated, garbage
  for (const std::string& f : list(root + "/zips")) {
    mdc_host::ZipArchive z;
    std::string err;
    if (!z.open(f, &err)) continue;
    for (int i = 0; i < z.entries(); i++) {
      std::vector<unsigned char> data;
      z.read(i, data, &err);
      (void)z.find(z.name(i));
    }
  }

  // 4. the reader without a GPU: listing, times.txt, decode pool with prefetch, random access, the pool-backed image
  for (const std::string& seq : list(root + "/sequences")) {
    DatasetReader reader(seq + "/");
    const int n = reader.getNumImages();
    for (int threads : {0, 3, 1}) {
      reader.setDecodeThreads(threads);
      reader.setPrefetch(threads ? 5 : 0);
      for (int i = 0; i < n + 1; i++) {
        int w = 0, h = 0;
        (void)reader.getImageRaw(i % 2 ? i : n - 1 - i, &w, &h);
      }
      for (int i = 0; i < n; i++) (void)reader.getImageRaw(i, 0, 0);
    }
    (void)reader.getTimestamp(-1);
    (void)reader.getExposure(n + 5);
    ExposureImage* img = reader.getImage(0, true, true, true, true);  // no GPU here: must return 0, not crash
    delete img;
    ExposureImage* out[4] = {0, 0, 0, 0};
    (void)reader.getImages(0, n < 4 ? n : 4, true, true, true, true, out);
  }
  {
    ExposureImage a(33, 17, 1.0, 2.0f, 3);
    a.image[33 * 17 - 1] = 1.f;
    ExposureImage* b = new ExposureImage(33, 17, 0, 0, 0);
    delete b;
    mdch_image_pool_trim();
  }
  {  // the pool's slabs: blocks back to back (561 floats -> 576 apart), lowest free address first, double and foreign frees ignored
    mdch_image_pool_trim();
    std::vector<float*> blk;
    for (int i = 0; i < 192; i++) {  // three full slabs
      blk.push_back(mdch_image_alloc(561));
      blk.back()[0] = blk.back()[560] = (float)i;
    }
    int adjacent = 0;
    for (size_t i = 1; i < blk.size(); i++) adjacent += blk[i] == blk[i - 1] + 576;
    if (adjacent < 185) {
      std::printf("image pool: only %d of 191 consecutive blocks lie back to back\n", adjacent);
      return 1;
    }
    mdch_image_free(blk[7]);
    mdch_image_free(blk[7]);
    mdch_image_free(blk[7] + 3);
    float on_stack[4];
    mdch_image_free(on_stack);
    mdch_image_free(blk[5]);
    float* again = mdch_image_alloc(561);
    float* again2 = mdch_image_alloc(561);
    if (!((again == blk[5] && again2 == blk[7]) || (again == blk[7] && again2 == blk[5]))) {
      std::printf("image pool: freed blocks did not come back\n");
      return 1;
    }
    if (mdch_image_pool_idle_bytes() != 0) {
      std::printf("image pool: releasable bytes while every slab has live images\n");
      return 1;
    }
    for (float* p : blk) mdch_image_free(p);
    if (mdch_image_pool_idle_bytes() == 0) {
      std::printf("image pool: nothing releasable after every image came back\n");
      return 1;
    }
    mdch_image_pool_trim();
    if (mdch_image_pool_idle_bytes() != 0) return 1;
    float* big = mdch_image_alloc(5u << 20);  // one image larger than a pageable slab
    big[(5u << 20) - 1] = 1.f;
    mdch_image_free(big);
    mdch_image_pool_trim();
  }
  std::printf("HOST_SANITIZE_OK decoded %ld refused %ld\n", decoded, refused);
  return 0;
}
