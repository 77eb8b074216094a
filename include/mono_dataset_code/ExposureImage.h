// Drop-in for the reference's src/ExposureImage.h:33-51 -- the frame container
// DatasetReader::getImage returns (src/BenchmarkDatasetReader.h:221).  Field
// names, order and the constructor signature are the public API that callers
// (main_playbackDataset.cpp:82,116, DSO-style consumers) rely on.
#pragma once

class ExposureImage {
 public:
  float* image;         // w*h irradiance / intensity values, row-major, owned
  double timestamp;     // seconds, from times.txt
  int w, h;
  float exposure_time;  // milliseconds, from times.txt (0 if unknown)
  int id;               // frame index in the sequence

  ExposureImage(int width, int height, double stamp, float exposure, int frame_id)
      : image(new float[static_cast<unsigned long>(width) * static_cast<unsigned long>(height)]),
        timestamp(stamp), w(width), h(height), exposure_time(exposure), id(frame_id) {}
  ~ExposureImage() { delete[] image; }

  // the reference type is used through pointers only; copying would double-free
  ExposureImage(const ExposureImage&) = delete;
  ExposureImage& operator=(const ExposureImage&) = delete;
};
