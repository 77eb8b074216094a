// C++ glue between the drop-in classes and the fused GPU entry points of include/mdc_hip.h.
//
// DatasetReader (reference src/BenchmarkDatasetReader.h:135-136) owns one UndistorterFOV and one
// PhotometricUndistorter; its getImage (:207-241) calls them one after the other.  A reader that
// wants the whole body in ONE launch -- mdc_process_host / mdc_process_frames_host /
// mdc_process_batch_device -- needs both objects' tables in one mdc_ctx:
//
//   mdc_ctx* gpu; mdc_create(-1, &gpu);
//   mdc_bind_objects(gpu, reader->getUndistorter(), reader->getPhotoUndistorter());
//
// (The C facade's mdch_bind, include/mdc_host.h, does the same for non-C++ callers.)
#pragma once
#include "FOVUndistorter.h"
#include "PhotometricUndistorter.h"
#include "mdc_hip.h"

// Uploads the tables of the two objects (either may be NULL) into `ctx`.  Returns MDC_OK or the
// status of mdc_set_photometric / mdc_set_remap.
MDC_API int mdc_bind_objects(mdc_ctx* ctx, const UndistorterFOV* fov, const PhotometricUndistorter* photo);

// The lens model of `fov` for mdc_distort_points_* (distortCoordinates on the GPU).
MDC_API void mdc_fov_model_of(const UndistorterFOV& fov, mdc_fov_model* model);
