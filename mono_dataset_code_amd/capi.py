"""ctypes bindings of include/mdc_hip.h (libmdc_hip.so) and include/mdc_host.h (libmdc_host.so).

Thin, explicit, no logic: every Python method is one C call.  Device buffers are
passed as integer addresses (e.g. torch.Tensor.data_ptr()), host buffers as numpy
arrays.  Loading fails loudly (OSError) when the libraries have not been built --
there is no Python or CPU fallback for the per-frame work.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_HIP_PATH = os.environ.get("MDC_LIB_HIP") or os.path.join(_PKG, "libmdc_hip.so")  # override: experiment builds
LIB_HOST_PATH = os.path.join(_PKG, "libmdc_host.so")

# flag word (include/mdc_hip.h)
GAMMA, VIGNETTE, KILL_OVEREXPOSED, RECTIFY = 1, 2, 4, 8
KERNEL_AUTO, KERNEL_GATHER, KERNEL_TILED = 0, 1, 2
OPT_KERNEL, OPT_FRAMES_PER_BLOCK, OPT_TILE_ROWS, OPT_TILE_ORDER, OPT_WINDOW_BUFFERS, OPT_FRAME_INTERLEAVE, OPT_TILE_COLS, OPT_PIN_CALLER_BUFFERS = 1, 2, 5, 6, 7, 8, 9, 10
OPT_TWO_STAGE = 11
OPT_PREFETCH_CHUNK = 12
OPT_PREFETCH_STREAMS = 13
OPT_ZERO_COPY = 14
OPT_TAIL_TAPER = 15
OPT_DEVICE_PIPELINE_CHUNK, OPT_DEVICE_PIPELINE_CHUNK_HINT = 16, 17
ORDER_BANDS, ORDER_ROWS, ORDER_IDENTITY, ORDER_BLOCKS2D = 0, 1, 2, 3
OK, ERR_ARG, ERR_STATE, ERR_SIZE, ERR_HIP, ERR_NO_DEVICE, ERR_NOMEM = 0, -1, -2, -3, -4, -5, -6

# every symbol include/mdc_hip.h declares (checked by tests/test_abi.py)
HIP_SYMBOLS = [
    "mdc_create", "mdc_destroy", "mdc_device_count", "mdc_device_pci_bus_id", "mdc_last_error", "mdc_build_flags", "mdc_code_id", "mdc_get_info", "mdc_set_option", "mdc_set_photometric",
    "mdc_set_remap", "mdc_unmap_host", "mdc_undistort_host_f32", "mdc_undistort_host_u8", "mdc_process_host",
    "mdc_process_frames_host_to_device", "mdc_process_jpeg_frames_host_to_device", "mdc_process_jpeg_streams_host_to_device", "mdc_device_alloc", "mdc_tune_placement_device", "mdc_alloc_placed_device", "mdc_free_placed_device", "mdc_alloc_striped_set_device", "mdc_free_striped_set_device", "mdc_device_free", "mdc_copy_to_host",
    "mdc_host_alloc", "mdc_host_free", "mdc_process_frames_host", "mdc_process_jpeg_frames_host", "mdc_jpeg_idct_batch_device", "mdc_process_jpeg_streams_host", "mdc_jpeg_huffman_batch_device",
    "mdc_unmap_batch_device", "mdc_process_batch_device", "mdc_undistort_batch_device_f32",
    "mdc_pyramid_batch_device", "mdc_process_pyramid_batch_device",
    "mdc_distort_points_device", "mdc_distort_points_host", "mdc_export_tables", "mdc_import_tables",
    "mdc_synchronize", "mdc_describe_launch", "mdc_vcal_plane_step_device",
    "mdc_vcal_vignette_step_device", "mdc_gradients_batch_device", "mdc_process_pyramid_gradients_batch_device", "mdc_tune_device",
    "mdc_vcal_index_create", "mdc_vcal_index_destroy", "mdc_vcal_index_bytes", "mdc_vcal_index_entries",
    "mdc_vcal_vignette_step_indexed_device", "mdc_vcal_solve_device", "mdc_vcal_smooth_device", "mdc_vcal_mask_coords_device", "mdc_vcal_gradient_mask_device", "mdc_vcal_scale_images_device",
]
HOST_SYMBOLS = [
    "mdch_fov_create", "mdch_fov_destroy", "mdch_fov_valid", "mdch_fov_has_gpu", "mdch_fov_dims",
    "mdch_fov_intrinsics", "mdch_fov_model", "mdch_fov_remap", "mdch_fov_distort", "mdch_fov_undistort_f32", "mdch_fov_undistort_u8",
    "mdch_photo_create", "mdch_photo_destroy", "mdch_photo_valid", "mdch_photo_has_gpu", "mdch_photo_ginv",
    "mdch_photo_g", "mdch_photo_vignette", "mdch_photo_unmap", "mdch_bind", "mdch_pack_tables",
    "mdch_reader_create", "mdch_reader_destroy", "mdch_reader_num_images", "mdch_reader_timestamp", "mdch_reader_exposure",
    "mdch_reader_dims", "mdch_reader_get_image", "mdch_reader_get_images", "mdch_reader_get_images_device", "mdch_reader_context", "mdch_reader_device", "mdch_reader_get_raw", "mdch_reader_set_threads",
    "mdch_reader_set_prefetch", "mdch_reader_set_gpu_jpeg", "mdch_reader_set_lookahead", "mdch_reader_last_error", "mdch_reader_prefetch_stats", "mdch_reader_device_stats", "mdch_decode_gray8", "mdch_jpeg_record_bytes",
    "mdch_decode_jpeg_record", "mdch_jpeg_stream", "mdch_image_alloc", "mdch_image_free",
    "mdch_image_pool_trim", "mdch_image_pool_idle_bytes",
]


PLACE_AUTO, PLACE_FIRST, PLACE_MALLOC, PLACE_VMM = 0, 1, 2, 3
PLACE_NAMES = {PLACE_AUTO: "auto", PLACE_FIRST: "first", PLACE_MALLOC: "malloc", PLACE_VMM: "vmm"}


class PlacedBuffers(C.Structure):
    """mdc_placed_buffers (include/mdc_hip.h)"""
    _fields_ = [("d_in", C.c_void_p), ("d_out", C.c_void_p), ("in_bytes", C.c_size_t), ("out_bytes", C.c_size_t), ("nframes", C.c_int64),
                ("probe_frames", C.c_int64), ("strategy", C.c_int), ("candidates_in", C.c_int), ("candidates_out", C.c_int), ("picked_in", C.c_int),
                ("picked_out", C.c_int), ("pair_ms", C.c_float * 64), ("pieces", C.c_int), ("piece_mib", C.c_int), ("class_count", C.c_int * 3),
                ("ms_first", C.c_float), ("ms_chosen", C.c_float), ("note", C.c_char * 384), ("handle", C.c_void_p)]

    def describe(self):
        """what was done, for a bench line / a log"""
        d = {"strategy": PLACE_NAMES.get(self.strategy, str(self.strategy)), "how": self.note.decode(errors="replace"),
             "probe_frames": int(self.probe_frames), "ms_on_first_allocations": round(float(self.ms_first), 4) or None,
             "ms_on_chosen_pair": round(float(self.ms_chosen), 4) or None}
        if self.strategy == PLACE_MALLOC and self.candidates_in > 1:
            ki, ko = self.candidates_in, self.candidates_out
            d["ms_frames_i_results_j"] = [[round(float(self.pair_ms[i * ko + j]), 4) for j in range(ko)] for i in range(ki)]
            d["picked_frames"], d["picked_results"] = int(self.picked_in), int(self.picked_out)
        if self.strategy == PLACE_VMM:
            d["pieces"], d["piece_mib"], d["class_count"] = int(self.pieces), int(self.piece_mib), [int(x) for x in self.class_count]
        return d


class StripedSet(C.Structure):
    """mdc_striped_set (include/mdc_hip.h)"""
    _fields_ = [("n", C.c_int), ("d_ptr", C.c_void_p * 16), ("bytes", C.c_size_t * 16), ("strategy", C.c_int), ("pieces", C.c_int), ("piece_mib", C.c_int),
                ("class_count", C.c_int * 3), ("note", C.c_char * 384), ("handle", C.c_void_p)]


class FovModel(C.Structure):
    _fields_ = [("in_calib", C.c_float * 5), ("in_w", C.c_int), ("in_h", C.c_int), ("out_calib", C.c_float * 5),
                ("out_w", C.c_int), ("out_h", C.c_int)]


class MdcInfo(C.Structure):
    _fields_ = [("device", C.c_int), ("in_w", C.c_int), ("in_h", C.c_int), ("out_w", C.c_int), ("out_h", C.c_int),
                ("valid_gamma", C.c_int), ("valid_vignette", C.c_int), ("valid_remap", C.c_int), ("tiled", C.c_int),
                ("tile_w", C.c_int), ("tile_h", C.c_int), ("n_tiles", C.c_int), ("lds_bytes", C.c_int),
                ("window_buffers", C.c_int), ("f32_tiled", C.c_int), ("f32_tile_w", C.c_int), ("f32_tile_h", C.c_int),
                ("src_bbox", C.c_int * 4), ("src_bbox_bytes", C.c_int64), ("src_staged_bytes", C.c_int64),
                ("n_black", C.c_int64), ("two_stage", C.c_int), ("prefetch_chunk", C.c_int), ("prefetch_streams", C.c_int)]


class DeviceOutputs(C.Structure):
    """include/mdc_hip.h: mdc_device_outputs -- device arrays the *_to_device calls / DatasetReader.get_images_device fill."""
    _fields_ = [("base", C.c_void_p), ("levels", C.c_int), ("level", C.c_void_p * 3), ("dI", C.c_void_p * 4), ("abs_squared_grad", C.c_void_p * 4)]

    @classmethod
    def make(cls, base, levels=1, level=(), dI=(), abs2=()):
        o = cls()
        o.base, o.levels = base, levels
        for i, p in enumerate(level):
            o.level[i] = p
        for i, p in enumerate(dI):
            o.dI[i] = p
        for i, p in enumerate(abs2):
            o.abs_squared_grad[i] = p
        return o


class TuneResult(C.Structure):
    _fields_ = [("tile_w", C.c_int), ("tile_h", C.c_int), ("frames_per_block", C.c_int), ("ms", C.c_float), ("candidates", C.c_int)]


class MdcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("mdc error %d: %s" % (code, msg))
        self.code = code


_hip = None
_host = None
_vp, _i, _i64, _u32, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_uint32, C.c_size_t


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64 (same soname as /opt/rocm's).
    Two HIP runtimes in one process cannot both own the GPU, so when torch is
    installed it must be loaded FIRST: libmdc_hip.so's NEEDED libamdhip64.so.7 then
    resolves to the copy torch already mapped.  Without torch the system runtime
    (RUNPATH /opt/rocm) is used."""
    try:
        import torch  # noqa: F401
    except ImportError:
        pass


LIB_BENCH_PATH = os.path.join(_PKG, "libmdc_bench.so")
BENCH_SYMBOLS = ["mdcb_synth_frames_device", "mdcb_ceiling_mix_device", "mdcb_alias_alloc", "mdcb_alias_free", "mdcb_chunked_alloc", "mdcb_marker_device"]  # include/mdc_bench.h (not the product ABI)
_bench = None


def bench_lib():
    """libmdc_bench.so: the synthetic sequence generator and the linear-stream yardstick (bench.py, tools/, tests/)."""
    global _bench
    if _bench is None:
        _share_hip_runtime_with_torch()
        if not os.path.exists(LIB_BENCH_PATH):
            raise OSError("%s not built: run `python -m mono_dataset_code_amd.build`" % LIB_BENCH_PATH)
        L = C.CDLL(LIB_BENCH_PATH)
        L.mdcb_synth_frames_device.argtypes = [_i, _vp, _i64, _i64, _i, _u32, _vp]
        L.mdcb_ceiling_mix_device.argtypes = [_i, _vp, C.c_int64, _vp, C.c_int64, _i, _i, _vp]
        L.mdcb_alias_alloc.argtypes = [_i, C.c_int64, _i, C.POINTER(_vp), C.POINTER(C.c_int64)]
        L.mdcb_alias_free.argtypes = [_i, _vp, C.c_int64, _i]
        L.mdcb_chunked_alloc.argtypes = [_i, C.c_int64, _i, _i, C.POINTER(_vp)]
        L.mdcb_marker_device.argtypes = [_i, _i, _vp]
        _bench = L
    return _bench


def hip_lib():
    global _hip
    if _hip is None:
        _share_hip_runtime_with_torch()
        if not os.path.exists(LIB_HIP_PATH):
            raise OSError("%s not built: run `python -m mono_dataset_code_amd.build`" % LIB_HIP_PATH)
        L = C.CDLL(LIB_HIP_PATH)
        L.mdc_create.argtypes = [_i, C.POINTER(_vp)]
        L.mdc_destroy.argtypes = [_vp]
        L.mdc_destroy.restype = None
        L.mdc_last_error.argtypes = [_vp]
        L.mdc_last_error.restype = C.c_char_p
        if hasattr(L, "mdc_build_flags"):  # (absent from libraries built before round 4: tools/sweep.py --libs)
            L.mdc_build_flags.argtypes = []
            L.mdc_build_flags.restype = C.c_char_p
        if hasattr(L, "mdc_code_id"):  # (absent from libraries built before round 5)
            L.mdc_code_id.argtypes = []
            L.mdc_code_id.restype = C.c_char_p
        L.mdc_get_info.argtypes = [_vp, C.POINTER(MdcInfo)]
        L.mdc_set_option.argtypes = [_vp, _i, _i]
        L.mdc_set_photometric.argtypes = [_vp, _vp, _vp, _i, _i]
        L.mdc_set_remap.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i]
        L.mdc_unmap_host.argtypes = [_vp, _vp, _vp, _i, C.c_uint]
        L.mdc_undistort_host_f32.argtypes = [_vp, _vp, _vp, _i, _i]
        L.mdc_undistort_host_u8.argtypes = [_vp, _vp, _vp, _i, _i]
        L.mdc_process_host.argtypes = [_vp, _vp, _vp, C.c_uint]
        L.mdc_host_alloc.argtypes = [_sz]
        L.mdc_host_alloc.restype = _vp
        L.mdc_host_free.argtypes = [_vp]
        L.mdc_host_free.restype = None
        L.mdc_process_frames_host.argtypes = [_vp, C.POINTER(_vp), C.POINTER(_vp), _i64, C.c_uint]
        if hasattr(L, "mdc_process_jpeg_frames_host"):
            L.mdc_process_jpeg_frames_host.argtypes = [_vp, C.POINTER(_vp), _i64, _i, _i, C.POINTER(_vp), _i64, C.c_uint]
            L.mdc_jpeg_idct_batch_device.argtypes = [_vp, _vp, _i64, _vp, _i, _i, _i, _i, _i64, _vp]
            L.mdc_process_jpeg_streams_host.argtypes = [_vp, C.POINTER(_vp), C.POINTER(C.c_int64), C.POINTER(_vp), _i64, C.c_uint, C.POINTER(C.c_int)]
            L.mdc_jpeg_huffman_batch_device.argtypes = [_vp, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _i64, _vp, _vp]
        if hasattr(L, "mdc_process_jpeg_streams_host_to_device"):  # (absent from libraries built before round 5)
            L.mdc_process_frames_host_to_device.argtypes = [_vp, C.POINTER(_vp), _i64, C.c_uint, C.POINTER(DeviceOutputs), C.POINTER(C.c_int64)]
            L.mdc_process_jpeg_frames_host_to_device.argtypes = [_vp, C.POINTER(_vp), _i64, _i, _i, _i64, C.c_uint, C.POINTER(DeviceOutputs), C.POINTER(C.c_int64)]
            L.mdc_process_jpeg_streams_host_to_device.argtypes = [_vp, C.POINTER(_vp), C.POINTER(C.c_int64), _i64, C.c_uint, C.POINTER(DeviceOutputs),
                                                                  C.POINTER(C.c_int64), C.POINTER(C.c_int)]
            L.mdc_device_alloc.argtypes = [_vp, _sz, C.POINTER(_vp)]
            L.mdc_tune_placement_device.argtypes = [_vp, C.POINTER(_vp), _i, C.POINTER(_vp), _i, _i64, C.c_uint, _vp, C.POINTER(_i), C.POINTER(_i),
                                                    C.POINTER(C.c_float)]
            L.mdc_device_free.argtypes = [_vp, _vp]
            L.mdc_device_free.restype = None
        if hasattr(L, "mdc_alloc_placed_device"):  # (absent from libraries built before round 6)
            L.mdc_alloc_placed_device.argtypes = [_vp, _sz, _sz, _i64, C.c_uint, _i, _vp, C.POINTER(PlacedBuffers)]
            L.mdc_free_placed_device.argtypes = [_vp, C.POINTER(PlacedBuffers)]
            L.mdc_alloc_striped_set_device.argtypes = [_vp, _i, C.POINTER(_sz), _vp, C.POINTER(StripedSet)]
            L.mdc_free_striped_set_device.argtypes = [_vp, C.POINTER(StripedSet)]
            L.mdc_copy_to_host.argtypes = [_vp, _vp, _vp, _sz]
        L.mdc_unmap_batch_device.argtypes = [_vp, _vp, _vp, _i64, C.c_uint, _vp]
        L.mdc_process_batch_device.argtypes = [_vp, _vp, _vp, _i64, C.c_uint, _vp]
        L.mdc_undistort_batch_device_f32.argtypes = [_vp, _vp, _vp, _i64, _vp]
        L.mdc_pyramid_batch_device.argtypes = [_vp, _vp, _i, _i, _i, C.POINTER(_vp), _i64, _vp]
        L.mdc_process_pyramid_batch_device.argtypes = [_vp, _vp, _vp, _i, C.POINTER(_vp), _i64, C.c_uint, _vp]
        L.mdc_distort_points_device.argtypes = [_vp, C.POINTER(FovModel), _vp, _vp, _i64, _vp]
        L.mdc_distort_points_host.argtypes = [_vp, C.POINTER(FovModel), _vp, _vp, _i64]
        L.mdc_export_tables.argtypes = [_vp, _vp, _sz, C.POINTER(_sz)]
        L.mdc_import_tables.argtypes = [_vp, _vp, _sz]
        L.mdc_synchronize.argtypes = [_vp]
        if hasattr(L, "mdc_device_pci_bus_id"):
            L.mdc_device_pci_bus_id.argtypes = [_vp, C.c_char_p, _sz]
        old_build = LIB_HIP_PATH != os.path.join(_PKG, "libmdc_hip.so")  # tools/sweep.py --libs: A/B against earlier builds
        if not old_build or hasattr(L, "mdc_describe_launch"):
            L.mdc_describe_launch.argtypes = [_vp, C.c_uint, _i, C.c_char_p, _sz]
            L.mdc_vcal_plane_step_device.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp]
            L.mdc_tune_device.argtypes = [_vp, _vp, _vp, C.c_int64, C.c_uint, _vp, C.POINTER(TuneResult)]
            L.mdc_gradients_batch_device.argtypes = [_vp, _vp, _i, _i, _vp, _vp, C.c_int64, _vp]
            if hasattr(L, "mdc_process_pyramid_gradients_batch_device"):
                L.mdc_process_pyramid_gradients_batch_device.argtypes = [_vp, _vp, _vp, _i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _i64,
                                                                         C.c_uint, _i, _vp]
            L.mdc_vcal_vignette_step_device.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp]
        if not old_build or hasattr(L, "mdc_vcal_index_create"):
            L.mdc_vcal_index_create.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, C.POINTER(_vp)]
            L.mdc_vcal_index_destroy.argtypes = [_vp]
            L.mdc_vcal_index_destroy.restype = None
            L.mdc_vcal_index_bytes.argtypes = [_vp]
            L.mdc_vcal_index_bytes.restype = C.c_int64
            L.mdc_vcal_index_entries.argtypes = [_vp]
            L.mdc_vcal_index_entries.restype = C.c_int64
            L.mdc_vcal_vignette_step_indexed_device.argtypes = [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]
        if not old_build or hasattr(L, "mdc_vcal_solve_device"):
            L.mdc_vcal_solve_device.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp, _vp]
            L.mdc_vcal_smooth_device.argtypes = [_vp, _vp, _i, _i, _vp, _vp, _vp]
            L.mdc_vcal_mask_coords_device.argtypes = [_vp, _vp, _vp, C.c_int64, _i, _i, _vp]
            L.mdc_vcal_gradient_mask_device.argtypes = [_vp, _vp, _i, _i, _i, _i, _vp]
            L.mdc_vcal_scale_images_device.argtypes = [_vp, _vp, _i, C.c_int64, C.c_float, _vp, _vp]
        for n in HIP_SYMBOLS:
            if old_build and not hasattr(L, n):
                continue
            if n not in ("mdc_destroy", "mdc_last_error", "mdc_build_flags", "mdc_code_id", "mdc_device_free", "mdc_host_alloc", "mdc_host_free", "mdc_vcal_index_destroy",
                         "mdc_vcal_index_bytes", "mdc_vcal_index_entries"):
                getattr(L, n).restype = _i
        _hip = L
    return _hip


def build_flags():
    """Build-time switches of the loaded libmdc_hip.so that are not at their shipped value ("" = the product build)."""
    L = hip_lib()
    return L.mdc_build_flags().decode() if hasattr(L, "mdc_build_flags") else "unknown (library predates mdc_build_flags)"


def code_id():
    """Identity of the loaded libmdc_hip.so's kernel build (hash of sources + flags, include/mdc_hip.h: mdc_code_id)."""
    L = hip_lib()
    return L.mdc_code_id().decode() if hasattr(L, "mdc_code_id") else None


def host_lib():
    global _host
    if _host is None:
        hip_lib()
        if not os.path.exists(LIB_HOST_PATH):
            raise OSError("%s not built: run `python -m mono_dataset_code_amd.build`" % LIB_HOST_PATH)
        L = C.CDLL(LIB_HOST_PATH)
        L.mdch_fov_create.argtypes = [C.c_char_p]
        L.mdch_fov_create.restype = _vp
        L.mdch_fov_destroy.argtypes = [_vp]
        L.mdch_fov_destroy.restype = None
        for n in ("mdch_fov_valid", "mdch_fov_has_gpu"):
            getattr(L, n).argtypes = [_vp]
            getattr(L, n).restype = _i
        L.mdch_fov_dims.argtypes = [_vp, _vp]
        L.mdch_fov_dims.restype = None
        L.mdch_fov_intrinsics.argtypes = [_vp, _vp]
        L.mdch_fov_intrinsics.restype = None
        L.mdch_fov_model.argtypes = [_vp, C.POINTER(FovModel)]
        L.mdch_fov_model.restype = None
        L.mdch_fov_remap.argtypes = [_vp, _vp, _vp]
        L.mdch_fov_remap.restype = _i
        L.mdch_fov_distort.argtypes = [_vp, _vp, _vp, _i]
        L.mdch_fov_distort.restype = None
        L.mdch_fov_undistort_f32.argtypes = [_vp, _vp, _vp, _i, _i]
        L.mdch_fov_undistort_f32.restype = None
        L.mdch_fov_undistort_u8.argtypes = [_vp, _vp, _vp, _i, _i]
        L.mdch_fov_undistort_u8.restype = None
        L.mdch_photo_create.argtypes = [C.c_char_p, C.c_char_p, _i, _i]
        L.mdch_photo_create.restype = _vp
        L.mdch_photo_destroy.argtypes = [_vp]
        L.mdch_photo_destroy.restype = None
        for n in ("mdch_photo_valid", "mdch_photo_has_gpu"):
            getattr(L, n).argtypes = [_vp]
            getattr(L, n).restype = _i
        L.mdch_photo_ginv.argtypes = [_vp, _vp]
        L.mdch_photo_ginv.restype = _i
        L.mdch_photo_g.argtypes = [_vp, _vp]
        L.mdch_photo_g.restype = _i
        L.mdch_photo_vignette.argtypes = [_vp, _vp, _vp]
        L.mdch_photo_vignette.restype = _i
        L.mdch_photo_unmap.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i]
        L.mdch_photo_unmap.restype = None
        L.mdch_bind.argtypes = [_vp, _vp, _vp]
        L.mdch_bind.restype = _i
        L.mdch_pack_tables.argtypes = [_vp, _vp, _vp, _sz, C.POINTER(_sz)]
        L.mdch_pack_tables.restype = _i
        L.mdch_reader_create.argtypes = [C.c_char_p]
        L.mdch_reader_create.restype = _vp
        L.mdch_reader_destroy.argtypes = [_vp]
        L.mdch_reader_destroy.restype = None
        L.mdch_reader_num_images.argtypes = [_vp]
        L.mdch_reader_timestamp.argtypes = [_vp, _i]
        L.mdch_reader_timestamp.restype = C.c_double
        L.mdch_reader_exposure.argtypes = [_vp, _i]
        L.mdch_reader_exposure.restype = C.c_float
        L.mdch_reader_dims.argtypes = [_vp, _vp]
        L.mdch_reader_dims.restype = None
        L.mdch_reader_get_image.argtypes = [_vp, _i, _i, _i, _i, _i, _vp, C.c_long, _vp, C.POINTER(C.c_double), C.POINTER(C.c_float)]
        L.mdch_reader_get_images.argtypes = [_vp, _i, _i, _i, _i, _i, _i, _vp, C.c_long, _vp]
        L.mdch_reader_get_images_device.argtypes = [_vp, _i, _i, _i, _i, _i, _i, C.POINTER(DeviceOutputs), _vp]
        L.mdch_reader_context.argtypes = [_vp]
        L.mdch_reader_context.restype = _vp
        L.mdch_reader_device.argtypes = [_vp]
        L.mdch_reader_get_raw.argtypes = [_vp, _i, _vp, C.c_long, _vp]
        L.mdch_reader_set_threads.argtypes = [_vp, _i]
        L.mdch_reader_set_threads.restype = None
        L.mdch_reader_set_prefetch.argtypes = [_vp, _i]
        L.mdch_reader_set_gpu_jpeg.argtypes = [_vp, _i]
        L.mdch_reader_set_lookahead.argtypes = [_vp, _i]
        L.mdch_reader_set_prefetch.restype = None
        L.mdch_reader_last_error.argtypes = [_vp]
        L.mdch_reader_last_error.restype = C.c_char_p
        L.mdch_reader_prefetch_stats.argtypes = [_vp, _vp]
        L.mdch_reader_prefetch_stats.restype = None
        L.mdch_reader_device_stats.argtypes = [_vp, _i, _vp, _vp]
        L.mdch_decode_gray8.argtypes = [_vp, _sz, _vp, _sz, _vp, C.c_char_p, _sz]
        L.mdch_jpeg_record_bytes.argtypes = [_i, _i, _vp]
        L.mdch_jpeg_record_bytes.restype = _sz
        L.mdch_decode_jpeg_record.argtypes = [_vp, _sz, _vp, _sz, _i, _vp, C.c_char_p, _sz]
        L.mdch_jpeg_stream.argtypes = [_vp, _sz, _vp, _sz, _vp, C.c_char_p, _sz]
        L.mdch_jpeg_stream.restype = C.c_longlong
        L.mdch_image_alloc.argtypes = [C.c_ulong]
        L.mdch_image_alloc.restype = _vp
        L.mdch_image_free.argtypes = [_vp]
        L.mdch_image_free.restype = None
        L.mdch_image_pool_trim.restype = None
        L.mdch_image_pool_idle_bytes.restype = C.c_ulong
        _host = L
    return _host


def _np_ptr(a):
    return a.ctypes.data_as(_vp) if a is not None else None


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


class PinnedArray:
    """A numpy view of page-locked host memory (mdc_host_alloc); keep the object alive while the view is used."""

    def __init__(self, shape, dtype):
        self._L = hip_lib()
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self._p = self._L.mdc_host_alloc(self.nbytes)
        if not self._p:
            raise MemoryError("mdc_host_alloc(%d) failed" % self.nbytes)
        buf = (C.c_char * self.nbytes).from_address(self._p)
        self.array = np.frombuffer(buf, dtype=dtype).reshape(shape)

    def __del__(self):
        if getattr(self, "_p", None):
            self.array = None
            self._L.mdc_host_free(self._p)
            self._p = None


class Context:
    """One mdc_ctx (one GPU).  Methods map 1:1 onto include/mdc_hip.h."""

    def __init__(self, device=0):
        self._L = hip_lib()
        h = _vp()
        rc = self._L.mdc_create(int(device), C.byref(h))
        if rc != OK:
            raise MdcError(rc, self._L.mdc_last_error(None).decode())
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.mdc_destroy(self._h)
            self._h = None

    __del__ = close

    @property
    def handle(self):
        return self._h

    def _chk(self, rc):
        if rc != OK:
            raise MdcError(rc, self._L.mdc_last_error(self._h).decode())

    def last_error(self):
        return self._L.mdc_last_error(self._h).decode()

    def info(self):
        i = MdcInfo()
        self._chk(self._L.mdc_get_info(self._h, C.byref(i)))
        return i

    def device(self):
        return self.info().device

    def set_option(self, opt, value):
        self._chk(self._L.mdc_set_option(self._h, opt, value))

    def set_photometric(self, ginv, vinv, w, h):
        g = _f32(ginv) if ginv is not None else None
        v = _f32(vinv) if vinv is not None else None
        if g is not None:
            assert g.size == 256
        if v is not None:
            assert v.size == w * h
        self._chk(self._L.mdc_set_photometric(self._h, _np_ptr(g), _np_ptr(v), w, h))

    def set_remap(self, rx, ry, in_w, in_h, out_w, out_h):
        if rx is None:
            self._chk(self._L.mdc_set_remap(self._h, None, None, 0, 0, 0, 0))
            return
        rx, ry = _f32(rx), _f32(ry)
        assert rx.size == out_w * out_h and ry.size == out_w * out_h
        self._chk(self._L.mdc_set_remap(self._h, _np_ptr(rx), _np_ptr(ry), in_w, in_h, out_w, out_h))

    # host-pointer single-frame calls: return the status code, raise only if asked
    def unmap_host(self, img_u8, out_f32, flags, check=True):
        rc = self._L.mdc_unmap_host(self._h, _np_ptr(img_u8), _np_ptr(out_f32), img_u8.size, flags)
        if check:
            self._chk(rc)
        return rc

    def undistort_host(self, img, out_f32, check=True):
        fn = self._L.mdc_undistort_host_f32 if img.dtype == np.float32 else self._L.mdc_undistort_host_u8
        rc = fn(self._h, _np_ptr(img), _np_ptr(out_f32), img.size, out_f32.size)
        if check:
            self._chk(rc)
        return rc

    def process_host(self, raw_u8, out_f32, flags, check=True):
        rc = self._L.mdc_process_host(self._h, _np_ptr(raw_u8), _np_ptr(out_f32), flags)
        if check:
            self._chk(rc)
        return rc

    def process_frames_host(self, raws, outs, flags):
        """raws / outs: sequences of numpy arrays (u8 frames, f32 results), one pair per frame."""
        n = len(raws)
        assert len(outs) == n
        a = (_vp * max(1, n))(*[_np_ptr(r) for r in raws])
        b = (_vp * max(1, n))(*[_np_ptr(o) for o in outs])
        self._chk(self._L.mdc_process_frames_host(self._h, a, b, n, flags))

    def process_jpeg_frames_host(self, records, record_bytes, blocks_w, blocks_rows, outs, flags):
        """records: numpy uint8 arrays (one JPEG coefficient record per frame, decode_jpeg_record), outs: f32 result arrays."""
        n = len(records)
        assert len(outs) == n
        a = (_vp * max(1, n))(*[_np_ptr(r) for r in records])
        b = (_vp * max(1, n))(*[_np_ptr(o) for o in outs])
        self._chk(self._L.mdc_process_jpeg_frames_host(self._h, a, record_bytes, blocks_w, blocks_rows, b, n, flags))

    def process_jpeg_streams_host(self, streams, sizes, outs, flags):
        """streams: numpy uint8 arrays (jpeg_stream), sizes: bytes used of each, outs: f32 result arrays -> per-frame status list."""
        n = len(streams)
        assert len(outs) == n and len(sizes) == n
        a = (_vp * max(1, n))(*[_np_ptr(r) for r in streams])
        b = (_vp * max(1, n))(*[_np_ptr(o) for o in outs])
        sz = (C.c_int64 * max(1, n))(*[int(x) for x in sizes])
        st = (C.c_int * max(1, n))()
        self._chk(self._L.mdc_process_jpeg_streams_host(self._h, a, sz, b, n, flags, st))
        return [int(st[i]) for i in range(n)]

    def process_frames_host_to_device(self, raws, flags, outputs, frame_index=None):
        """raw u8 frames (numpy) -> the device arrays of `outputs` (DeviceOutputs); frame i at position frame_index[i] (None: i)."""
        n = len(raws)
        a = (_vp * max(1, n))(*[_np_ptr(r) for r in raws])
        idx = None if frame_index is None else (C.c_int64 * max(1, n))(*[int(x) for x in frame_index])
        self._chk(self._L.mdc_process_frames_host_to_device(self._h, a, n, flags, C.byref(outputs), idx))

    def process_jpeg_streams_host_to_device(self, streams, sizes, flags, outputs, frame_index=None):
        n = len(streams)
        a = (_vp * max(1, n))(*[_np_ptr(r) for r in streams])
        sz = (C.c_int64 * max(1, n))(*[int(x) for x in sizes])
        idx = None if frame_index is None else (C.c_int64 * max(1, n))(*[int(x) for x in frame_index])
        st = (C.c_int * max(1, n))()
        self._chk(self._L.mdc_process_jpeg_streams_host_to_device(self._h, a, sz, n, flags, C.byref(outputs), idx, st))
        return [int(st[i]) for i in range(n)]

    def jpeg_huffman_batch(self, d_streams, stream_stride, d_records, record_bytes, w, h, blocks_w, blocks_rows, nframes, d_status, stream=0):
        self._chk(self._L.mdc_jpeg_huffman_batch_device(self._h, d_streams, stream_stride, d_records, record_bytes, w, h, blocks_w, blocks_rows,
                                                        nframes, d_status, stream if stream else None))

    def jpeg_idct_batch(self, d_records, record_bytes, d_frames, w, h, blocks_w, blocks_rows, nframes, stream=0):
        self._chk(self._L.mdc_jpeg_idct_batch_device(self._h, d_records, record_bytes, d_frames, w, h, blocks_w, blocks_rows, nframes,
                                                     stream if stream else None))

    # device-pointer batched calls (addresses as ints)
    def unmap_batch(self, d_in, d_out, nframes, flags, stream=0):
        self._chk(self._L.mdc_unmap_batch_device(self._h, d_in, d_out, nframes, flags, stream if stream else None))

    def process_batch(self, d_in, d_out, nframes, flags, stream=0):
        self._chk(self._L.mdc_process_batch_device(self._h, d_in, d_out, nframes, flags, stream if stream else None))

    def undistort_batch_f32(self, d_in, d_out, nframes, stream=0):
        self._chk(self._L.mdc_undistort_batch_device_f32(self._h, d_in, d_out, nframes, stream if stream else None))

    def pyramid_batch(self, d_base, w, h, levels, d_levels, nframes, stream=0):
        arr = (_vp * max(1, len(d_levels)))(*d_levels)
        self._chk(self._L.mdc_pyramid_batch_device(self._h, d_base, w, h, levels, arr, nframes, stream if stream else None))

    def process_pyramid_batch(self, d_in, d_base, levels, d_levels, nframes, flags, stream=0):
        arr = (_vp * max(1, len(d_levels)))(*d_levels)
        self._chk(self._L.mdc_process_pyramid_batch_device(self._h, d_in, d_base, levels, arr, nframes, flags,
                                                           stream if stream else None))

    def process_pyramid_gradients_batch(self, d_in, d_base, levels, d_levels, d_dI, d_abs, nframes, flags, chunk_frames=0, stream=0):
        """base + levels + (I, dx, dy) / absSquaredGrad of every level in one call; d_dI / d_abs: one address per level (0 = base)."""
        lv = (_vp * max(1, len(d_levels)))(*d_levels)
        di = (_vp * levels)(*d_dI)
        ab = (_vp * levels)(*d_abs)
        self._chk(self._L.mdc_process_pyramid_gradients_batch_device(self._h, d_in, d_base, levels, lv, di, ab, nframes, flags, chunk_frames,
                                                                     stream if stream else None))

    def distort_points_host(self, model, x, y):
        assert x.dtype == np.float32 and y.dtype == np.float32 and x.size == y.size
        self._chk(self._L.mdc_distort_points_host(self._h, C.byref(model), _np_ptr(x), _np_ptr(y), x.size))

    def distort_points_device(self, model, d_x, d_y, n, stream=0):
        self._chk(self._L.mdc_distort_points_device(self._h, C.byref(model), d_x, d_y, n, stream if stream else None))

    def synth_frames(self, d_out, first_frame, nframes, npix, seed, stream=0):
        """(bench / test utility, libmdc_bench.so -- not part of the product ABI)"""
        rc = bench_lib().mdcb_synth_frames_device(self.device(), d_out, first_frame, nframes, npix, seed, stream if stream else None)
        if rc != 0:
            raise MdcError(rc, "mdcb_synth_frames_device failed")

    def export_tables(self):
        n = _sz(0)
        self._chk(self._L.mdc_export_tables(self._h, None, 0, C.byref(n)))
        buf = np.zeros(n.value, dtype=np.uint8)
        self._chk(self._L.mdc_export_tables(self._h, _np_ptr(buf), buf.size, C.byref(n)))
        return buf

    def import_tables(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        self._chk(self._L.mdc_import_tables(self._h, _np_ptr(blob), blob.size))

    def synchronize(self):
        self._chk(self._L.mdc_synchronize(self._h))

    def tune_placement(self, d_ins, d_outs, nframes, flags, stream=0):
        """Which PAIR of the candidate input / output buffers (device addresses) does the fused pass run fastest on?  include/mdc_hip.h:
        mdc_tune_placement_device.  -> (best input index, best output index, ms[i][j] for input i with output j)"""
        ni, no = len(d_ins), len(d_outs)
        a, b = (_vp * max(ni, 1))(*d_ins), (_vp * max(no, 1))(*d_outs)
        bi, bo = _i(0), _i(0)
        ms = (C.c_float * max(ni * no, 1))()
        self._chk(self._L.mdc_tune_placement_device(self._h, a, ni, b, no, nframes, flags, stream if stream else None, C.byref(bi), C.byref(bo), ms))
        return bi.value, bo.value, [[float(ms[i * no + j]) for j in range(no)] for i in range(ni)]

    def alloc_placed(self, nframes, flags, strategy=PLACE_AUTO, stream=0, in_bytes=0, out_bytes=0):
        """A frame buffer and a result buffer for nframes frames of the pass `flags`, placed by measurement (include/mdc_hip.h:
        mdc_alloc_placed_device).  -> PlacedBuffers; give it back with free_placed()."""
        b = PlacedBuffers()
        self._chk(self._L.mdc_alloc_placed_device(self._h, in_bytes, out_bytes, nframes, flags, strategy, stream if stream else None, C.byref(b)))
        return b

    def free_placed(self, b):
        self._chk(self._L.mdc_free_placed_device(self._h, C.byref(b)))

    def alloc_striped_set(self, sizes, stream=0):
        """len(sizes) device buffers, each striped over the device's memory classes (include/mdc_hip.h: mdc_alloc_striped_set_device)
        -> StripedSet (d_ptr[k] = buffer k); give it back with free_striped_set()."""
        b = StripedSet()
        arr = (_sz * len(sizes))(*[int(x) for x in sizes])
        self._chk(self._L.mdc_alloc_striped_set_device(self._h, len(sizes), arr, stream if stream else None, C.byref(b)))
        return b

    def free_striped_set(self, b):
        self._chk(self._L.mdc_free_striped_set_device(self._h, C.byref(b)))

    def copy_to_host(self, d_src, count, dtype):
        """count elements of dtype from device address d_src -> numpy array (mdc_copy_to_host: blocking)"""
        out = np.empty(count, dtype=dtype)
        self._chk(self._L.mdc_copy_to_host(self._h, _np_ptr(out), d_src, out.nbytes))
        return out

    def device_alloc(self, nbytes):
        p = _vp()
        self._chk(self._L.mdc_device_alloc(self._h, nbytes, C.byref(p)))
        return p.value

    def device_free(self, d_ptr):
        self._L.mdc_device_free(self._h, d_ptr)

    def pci_bus_id(self):
        buf = C.create_string_buffer(32)
        self._chk(self._L.mdc_device_pci_bus_id(self._h, buf, 32))
        return buf.value.decode()

    def describe_launch(self, flags, pyramid_levels=0):
        buf = C.create_string_buffer(256)
        self._chk(self._L.mdc_describe_launch(self._h, flags, pyramid_levels, buf, 256))
        return buf.value.decode()

    def marker(self, ident, stream=0):
        """(bench utility, libmdc_bench.so) a no-op kernel named mdcb_marker_kernel on `stream`: a cut mark in a profiler's kernel trace"""
        if bench_lib().mdcb_marker_device(self.device(), ident, stream if stream else None) != 0:
            raise MdcError(-4, "mdcb_marker_device failed")

    def ceiling_mix(self, d_read, read_bytes, d_write, write_bytes, blocks=16384, span=0, stream=0):
        """(bench utility, libmdc_bench.so -- not part of the product ABI)"""
        rc = bench_lib().mdcb_ceiling_mix_device(self.device(), d_read, read_bytes, d_write, write_bytes, blocks, span, stream if stream else None)
        if rc != 0:
            raise MdcError(rc, "mdcb_ceiling_mix_device failed")

    def tune(self, d_in, d_out, nframes, flags, stream=0):
        r = TuneResult()
        self._chk(self._L.mdc_tune_device(self._h, d_in, d_out, nframes, flags, stream if stream else None, C.byref(r)))
        return r

    def gradients_batch(self, d_level, w, h, d_dI, d_abs, nframes, stream=0):
        self._chk(self._L.mdc_gradients_batch_device(self._h, d_level, w, h, d_dI, d_abs, nframes, stream if stream else None))

    def vcal_plane_step(self, d_images, d_p2x, d_p2y, d_plane_color, d_vig, oth2, stream=0):
        """torch tensors on the device; d_plane_color is updated in place -> (FF, FC, E, R)."""
        import torch

        n, h, w = d_images.shape
        npnt = d_p2x.shape[1]
        ff = torch.empty(npnt, dtype=torch.float32, device=d_images.device)
        fc = torch.empty_like(ff)
        er = torch.zeros(2, dtype=torch.float64, device=d_images.device)
        self._chk(self._L.mdc_vcal_plane_step_device(self._h, d_images.data_ptr(), d_p2x.data_ptr(), d_p2y.data_ptr(), n, w, h, npnt,
                                                     d_plane_color.data_ptr(), d_vig.data_ptr(), int(oth2), ff.data_ptr(), fc.data_ptr(),
                                                     er.data_ptr(), stream if stream else None))
        e, r = er.cpu().tolist()
        return ff, fc, e, r

    def vcal_vignette_step(self, d_images, d_p2x, d_p2y, d_plane_color, d_vig, oth2, stream=0):
        """d_vig is updated in place -> (TT, CT, E, R)."""
        import torch

        n, h, w = d_images.shape
        npnt = d_p2x.shape[1]
        tt = torch.empty(h * w, dtype=torch.float32, device=d_images.device)
        ct = torch.empty_like(tt)
        er = torch.zeros(2, dtype=torch.float64, device=d_images.device)
        self._chk(self._L.mdc_vcal_vignette_step_device(self._h, d_images.data_ptr(), d_p2x.data_ptr(), d_p2y.data_ptr(), n, w, h, npnt,
                                                        d_plane_color.data_ptr(), d_vig.data_ptr(), int(oth2), tt.data_ptr(), ct.data_ptr(),
                                                        er.data_ptr(), stream if stream else None))
        e, r = er.cpu().tolist()
        return tt, ct, e, r

    def vcal_solve(self, d_images, d_p2x, d_p2y, d_plane_color, d_vig, max_iterations=20, outlier_th=15, stream=0):
        """The reference's whole iteration loop (src/main_vignetteCalib.cpp:395-527); d_plane_color and d_vig are updated
        in place -> array [max_iterations][4] = E, R of the plane step, E, R of the vignette step."""
        n, h, w = d_images.shape
        er = np.zeros((max(max_iterations, 0), 4), np.float64)
        self._chk(self._L.mdc_vcal_solve_device(self._h, d_images.data_ptr(), d_p2x.data_ptr(), d_p2y.data_ptr(), n, w, h, d_p2x.shape[1],
                                                d_plane_color.data_ptr(), d_vig.data_ptr(), int(max_iterations), int(outlier_th),
                                                _np_ptr(er), stream if stream else None))
        return er

    def vcal_scale_images(self, d_images, mean_exposure, d_exposure_times, stream=0):
        """image k = mean_exposure * image k / exposure_time k, in place (src/main_vignetteCalib.cpp:286-291)."""
        n = d_images.shape[0]
        self._chk(self._L.mdc_vcal_scale_images_device(self._h, d_images.data_ptr(), n, d_images[0].numel(), float(mean_exposure),
                                                       d_exposure_times.data_ptr(), stream if stream else None))

    def vcal_gradient_mask(self, d_images, max_abs_grad=255, stream=0):
        """Gradient mask of a stack of calibration images (n, h, w), in place (src/main_vignetteCalib.cpp:293-301)."""
        n, h, w = d_images.shape
        self._chk(self._L.mdc_vcal_gradient_mask_device(self._h, d_images.data_ptr(), n, w, h, int(max_abs_grad), stream if stream else None))

    def vcal_mask_coords(self, d_x, d_y, w, h, stream=0):
        """NaN coordinates for plane points outside the w x h image (src/main_vignetteCalib.cpp:345-357), in place."""
        self._chk(self._L.mdc_vcal_mask_coords_device(self._h, d_x.data_ptr(), d_y.data_ptr(), d_x.numel(), w, h, stream if stream else None))

    def vcal_smooth(self, d_vig, w, h, stream=0):
        """vignetteCalib's output smoothing (src/main_vignetteCalib.cpp:541-566) -> (smoothed, scratch) device tensors."""
        import torch

        tt, ct = torch.empty_like(d_vig), torch.empty_like(d_vig)
        self._chk(self._L.mdc_vcal_smooth_device(self._h, d_vig.data_ptr(), w, h, tt.data_ptr(), ct.data_ptr(), stream if stream else None))
        return tt, ct

    def vcal_index(self, d_images, d_p2x, d_p2y, stream=0):
        """Contribution index of the vignette half-iteration for these images / coordinates (mdc_vcal_index_create)."""
        return VcalIndex(self, d_images, d_p2x, d_p2y, stream)

    def vcal_vignette_step_indexed(self, index, d_plane_color, d_vig, oth2, stream=0):
        """The vignette half-iteration as an ordered gather (bit-identical to the reference); d_vig is updated in place
        -> (TT, CT, E, R)."""
        import torch

        tt = torch.empty(index.h * index.w, dtype=torch.float32, device=d_vig.device)
        ct = torch.empty_like(tt)
        er = torch.zeros(2, dtype=torch.float64, device=d_vig.device)
        self._chk(self._L.mdc_vcal_vignette_step_indexed_device(self._h, index._h, d_plane_color.data_ptr(), d_vig.data_ptr(), int(oth2),
                                                                tt.data_ptr(), ct.data_ptr(), er.data_ptr(), stream if stream else None))
        e, r = er.cpu().tolist()
        return tt, ct, e, r

    def bind(self, fov=None, photo=None):
        rc = host_lib().mdch_bind(self._h, fov._h if fov is not None else None, photo._h if photo is not None else None)
        self._chk(rc)


class VcalIndex:
    """mdc_vcal_index: per image pixel, the (image, plane point, corner) contributions in the reference's order."""

    def __init__(self, ctx, d_images, d_p2x, d_p2y, stream=0):
        self._L = ctx._L
        self._h = _vp()
        n, self.h, self.w = d_images.shape
        ctx._chk(self._L.mdc_vcal_index_create(ctx._h, d_images.data_ptr(), d_p2x.data_ptr(), d_p2y.data_ptr(), n, self.w, self.h,
                                               d_p2x.shape[1], stream if stream else None, C.byref(self._h)))
        self.bytes = self._L.mdc_vcal_index_bytes(self._h)
        self.entries = self._L.mdc_vcal_index_entries(self._h)

    def close(self):
        if self._h:
            self._L.mdc_vcal_index_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pack_tables(fov=None, photo=None):
    """Host-side table blob (mdch_pack_tables): what rank 0 broadcasts; needs no GPU."""
    L = host_lib()
    n = _sz(0)
    fh = fov._h if fov is not None else None
    ph = photo._h if photo is not None else None
    if L.mdch_pack_tables(fh, ph, None, 0, C.byref(n)) != OK:
        raise MdcError(ERR_ARG, "mdch_pack_tables")
    buf = np.zeros(n.value, dtype=np.uint8)
    if L.mdch_pack_tables(fh, ph, _np_ptr(buf), buf.size, C.byref(n)) != OK:
        raise MdcError(ERR_ARG, "mdch_pack_tables")
    return buf


class UndistorterFOV:
    """Handle on the C++ class UndistorterFOV (include/mono_dataset_code/FOVUndistorter.h)."""

    def __init__(self, camera_txt):
        self._L = host_lib()
        self._h = self._L.mdch_fov_create(os.fsencode(camera_txt))

    def close(self):
        if getattr(self, "_h", None):
            self._L.mdch_fov_destroy(self._h)
            self._h = None

    __del__ = close

    def is_valid(self):
        return bool(self._L.mdch_fov_valid(self._h))

    def has_gpu(self):
        return bool(self._L.mdch_fov_has_gpu(self._h))

    def dims(self):
        d = np.zeros(4, dtype=np.int32)
        self._L.mdch_fov_dims(self._h, _np_ptr(d))
        return tuple(int(x) for x in d)

    def intrinsics(self):
        o = np.zeros(29, dtype=np.float32)
        self._L.mdch_fov_intrinsics(self._h, _np_ptr(o))
        return {"K_rect": o[0:9].reshape(3, 3).copy(), "K_org": o[9:18].reshape(3, 3).copy(),
                "original": o[18:23].copy(), "omega": float(o[23]), "out_calib": o[24:29].copy()}

    def remap(self):
        _, _, ow, oh = self.dims()
        if not self.is_valid():
            return None
        rx = np.zeros(ow * oh, dtype=np.float32)
        ry = np.zeros(ow * oh, dtype=np.float32)
        if not self._L.mdch_fov_remap(self._h, _np_ptr(rx), _np_ptr(ry)):
            return None
        return rx, ry

    def model(self):
        m = FovModel()
        self._L.mdch_fov_model(self._h, C.byref(m))
        return m

    def distort_coordinates(self, x, y):
        assert x.dtype == np.float32 and y.dtype == np.float32 and x.size == y.size
        self._L.mdch_fov_distort(self._h, _np_ptr(x), _np_ptr(y), x.size)

    def undistort(self, img, out):
        fn = self._L.mdch_fov_undistort_f32 if img.dtype == np.float32 else self._L.mdch_fov_undistort_u8
        fn(self._h, _np_ptr(img), _np_ptr(out), img.size, out.size)


class PhotometricUndistorter:
    """Handle on the C++ class PhotometricUndistorter (include/mono_dataset_code/PhotometricUndistorter.h)."""

    def __init__(self, pcalib_txt, vignette_image, w, h):
        self._L = host_lib()
        self.w, self.h = w, h
        self._h = self._L.mdch_photo_create(os.fsencode(pcalib_txt), os.fsencode(vignette_image), w, h)

    def close(self):
        if getattr(self, "_h", None):
            self._L.mdch_photo_destroy(self._h)
            self._h = None

    __del__ = close

    def valid(self):
        return self._L.mdch_photo_valid(self._h)

    def has_gpu(self):
        return bool(self._L.mdch_photo_has_gpu(self._h))

    def ginv(self):
        o = np.zeros(256, dtype=np.float32)
        return o if self._L.mdch_photo_ginv(self._h, _np_ptr(o)) else None

    def g(self):
        o = np.zeros(256, dtype=np.float32)
        return o if self._L.mdch_photo_g(self._h, _np_ptr(o)) else None

    def vignette(self):
        m = np.zeros(self.w * self.h, dtype=np.float32)
        i = np.zeros(self.w * self.h, dtype=np.float32)
        return (m, i) if self._L.mdch_photo_vignette(self._h, _np_ptr(m), _np_ptr(i)) else None

    def unmap(self, img_u8, out_f32, g, v, o):
        self._L.mdch_photo_unmap(self._h, _np_ptr(img_u8), _np_ptr(out_f32), img_u8.size, int(g), int(v), int(o))


def decode_gray8(data):
    """The reader's frame decoders (8-bit gray PNG, PGM P5, baseline JPEG) on a byte string -> (h, w) uint8 array;
    raises ValueError with the decoder's message."""
    L = host_lib()
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    wh = np.zeros(2, np.int32)
    err = C.create_string_buffer(256)
    out = np.zeros(1, np.uint8)
    if not L.mdch_decode_gray8(_np_ptr(buf), buf.size, _np_ptr(out), 0, _np_ptr(wh), err, 256) and (wh[0] <= 0 or wh[1] <= 0):
        raise ValueError(err.value.decode())
    out = np.zeros(int(wh[0]) * int(wh[1]), np.uint8)
    if not L.mdch_decode_gray8(_np_ptr(buf), buf.size, _np_ptr(out), out.size, _np_ptr(wh), err, 256):
        raise ValueError(err.value.decode())
    return out.reshape(int(wh[1]), int(wh[0]))


def jpeg_record_bytes(w, h):
    """-> (record bytes, pitch in blocks, block rows) that fit every sampling layout of a w x h JPEG."""
    pr = np.zeros(2, np.int32)
    n = host_lib().mdch_jpeg_record_bytes(w, h, _np_ptr(pr))
    return int(n), int(pr[0]), int(pr[1])


def decode_jpeg_record(data, record, pitch_blocks):
    """Huffman-decodes a JPEG byte string into `record` (numpy uint8, jpeg_record_bytes long; page-locked for the GPU stage):
    quantisation table + quantised luma coefficients, no inverse DCT.  -> (w, h, pitch_blocks, block rows); ValueError on failure."""
    L = host_lib()
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    dims = np.zeros(4, np.int32)
    err = C.create_string_buffer(256)
    if not L.mdch_decode_jpeg_record(_np_ptr(buf), buf.size, _np_ptr(record), record.size, pitch_blocks, _np_ptr(dims), err, 256):
        raise ValueError(err.value.decode())
    return tuple(int(x) for x in dims)


JPEG_STREAM_HEADER_BYTES = 24736


def jpeg_stream(data, stream):
    """Markers parsed, decode tables built, entropy-coded segment unstuffed into `stream` (numpy uint8; page-locked for the GPU
    stage) -> (bytes used, w, h); ValueError for files the device Huffman decoder does not take."""
    L = host_lib()
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    wh = np.zeros(2, np.int32)
    err = C.create_string_buffer(256)
    used = L.mdch_jpeg_stream(_np_ptr(buf), buf.size, _np_ptr(stream), stream.size, _np_ptr(wh), err, 256)
    if not used:
        raise ValueError(err.value.decode())
    return int(used), int(wh[0]), int(wh[1])


class DatasetReader:
    """class DatasetReader (include/mono_dataset_code/BenchmarkDatasetReader.h) through the C facade."""

    def __init__(self, folder):
        if not folder.endswith("/"):
            folder += "/"
        self._L = host_lib()
        self._h = self._L.mdch_reader_create(os.fsencode(folder))
        d = np.zeros(4, np.int32)
        self._L.mdch_reader_dims(self._h, _np_ptr(d))
        self.in_w, self.in_h, self.out_w, self.out_h = (int(x) for x in d)

    def close(self):
        if getattr(self, "_h", None):
            self._L.mdch_reader_destroy(self._h)
            self._h = None

    __del__ = close

    def __len__(self):
        return self._L.mdch_reader_num_images(self._h)

    def timestamp(self, i):
        return self._L.mdch_reader_timestamp(self._h, i)

    def exposure(self, i):
        return self._L.mdch_reader_exposure(self._h, i)

    def last_error(self):
        return self._L.mdch_reader_last_error(self._h).decode()

    def prefetch_stats(self):
        hm = np.zeros(2, np.int64)
        self._L.mdch_reader_prefetch_stats(self._h, _np_ptr(hm))
        return int(hm[0]), int(hm[1])

    def device_stats(self):
        """Per device the reader deals getImages chunks to (MDC_DEVICES): (device ordinal, frames produced, seconds waiting for
        the decoders, seconds inside the GPU calls), over the reader's life."""
        res = []
        for lane in range(64):
            idf = np.zeros(2, np.int64)
            t = np.zeros(2, np.float64)
            if not self._L.mdch_reader_device_stats(self._h, lane, _np_ptr(idf), _np_ptr(t)):
                break
            res.append((int(idf[0]), int(idf[1]), float(t[0]), float(t[1])))
        return res

    def set_threads(self, n):
        self._L.mdch_reader_set_threads(self._h, n)

    def set_prefetch(self, n):
        self._L.mdch_reader_set_prefetch(self._h, n)

    def set_lookahead(self, frames):
        self._L.mdch_reader_set_lookahead(self._h, int(frames))

    def set_gpu_jpeg(self, stage):
        """True / 2: Huffman decoding + inverse DCT on the GPU; 1: inverse DCT only; False / 0: JPEG decoded on the host."""
        self._L.mdch_reader_set_gpu_jpeg(self._h, (2 if stage else 0) if isinstance(stage, bool) else int(stage))

    def get_image(self, i, rectify, g, v, o):
        """-> (image (h, w) float32, timestamp, exposure, id) or None (getImage returned 0)."""
        n = max(self.in_w * self.in_h, self.out_w * self.out_h)
        out = np.empty(n, np.float32)
        meta = np.zeros(3, np.int32)
        ts, ex = C.c_double(0), C.c_float(0)
        if not self._L.mdch_reader_get_image(self._h, i, int(rectify), int(g), int(v), int(o), _np_ptr(out), n, _np_ptr(meta),
                                             C.byref(ts), C.byref(ex)):
            return None
        w, h = int(meta[0]), int(meta[1])
        return out[: w * h].reshape(h, w).copy(), ts.value, ex.value, int(meta[2])

    def get_images(self, first, count, rectify, g, v, o):
        """-> (images (count, h*w) float32, ok mask, number produced)."""
        n = self.out_w * self.out_h if rectify else self.in_w * self.in_h
        out = np.zeros((count, n), np.float32)
        ok = np.zeros(count, np.uint8)
        got = self._L.mdch_reader_get_images(self._h, first, count, int(rectify), int(g), int(v), int(o), _np_ptr(out), n, _np_ptr(ok))
        return out, ok.astype(bool), got

    def get_images_device(self, first, count, rectify, g, v, o, outputs):
        """getImagesDevice: results into the device arrays of `outputs` (DeviceOutputs; frame first + i at position i) -> (valid mask, number produced)."""
        valid = np.zeros(count, np.uint8)
        got = self._L.mdch_reader_get_images_device(self._h, first, count, int(rectify), int(g), int(v), int(o), C.byref(outputs), _np_ptr(valid))
        return valid.astype(bool), got

    def device(self):
        return int(self._L.mdch_reader_device(self._h))

    def get_raw(self, i):
        out = np.empty(self.in_w * self.in_h * 4 + 16, np.uint8)
        wh = np.zeros(2, np.int32)
        if not self._L.mdch_reader_get_raw(self._h, i, _np_ptr(out), out.size, _np_ptr(wh)):
            return None
        return out[: int(wh[0]) * int(wh[1])].reshape(int(wh[1]), int(wh[0])).copy()
