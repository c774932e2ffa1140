"""Loader for libpcs_hip.so — the C-ABI product library (include/pcs_hip.h).

There is no Python or CPU fallback for the compute path: if the shared library cannot be built or
loaded this module raises, loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

from .types import CloudDesc, Config, PayloadDesc, StreamConfig

_PKG = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.join(_PKG, "csrc")
LIB_PATH = os.path.join(_PKG, "lib", "libpcs_hip.so")
INCLUDE_DIR = os.path.join(os.path.dirname(_PKG), "include")

# every symbol include/pcs_hip.h declares: (name, restype, argtypes)
_P = C.POINTER
_VP = C.c_void_p
SYMBOLS = [
    ("pcs_abi_version", C.c_int, []),
    ("pcs_device_count", C.c_int, []),
    ("pcs_create", C.c_int, [_P(_VP), _P(Config)]),
    ("pcs_destroy", None, [_VP]),
    ("pcs_strerror", C.c_char_p, [C.c_int]),
    ("pcs_last_error", C.c_char_p, [_VP]),
    ("pcs_set_cam_to_world", C.c_int, [_VP, C.c_int, _P(C.c_float)]),
    ("pcs_stream_points", C.c_int, [_VP, C.c_int]),
    ("pcs_stream_math", C.c_int, [_VP, C.c_int]),
    ("pcs_stream_color_row_const", C.c_int, [_VP, C.c_int]),
    ("pcs_max_payload_shorts", C.c_size_t, [_VP]),
    ("pcs_copy_pointcloud_xyzrgb_to_buffer", C.c_int,
     [_VP, C.c_int, _VP, _VP, C.c_int, _VP, _VP, _P(C.c_int)]),
    ("pcs_copy_pointcloud_xyzrgb_to_buffer_device", C.c_int,
     [_VP, C.c_int, _VP, _VP, C.c_int, _VP, _VP, _VP]),
    ("pcs_copy_pointclouds_xyzrgb_to_buffer_device", C.c_int, [_VP, C.c_int, _P(CloudDesc), _VP]),
    ("pcs_send_xyzrgb_pointcloud", C.c_int,
     [_VP, C.c_int, _VP, _VP, C.c_int, _VP, _VP, C.c_size_t, C.c_int, _P(C.c_int)]),
    ("pcs_process_frames", C.c_int,
     [_VP, _P(_VP), _P(_VP), _VP, C.c_size_t, C.c_int, _P(C.c_int), _P(C.c_int)]),
    ("pcs_process_frames_device", C.c_int, [_VP, _P(_VP), _P(_VP), _VP, C.c_size_t, _VP]),
    ("pcs_process_frames_device_counted", C.c_int, [_VP, _P(_VP), _P(_VP), _VP, _VP, C.c_size_t, _VP]),
    ("pcs_stream_tile_base", C.c_int, [_VP, C.c_int]),
    ("pcs_process_frames_device_batch", C.c_int, [_VP, C.c_int, _P(_VP), _P(_VP), _P(_VP), C.c_size_t, _P(_VP)]),
    ("pcs_submit_frames", C.c_int, [_VP, _P(_VP), _P(_VP), _P(C.c_int)]),
    ("pcs_collect_frames", C.c_int, [_VP, C.c_int, _VP, C.c_size_t, C.c_int, _P(C.c_int), _P(C.c_int)]),
    ("pcs_deproject", C.c_int, [_VP, C.c_int, _VP, _VP, _VP]),
    ("pcs_stitch_device", C.c_int, [_VP, _P(_VP), _P(C.c_int), C.c_int, C.c_int, _VP, C.c_size_t, _P(C.c_int)]),
    ("pcs_transform_payloads_device", C.c_int, [_VP, C.c_int, _P(PayloadDesc), C.c_int, _VP, C.c_size_t, _P(C.c_int), _P(C.c_int)]),
    ("pcs_set_voxel_tail", C.c_int, [_VP, C.c_int]),
    ("pcs_voxel_tail_reruns", C.c_int, [_VP]),
    ("pcs_inject_voxel_stall", C.c_int, [C.c_int]),
    ("pcs_voxel_grid_device", C.c_int, [_VP, _VP, C.c_int, C.c_int, _VP, C.c_size_t, _VP]),
    ("pcs_voxel_grid_device_counted", C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int, _VP, C.c_size_t, _VP]),
    ("pcs_process_frames_voxel_device", C.c_int, [_VP, _P(_VP), _P(_VP), C.c_int, _VP, C.c_size_t, _VP]),
    ("pcs_voxel_grid", C.c_int, [_VP, _VP, C.c_int, C.c_int, _VP, C.c_size_t, _P(C.c_int)]),
    ("pcs_process_frames_voxel_partials_device", C.c_int, [_VP, _P(_VP), _P(_VP), C.c_int, _VP, _VP, C.c_size_t, _VP]),
    ("pcs_voxel_grid_from_partials_device", C.c_int, [_VP, _VP, _VP, C.c_int, _VP, C.c_int, _VP, C.c_size_t, _VP]),
    ("pcs_voxel_sink_begin", C.c_int, [_VP, C.c_size_t, C.c_int, _VP]),
    ("pcs_process_frames_voxel_into_sink_device", C.c_int, [_VP, _P(_VP), _P(_VP), _VP]),
    ("pcs_voxel_sink_finish", C.c_int, [_VP, _VP, _VP, C.c_size_t, _VP]),
    ("pcs_set_stream", C.c_int, [_VP, _VP]),
    ("pcs_get_stream", _VP, [_VP]),
    ("pcs_synchronize", C.c_int, [_VP]),
    ("pcs_use_stream_beside", C.c_int, [_VP, _VP]),
    ("pcs_pick_concurrent_stream", C.c_int, [_VP, _P(_VP)]),
    ("pcs_timer_begin", C.c_int, [_VP]),
    ("pcs_timer_end", C.c_int, [_VP]),
    ("pcs_timer_elapsed_ms", C.c_int, [_VP, _P(C.c_float)]),
    ("pcs_kernel_timing", C.c_int, [_VP, C.c_int]),
    ("pcs_kernel_times_ms", C.c_int, [_VP, _P(C.c_float), C.c_int, _P(C.c_int)]),
    ("pcs_host_malloc", C.c_int, [_VP, _P(_VP), C.c_size_t]),
    ("pcs_host_free", C.c_int, [_VP, _VP]),
    ("pcs_host_register", C.c_int, [_VP, _VP, C.c_size_t]),
    ("pcs_host_unregister", C.c_int, [_VP, _VP]),
    ("pcs_device_malloc", C.c_int, [_VP, _P(_VP), C.c_size_t]),
    ("pcs_device_free", C.c_int, [_VP, _VP]),
    ("pcs_memcpy_h2d", C.c_int, [_VP, _VP, _VP, C.c_size_t]),
    ("pcs_memcpy_d2h", C.c_int, [_VP, _VP, _VP, C.c_size_t]),
]

_lib: Optional[C.CDLL] = None


class PcsBuildError(RuntimeError):
    pass


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile libpcs_hip.so and libpcs_node.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
    Always defers to make — a no-op when both libraries are newer than every source, header and the Makefile — so a
    library left over from before a source change cannot be loaded; a lock keeps concurrent processes from building at once.
    An install that cannot run make at all (read-only tree: no lock file; a box without make) loads the library it ships, with
    a warning that it could not be checked against the sources; with no library there either the error is PcsBuildError."""
    import fcntl
    import warnings
    try:
        os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
        with open(os.path.join(os.path.dirname(LIB_PATH), ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            cmd = ["make", "-C", CSRC_DIR] + (["-B"] if force else [])
            proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    except OSError as e:          # PermissionError (read-only install), FileNotFoundError (no make)
        if os.path.exists(LIB_PATH) and not force:
            warnings.warn(f"pointcloud_stitching_amd: cannot run make here ({e}); loading {LIB_PATH} as shipped, unchecked against "
                          "the sources", RuntimeWarning)
            return LIB_PATH
        raise PcsBuildError(f"cannot build libpcs_hip.so: {e} — the HIP extension is required, there is no fallback") from e
    if verbose:
        print(proc.stdout)
    if proc.returncode != 0 or not os.path.exists(LIB_PATH):
        raise PcsBuildError("building libpcs_hip.so failed (hipcc --offload-arch=gfx950):\n" + proc.stdout)
    return LIB_PATH


def load() -> C.CDLL:
    """Load libpcs_hip.so (make runs first: it rebuilds a missing or stale library). Raises if that is impossible."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("PCS_LIB_PATH")      # lab: a variant build of the same ABI (tools/lab/variants); never a fallback
    if not path:
        build()
        path = LIB_PATH
    try:
        lib = C.CDLL(path)
    except OSError as e:   # pragma: no cover - depends on the box
        raise PcsBuildError(f"cannot load {LIB_PATH}: {e} — the HIP extension is required, there is no fallback") from e
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)   # AttributeError if the header and the .so ever disagree
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib
