"""ctypes mirrors of the PODs in include/pcs_hip.h, plus the reference's calibration constants.

Field order follows include/pcs_hip.h exactly (which in turn follows rs2_intrinsics /
rs2_extrinsics so a librealsense binding can memcpy).
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Optional, Sequence

import numpy as np

POINT_SHORTS = 5          # src/pcs-camera-optimized.cpp:581-585 — x y z (R|G<<8) B
POINT_BYTES = 10
HEADER_SHORTS = 2         # src/pcs-camera-optimized.cpp:690 — payload at buffer + 2 shorts
REF_BUF_SIZE = 5_000_000  # src/pcs-camera-optimized.cpp:27
MAX_STREAMS = 64

FLAG_CUTOFF = 0x1
FLAG_CUTOFF_COMPAT = 0x2
FLAG_DROP_INVALID = 0x4
FLAG_FORCE_IEEE = 0x8
FLAG_TEXCOORD_HALF_PIXEL = 0x10     # u = (px + 0.5)/W as older librealsense releases

DISTORTION_NONE = 0
DISTORTION_MODIFIED_BROWN_CONRADY = 1
DISTORTION_INVERSE_BROWN_CONRADY = 2
DISTORTION_FTHETA = 3
DISTORTION_BROWN_CONRADY = 4

STATUS_NAMES = {
    0: "PCS_OK", -1: "PCS_ERR_INVALID_ARG", -2: "PCS_ERR_NO_DEVICE", -3: "PCS_ERR_HIP",
    -4: "PCS_ERR_UNSUPPORTED", -5: "PCS_ERR_CAPACITY", -6: "PCS_ERR_NOMEM",
}


class Intrinsics(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32),
                ("ppx", C.c_float), ("ppy", C.c_float),
                ("fx", C.c_float), ("fy", C.c_float),
                ("model", C.c_int32), ("coeffs", C.c_float * 5)]


class Extrinsics(C.Structure):
    _fields_ = [("rotation", C.c_float * 9), ("translation", C.c_float * 3)]


class StreamConfig(C.Structure):
    _fields_ = [("depth", Intrinsics), ("color", Intrinsics),
                ("depth_to_color", Extrinsics),
                ("depth_scale", C.c_float),
                ("color_bpp", C.c_int32), ("color_stride", C.c_int32),
                ("cam_to_world", C.c_float * 16)]

    @property
    def n_points(self) -> int:
        return int(self.depth.width) * int(self.depth.height)

    @property
    def color_bytes(self) -> int:
        return int(self.color_stride) * int(self.color.height)


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("n_streams", C.c_int32),
                ("streams", C.POINTER(StreamConfig)),
                ("flags", C.c_uint32), ("downsample", C.c_int32)]


class CloudDesc(C.Structure):
    """pcs_cloud_desc: one camera's rs2::points arrays (device pointers) for the batched a2 twin."""
    _fields_ = [("stream", C.c_int32), ("n_points", C.c_int32),
                ("vertices", C.c_void_p), ("texcoords", C.c_void_p),
                ("color", C.c_void_p), ("pc_buffer", C.c_void_p)]


class PayloadDesc(C.Structure):
    """pcs_payload_desc: one camera's packed records (device pointer) and the 4x4 the centre moves them by."""
    _fields_ = [("d_payload", C.c_void_p), ("n_points", C.c_int32), ("transform", C.c_float * 16)]


# --- the reference's surveyed extrinsics (data, not code) -------------------------------------
# src/pcs-camera-optimized.cpp:64-67
TF_MAT = np.array([
    -0.99977970,  0.00926272,  0.01883480,  0.00000000,
    -0.01638983,  0.21604544, -0.97624574,  3.41600000,
    -0.01311186, -0.97633937, -0.21584603,  1.80200000,
     0.00000000,  0.00000000,  0.00000000,  1.00000000], dtype=np.float32)

# src/pcs-multicamera-optimized.cpp:417-455 — transform[0..7]
TRANSFORMS = np.array([
    [-0.69888007, -0.32213748,  0.63858757, -2.22900000,
     -0.71520905,  0.32290986, -0.61984291,  2.91800000,
     -0.00653159, -0.88991947, -0.45607091,  0.36400000,
      0.0, 0.0, 0.0, 1.0],
    [-0.96127595,  0.09045863, -0.26031862,  0.31700000,
      0.27558764,  0.31552831, -0.90801615,  2.83300000,
      0.00000000, -0.94459469, -0.32823906,  0.38100000,
      0.0, 0.0, 0.0, 1.0],
    [-0.63305575,  0.28270490, -0.72063747,  2.80300000,
      0.77409926,  0.22724638, -0.59087175,  2.05500000,
     -0.00328008, -0.93189968, -0.36270128,  0.42100000,
      0.0, 0.0, 0.0, 1.0],
    [ 0.17021299,  0.28598815, -0.94299433,  2.51000000,
      0.98527137, -0.03349883,  0.16768470, -0.27300000,
      0.01636663, -0.95764743, -0.28747787,  0.35900000,
      0.0, 0.0, 0.0, 1.0],
    [ 0.72625904,  0.26139935, -0.63578155,  1.90900000,
      0.68735231, -0.26305364,  0.67701520, -2.81700000,
      0.00972668, -0.92869433, -0.37071853,  0.37900000,
      0.0, 0.0, 0.0, 1.0],
    [ 0.98744750,  0.00686296,  0.15779838, -0.57400000,
     -0.14665062, -0.33120318,  0.93209337, -2.69700000,
      0.05866025, -0.94353450, -0.32603930,  0.30900000,
      0.0, 0.0, 0.0, 1.0],
    [ 0.67295609,  0.40193638,  0.62094867, -2.97300000,
     -0.35777412, -0.55787451,  0.74884826, -0.41700000,
      0.64740079, -0.72610136, -0.23162261,  0.43400000,
      0.0, 0.0, 0.0, 1.0],
    [ 0.08929624, -0.21535297,  0.97244500, -2.95700000,
     -0.67610010, -0.73004840, -0.09958907, -0.33900000,
      0.73137872, -0.64857723, -0.21079074,  0.33800000,
      0.0, 0.0, 0.0, 1.0],
], dtype=np.float32)


def make_intrinsics(width: int, height: int, fx: float, fy: float, ppx: float, ppy: float,
                    model: int = DISTORTION_NONE, coeffs: Optional[Sequence[float]] = None) -> Intrinsics:
    it = Intrinsics()
    it.width, it.height = int(width), int(height)
    it.fx, it.fy, it.ppx, it.ppy = float(fx), float(fy), float(ppx), float(ppy)
    it.model = int(model)
    cs = list(coeffs) if coeffs is not None else [0.0] * 5
    if len(cs) != 5:
        raise ValueError("coeffs must have 5 entries (k1 k2 p1 p2 k3)")
    for k in range(5):
        it.coeffs[k] = float(cs[k])
    return it


def make_stream_config(depth: Intrinsics, color: Optional[Intrinsics] = None, *,
                       cam_to_world: Optional[Iterable[float]] = None,
                       rotation: Optional[Iterable[float]] = None,
                       translation: Iterable[float] = (0.015, 0.0, 0.0),
                       depth_scale: float = 0.001,
                       color_bpp: int = 3, color_stride: Optional[int] = None) -> StreamConfig:
    """Build one camera stream's configuration.

    Defaults follow SURVEY.md §8(d): colour at the depth resolution, depth->colour R = I,
    t = (0.015, 0, 0) m, depth scale 0.001, RGB8 with stride = 3*W, extrinsic = tf_mat.
    """
    sc = StreamConfig()
    sc.depth = depth
    sc.color = color if color is not None else depth
    rot = list(rotation) if rotation is not None else [1, 0, 0, 0, 1, 0, 0, 0, 1]
    tr = list(translation)
    if len(rot) != 9 or len(tr) != 3:
        raise ValueError("rotation needs 9 (column-major) and translation 3 entries")
    for k in range(9):
        sc.depth_to_color.rotation[k] = float(rot[k])
    for k in range(3):
        sc.depth_to_color.translation[k] = float(tr[k])
    sc.depth_scale = float(depth_scale)
    sc.color_bpp = int(color_bpp)
    sc.color_stride = int(color_stride) if color_stride is not None else int(color_bpp) * int(sc.color.width)
    m = np.asarray(TF_MAT if cam_to_world is None else cam_to_world, dtype=np.float32).reshape(-1)
    if m.size != 16:
        raise ValueError("cam_to_world must have 16 entries (row-major 4x4)")
    for k in range(16):
        sc.cam_to_world[k] = float(m[k])
    return sc


def stream_array(configs: Sequence[StreamConfig]):
    arr = (StreamConfig * len(configs))()
    for i, c in enumerate(configs):
        C.memmove(C.byref(arr, i * C.sizeof(StreamConfig)), C.byref(c), C.sizeof(StreamConfig))
    return arr
