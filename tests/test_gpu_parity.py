"""Parity proper: the HIP path, through the C ABI, against the CPU oracle on the same seeded inputs.
Bit-exact for the int16 records; bit-exact (NaN == NaN) for the float deprojection, which is stricter
than the 1-ulp tolerance DESIGN.md allows against a real librealsense."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from pointcloud_stitching_amd import synthetic as S
from pointcloud_stitching_amd.api import PcsContext, PcsError
from pointcloud_stitching_amd.types import (FLAG_CUTOFF, FLAG_CUTOFF_COMPAT, FLAG_DROP_INVALID, FLAG_FORCE_IEEE,
                                            FLAG_TEXCOORD_HALF_PIXEL, HEADER_SHORTS,
                                            POINT_SHORTS, TRANSFORMS, make_intrinsics, make_stream_config)

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def first_diff(a, b):
    a = np.asarray(a); b = np.asarray(b)
    if a.shape != b.shape:
        return f"shape {a.shape} vs {b.shape}"
    bad = np.argwhere(a != b)
    if bad.size == 0:
        return None
    i = tuple(bad[0])
    return f"{bad.shape[0]} mismatches; first at {i}: got {a[i]} want {b[i]}; row got {a[i[0]]} want {b[i[0]]}"


def assert_same(got, want):
    d = first_diff(got, want)
    assert d is None, d


def random_points(n, seed, cw=64, ch=48, spread=8.0):
    rng = np.random.default_rng(seed)
    V = (rng.standard_normal((n, 3)) * spread).astype(np.float32)
    T = rng.uniform(-0.2, 1.2, (n, 2)).astype(np.float32)
    col = rng.integers(0, 256, cw * ch * 3, dtype=np.uint8)
    it = make_intrinsics(cw, ch, 40, 40, cw / 2, ch / 2)
    return make_stream_config(it, it), V, T, col


# ---------------------------------------------------------------------------------------------
# a2 twin: copyPointCloudXYZRGBToBufferSIMD
# ---------------------------------------------------------------------------------------------
def test_pack_kat_appendix_b():
    with open(os.path.join(GOLD, "kat_appendix_b.json")) as f:
        k = json.load(f)
    V = np.array([v["vertex"] for v in k["vectors"]], np.float32)
    T = np.array([v["uv"] for v in k["vectors"]], np.float32)
    B = np.array([[int(x, 16) for x in v["bytes"].split()] for v in k["vectors"]], np.uint8)
    it = make_intrinsics(8, 4, 1, 1, 0, 0)
    sc = make_stream_config(it, it, color_stride=24)
    col = ((7 * np.arange(96) + 3) & 0xFF).astype(np.uint8)
    with PcsContext([sc]) as ctx:
        out, cnt = ctx.copy_pointcloud_xyzrgb_to_buffer(0, V, T, col)
    assert cnt == 8
    assert_same(out.view(np.uint8).reshape(8, 10), B)


@pytest.mark.parametrize("n", [0, 1, 3, 4, 7, 8, 9, 64, 2047, 2048, 2049, 5000, 307200])
def test_pack_random_sizes(oracle, n):
    sc, V, T, col = random_points(n, 1000 + n)
    with PcsContext([sc]) as ctx:
        out, cnt = ctx.copy_pointcloud_xyzrgb_to_buffer(0, V, T, col)
    assert cnt == n
    assert_same(out, oracle.pack(sc, V, T, col))


def test_pack_edge_values(oracle):
    sc, V, T, col = random_points(4096, 77, spread=30.0)
    edge_uv = [(-1, -1), (0, 0), (1, 1), (2, 2), (0.4999, 0.5), (0.5, 0.4999), (1e9, -1e9), (np.nan, np.inf),
               (-np.inf, np.nan), (3e38, 3e38), (0.9999999, 0.9999999), (1e-45, -1e-45), (-0.0, 0.0),
               (0.0078125, 0.0104166), (33554432.0, -33554432.0), (16777216.0, 0.5)]
    T[:len(edge_uv)] = edge_uv
    edge_v = [(40, 0, 0), (-40, 50, 60), (1e7, 0, 0), (np.nan, 0, 0), (np.inf, 1, 1), (-np.inf, 0, 0),
              (3e38, 3e38, 3e38), (2147483.648, 0, 0), (-2147483.648, 0, 0), (1e-45, 1e-45, 1e-45),
              (-0.0, -0.0, -0.0), (32.767, 32.768, -32.769), (65.535, 65.536, 65.537), (2.2e6, -2.2e6, 0)]
    V[100:100 + len(edge_v)] = edge_v
    with PcsContext([sc]) as ctx:
        out, _ = ctx.copy_pointcloud_xyzrgb_to_buffer(0, V, T, col)
    assert_same(out, oracle.pack(sc, V, T, col))


def test_pack_rgba_stride_padding(oracle):
    rng = np.random.default_rng(3)
    cw, ch, bpp, stride = 50, 20, 4, 224
    it = make_intrinsics(cw, ch, 30, 30, 25, 10)
    sc = make_stream_config(it, it, color_bpp=bpp, color_stride=stride)
    col = rng.integers(0, 256, stride * ch, dtype=np.uint8)
    V = rng.standard_normal((3000, 3)).astype(np.float32)
    T = rng.uniform(-0.1, 1.1, (3000, 2)).astype(np.float32)
    with PcsContext([sc]) as ctx:
        out, _ = ctx.copy_pointcloud_xyzrgb_to_buffer(0, V, T, col)
    assert_same(out, oracle.pack(sc, V, T, col))


def test_pack_last_pixel_reads_inside_raster(oracle):
    # the colour dword window must slide back at the very last pixel (bpp 3, no padding)
    it = make_intrinsics(5, 3, 1, 1, 0, 0)
    sc = make_stream_config(it, it)
    col = np.arange(45, dtype=np.uint8) + 100
    T = np.array([(1, 1), (0.99, 0.99), (0.7, 1.0), (0, 0)], np.float32)
    V = np.zeros((4, 3), np.float32)
    with PcsContext([sc]) as ctx:
        out, _ = ctx.copy_pointcloud_xyzrgb_to_buffer(0, V, T, col)
    assert_same(out, oracle.pack(sc, V, T, col))
    assert out[0, 3].view(np.uint16) == (142 | 143 << 8) and out[0, 4] == 144


@pytest.mark.parametrize("flags", [FLAG_CUTOFF, FLAG_CUTOFF | FLAG_CUTOFF_COMPAT, FLAG_DROP_INVALID,
                                   FLAG_CUTOFF | FLAG_DROP_INVALID])
@pytest.mark.parametrize("n", [5, 2048, 10007])
def test_pack_compaction_matches_t1_order(oracle, flags, n):
    sc, V, T, col = random_points(n, 31 + n, spread=1.5)
    V[:, 2] = np.abs(V[:, 2])
    V[::5, 2] = 0.0
    want = oracle.pack(sc, V, T, col, flags)
    with PcsContext([sc], flags=flags) as ctx:
        out, cnt = ctx.copy_pointcloud_xyzrgb_to_buffer(0, V, T, col)
    assert cnt == want.shape[0]
    assert_same(out, want)


def test_send_xyzrgb_pointcloud_layout(oracle):
    sc, V, T, col = random_points(1000, 5)
    want, wsize = oracle.send_xyzrgb_pointcloud(sc, V, T, col, buffer_shorts=2_600_000, write_header=True)
    buf = np.full(2_600_000, 0x5A5A, np.uint16).view(np.int16)
    with PcsContext([sc]) as ctx:
        size = ctx.send_xyzrgb_pointcloud(0, V, T, col, buf, write_header=True)
        assert size == wsize == 10000
        assert_same(buf, want)
        buf2 = np.full(2_600_000, 0x5A5A, np.uint16).view(np.int16)
        ctx.send_xyzrgb_pointcloud(0, V, T, col, buf2, write_header=False)
        want2, _ = oracle.send_xyzrgb_pointcloud(sc, V, T, col, buffer_shorts=2_600_000, write_header=False)
        assert_same(buf2, want2)
        small = np.zeros(100, np.int16)
        with pytest.raises(PcsError) as e:
            ctx.send_xyzrgb_pointcloud(0, V, T, col, small)
        assert e.value.status == -5


# ---------------------------------------------------------------------------------------------
# a5: deprojection contract
# ---------------------------------------------------------------------------------------------
def same_floats(a, b):
    a = a.view(np.uint32); b = b.view(np.uint32)
    nan_a = (a & 0x7FFFFFFF) > 0x7F800000
    nan_b = (b & 0x7FFFFFFF) > 0x7F800000
    return ((a == b) | (nan_a & nan_b)).all()


@pytest.mark.parametrize("mode", ["scene", "random"])
def test_deproject_bit_exact(oracle, mode):
    cfgs, depth, _ = S.synth_frame_set(1, 640, 480, single=True, mode=mode)
    with PcsContext(cfgs) as ctx:
        v, t = ctx.deproject(0, depth[0])
    v0, t0 = oracle.deproject(cfgs[0], depth[0])
    assert same_floats(v, v0) and same_floats(t, t0)


def distorted_config(w, h, cw, ch):
    di = make_intrinsics(w, h, 0.7 * w, 0.71 * w, w / 2 + 1.3, h / 2 - 2.2, model=2,
                         coeffs=[0.08, -0.03, 0.001, -0.002, 0.01])
    ci = make_intrinsics(cw, ch, 0.72 * cw, 0.72 * cw, cw / 2 - 3.1, ch / 2 + 1.7, model=1,
                         coeffs=[-0.05, 0.06, 0.0005, -0.0007, -0.02])
    a = 0.02
    rot = [np.cos(a), np.sin(a), 0, -np.sin(a), np.cos(a), 0, 0, 0, 1]     # column-major small roll
    return make_stream_config(di, ci, cam_to_world=TRANSFORMS[3], rotation=rot, translation=(0.0147, 0.0003, -0.0002))


def test_deproject_with_distortion_and_rotation(oracle):
    sc = distorted_config(128, 96, 192, 108)
    depth = S.synth_depth(128, 96, 2)
    with PcsContext([sc]) as ctx:
        v, t = ctx.deproject(0, depth)
    v0, t0 = oracle.deproject(sc, depth)
    assert same_floats(v, v0) and same_floats(t, t0)


# ---------------------------------------------------------------------------------------------
# fused a5 + a2 (+ a7 concat)
# ---------------------------------------------------------------------------------------------
def run_fused(cfgs, depth, color, flags=0, downsample=1):
    with PcsContext(cfgs, flags=flags, downsample=downsample) as ctx:
        buf, counts, size = ctx.process_frames(depth, color, write_header=True)
    assert int.from_bytes(buf[:2].tobytes(), "little", signed=True) == size
    assert size == 10 * sum(counts)
    return buf[HEADER_SHORTS:HEADER_SHORTS + 5 * sum(counts)].reshape(-1, 5), counts


@pytest.mark.parametrize("shape", [(64, 48), (640, 480), (1280, 720)])
def test_fused_single_stream(oracle, shape):
    cfgs, depth, color = S.synth_frame_set(1, *shape, single=True)
    got, counts = run_fused(cfgs, depth, color)
    want, wcounts = oracle.process_frames(cfgs, depth, color)
    assert counts == wcounts
    assert_same(got, want)


def test_fused_random_depth_full_range(oracle):
    cfgs, depth, color = S.synth_frame_set(2, 640, 480, mode="random")
    got, counts = run_fused(cfgs, depth, color)
    want, _ = oracle.process_frames(cfgs, depth, color)
    assert_same(got, want)


def test_fused_eight_streams_720p_full_compare(oracle):
    cfgs, depth, color = S.synth_frame_set(8, 1280, 720)
    got, counts = run_fused(cfgs, depth, color)
    want, wcounts = oracle.process_frames(cfgs, depth, color)
    assert counts == wcounts == [921600] * 8
    assert_same(got, want)
    # a7: camera-order concatenation — stream s occupies [s*N, (s+1)*N)
    one, _ = oracle.process_frames(cfgs[3:4], depth[3:4], color[3:4])
    assert_same(got[3 * 921600:4 * 921600], one)


def test_fused_mixed_geometry_streams(oracle):
    # different rasters per stream, colour at another resolution, one stream with distortion,
    # one with W % 8 != 0 and N % 8 != 0 (forces the generic path for the whole set)
    cfgs = [S.synth_stream_config(640, 480, 0),
            S.synth_stream_config(1280, 720, 1, color_size=(1920, 1080)),
            distorted_config(128, 96, 192, 108),
            S.synth_stream_config(100, 37, 3)]
    depth = [S.synth_depth(c.depth.width, c.depth.height, i) for i, c in enumerate(cfgs)]
    color = [S.synth_color(c.color.width, c.color.height, i) for i, c in enumerate(cfgs)]
    got, counts = run_fused(cfgs, depth, color)
    want, wcounts = oracle.process_frames(cfgs, depth, color)
    assert counts == wcounts
    assert_same(got, want)
    # and the dense path on the first three only
    got3, _ = run_fused(cfgs[:3], depth[:3], color[:3])
    want3, _ = oracle.process_frames(cfgs[:3], depth[:3], color[:3])
    assert_same(got3, want3)


def test_fused_more_streams_than_one_launch(oracle):
    cfgs, depth, color = S.synth_frame_set(19, 64, 48)
    got, counts = run_fused(cfgs, depth, color)
    want, _ = oracle.process_frames(cfgs, depth, color)
    assert_same(got, want)


@pytest.mark.parametrize("flags", [FLAG_DROP_INVALID, FLAG_CUTOFF, FLAG_CUTOFF | FLAG_CUTOFF_COMPAT,
                                   FLAG_CUTOFF | FLAG_DROP_INVALID])
@pytest.mark.parametrize("downsample", [1, 3])
def test_fused_compaction_and_stride(oracle, flags, downsample, compaction_path):
    cfgs, depth, color = S.synth_frame_set(3, 640, 480)
    for d in depth:
        d[100:140, :] //= 4          # bring part of the scene inside the 1.5 m cutoff
    got, counts = run_fused(cfgs, depth, color, flags, downsample)
    want, wcounts = oracle.process_frames(cfgs, depth, color, flags, downsample)
    assert counts == wcounts
    assert_same(got, want)


@pytest.mark.parametrize("downsample", [2, 5, 8, 2049])
def test_fused_stride_only(oracle, downsample):
    cfgs, depth, color = S.synth_frame_set(2, 640, 480)
    got, counts = run_fused(cfgs, depth, color, 0, downsample)
    want, wcounts = oracle.process_frames(cfgs, depth, color, 0, downsample)
    assert counts == wcounts
    assert_same(got, want)


def test_fused_all_invalid_and_all_dropped(oracle):
    cfgs, depth, color = S.synth_frame_set(2, 64, 48)
    depth[0][:] = 0
    got, counts = run_fused(cfgs, depth, color, FLAG_DROP_INVALID)
    want, wcounts = oracle.process_frames(cfgs, depth, color, FLAG_DROP_INVALID)
    assert counts == wcounts and counts[0] == 0
    assert_same(got, want)
    depth[1][:] = 0
    got, counts = run_fused(cfgs, depth, color, FLAG_DROP_INVALID)
    assert counts == [0, 0] and got.shape[0] == 0


def test_fused_device_api_unaligned_payload_and_idempotence(oracle):
    # the reference's payload pointer is buffer+4 bytes (:690): not 16-byte aligned -> generic store path
    cfgs, depth, color = S.synth_frame_set(2, 640, 480)
    want, _ = oracle.process_frames(cfgs, depth, color)
    n_sh = want.size
    with PcsContext(cfgs) as ctx:
        dd = [ctx.device_malloc(d.nbytes) for d in depth]
        dc = [ctx.device_malloc(c.nbytes) for c in color]
        for p, a in zip(dd + dc, depth + color):
            ctx.memcpy_h2d(p, a)
        out = ctx.device_malloc(n_sh * 2 + 64)
        for skew in (0, 4, 2, 10):
            ctx.process_frames_device(dd, dc, out + skew, n_sh)
            ctx.process_frames_device(dd, dc, out + skew, n_sh)      # idempotent
            ctx.synchronize()
            got = np.empty(n_sh, np.int16)
            ctx.memcpy_d2h(got, out + skew)
            assert_same(got.reshape(-1, 5), want)
        with pytest.raises(PcsError) as e:
            ctx.process_frames_device(dd, dc, out, n_sh - 5)
        assert e.value.status == -5
        for p in dd + dc + [out]:
            ctx.device_free(p)


def test_set_cam_to_world(oracle):
    cfgs, depth, color = S.synth_frame_set(1, 64, 48, single=True)
    with PcsContext(cfgs) as ctx:
        ctx.set_cam_to_world(0, TRANSFORMS[5])
        buf, counts, size = ctx.process_frames(depth, color)
    for k in range(16):
        cfgs[0].cam_to_world[k] = float(TRANSFORMS[5][k])
    want, _ = oracle.process_frames(cfgs, depth, color)
    assert_same(buf[2:2 + want.size].reshape(-1, 5), want)


def test_stitch_device_with_stride(oracle):
    rng = np.random.default_rng(9)
    cams = [rng.integers(-30000, 30000, (n, 5), dtype=np.int16) for n in (5000, 0, 2049, 1)]
    cfgs, _, _ = S.synth_frame_set(1, 64, 48)
    with PcsContext(cfgs) as ctx:
        dptr = [ctx.device_malloc(max(c.nbytes, 16)) for c in cams]
        for p, c in zip(dptr, cams):
            if c.size:
                ctx.memcpy_h2d(p, c)
        out = ctx.device_malloc(sum(c.nbytes for c in cams) + 64)
        for d in (1, 2, 3, 7):
            want = oracle.stitch(cams, d)
            total = ctx.stitch_device(dptr, [c.shape[0] for c in cams], d, out + 4, want.size)
            ctx.synchronize()
            assert total == want.shape[0]
            got = np.empty(want.size, np.int16)
            ctx.memcpy_d2h(got, out + 4)
            assert_same(got.reshape(-1, 5), want)


@pytest.mark.parametrize("stride", [1, 2, 7])
def test_centre_side_transform_of_packed_payloads(oracle, stride):
    """pcs_transform_payloads_device — the decode / pcl::transformPointCloud / re-encode pcs-multicamera-optimized applies to every
    camera's payload before concatenating (src/pcs-multicamera-optimized.cpp:226-265, 289) — against the oracle's restatement:
    every int16 value per coordinate (all 65 536 quotients by 1000.0f), int16 wrap-around of the moved coordinates, negative
    truncation, NaN / inf / huge matrix entries, ragged sizes (0, 1, 2047, 2049 records), payloads at 2-, 4- and 10-byte phases of a
    16-byte line (the wire's buffer + 2 shorts), more cameras than one launch holds, and an in-place transform."""
    from pointcloud_stitching_amd.types import TRANSFORMS, TF_MAT
    rng = np.random.default_rng(11 + stride)
    allv = np.arange(-32768, 32768, dtype=np.int16)
    sweep = np.zeros((65536 * 3, 5), np.int16)
    for k in range(3):
        sweep[65536 * k:65536 * (k + 1), k] = allv
        sweep[65536 * k:65536 * (k + 1), (k + 1) % 3] = rng.integers(-32768, 32768, 65536, dtype=np.int16)
    sweep[:, 3:] = rng.integers(-32768, 32768, (sweep.shape[0], 2), dtype=np.int16)
    sizes = [sweep.shape[0], 0, 1, 2047, 2049, 5000] + [300 + 17 * i for i in range(14)]          # 20 cameras: two launches
    cams = [sweep] + [rng.integers(-32768, 32768, (n, 5), dtype=np.int16) for n in sizes[1:]]
    wild = np.array([1e6, -3e7, 2.5, 7e9, np.nan, 1, 1, 0, 0, 0, np.inf, -4, 0, 0, 0, 1], np.float32)
    mats = [TRANSFORMS[i % 8] for i in range(len(cams))]
    mats[3], mats[4], mats[5] = wild, TF_MAT, np.eye(4, dtype=np.float32).reshape(-1)
    want = np.concatenate([oracle.transform_payload(c, m, stride) for c, m in zip(cams, mats)])
    cfgs, _, _ = S.synth_frame_set(1, 64, 48)
    with PcsContext(cfgs) as ctx:
        phases = [0, 4, 2, 10, 6, 14, 8, 12]
        dptr = []
        for i, c in enumerate(cams):
            base = ctx.device_malloc(max(c.nbytes, 16) + 64)
            dptr.append(base + phases[i % len(phases)])
            if c.size:
                ctx.memcpy_h2d(dptr[-1], c)
        out = ctx.device_malloc(want.nbytes + 64)
        for out_phase in (0, 4, 10):
            per, total = ctx.transform_payloads_device(dptr, [c.shape[0] for c in cams], mats, stride, out + out_phase, want.size)
            ctx.synchronize()
            assert total == want.shape[0] and per == [c.shape[0] // stride for c in cams]      # floor: the decoded cloud's width (:230)
            got = np.empty(want.size, np.int16)
            ctx.memcpy_d2h(got, out + out_phase)
            assert_same(got.reshape(-1, 5), want)
        # capacity and overlap checks
        with pytest.raises(PcsError) as e:
            ctx.transform_payloads_device(dptr, [c.shape[0] for c in cams], mats, stride, out, want.size - 5)
        assert e.value.status == -5
        with pytest.raises(PcsError) as e:                     # camera 1's output slice would land inside camera 0's input
            ctx.transform_payloads_device([dptr[0], dptr[5]], [cams[0].shape[0], cams[5].shape[0]], mats[:2], stride, dptr[0] + 10, 10 ** 7)
        assert e.value.status == -1 and "overlaps" in str(e.value)
        if stride == 1:                                         # in place: one camera, output = input
            per, total = ctx.transform_payloads_device([dptr[0]], [cams[0].shape[0]], [mats[0]], 1, dptr[0], cams[0].size)
            ctx.synchronize()
            got = np.empty(cams[0].size, np.int16)
            ctx.memcpy_d2h(got, dptr[0])
            assert_same(got.reshape(-1, 5), oracle.transform_payload(cams[0], mats[0], 1))


def test_sixteen_streams_1080p_digest(oracle):
    # config 5 geometry: full compare through a digest to keep host memory in check
    cfgs, depth, color = S.synth_frame_set(16, 1920, 1080)
    got, counts = run_fused(cfgs, depth, color, FLAG_DROP_INVALID)
    want, wcounts = oracle.process_frames(cfgs, depth, color, FLAG_DROP_INVALID)
    assert counts == wcounts
    assert hashlib.sha256(got.tobytes()).hexdigest() == hashlib.sha256(want.tobytes()).hexdigest()


# ---------------------------------------------------------------------------------------------
# Certified reduced-instruction arithmetic (CertMath) vs the IEEE expansion vs the oracle
# ---------------------------------------------------------------------------------------------
def test_policy_selection_and_equivalence_on_the_benchmark_config(oracle):
    cfgs, depth, color = S.synth_frame_set(8, 1280, 720)
    want, _ = oracle.process_frames(cfgs, depth, color)
    with PcsContext(cfgs) as ctx:
        assert [ctx.stream_math(s) for s in range(8)] == [4] * 8      # certified + identity R + no-overflow
        buf, _, _ = ctx.process_frames(depth, color)
    assert_same(buf[2:2 + want.size].reshape(-1, 5), want)
    with PcsContext(cfgs, flags=FLAG_FORCE_IEEE) as ctx:
        assert [ctx.stream_math(s) for s in range(8)] == [0] * 8
        buf2, _, _ = ctx.process_frames(depth, color)
    assert_same(buf2[2:2 + want.size].reshape(-1, 5), want)


def test_policy_falls_back_when_not_certifiable(oracle):
    # (a) depth distortion -> rays not separable -> IEEE; (b) rotation present -> certified without the shortcut;
    # (c) colour plane could pass through the camera (t_z << 0) -> denominator not bounded away from 0 -> IEEE
    a = distorted_config(128, 96, 192, 108)
    b = S.synth_stream_config(128, 96, 1)
    ang = 0.03
    for k, v in enumerate([np.cos(ang), 0, -np.sin(ang), 0, 1, 0, np.sin(ang), 0, np.cos(ang)]):
        b.depth_to_color.rotation[k] = float(v)
    c = S.synth_stream_config(128, 96, 2)
    c.depth_to_color.translation[2] = -0.7
    cfgs = [a, b, c]
    depth = [S.synth_depth(128, 96, i) for i in range(3)]
    color = [S.synth_color(cc.color.width, cc.color.height, i) for i, cc in enumerate(cfgs)]
    for i in range(3):
        with PcsContext([cfgs[i]]) as ctx:
            assert ctx.stream_math(0) == (0, 3, 0)[i]
            buf, counts, _ = ctx.process_frames([depth[i]], [color[i]])
        want, _ = oracle.process_frames([cfgs[i]], [depth[i]], [color[i]])
        assert_same(buf[2:2 + want.size].reshape(-1, 5), want)


def _random_config(rng, w, h, cw, ch, wild):
    f = rng.uniform(0.5, 1.5) * w
    di = make_intrinsics(w, h, f, f * rng.uniform(0.98, 1.02), w / 2 + rng.uniform(-20, 20), h / 2 + rng.uniform(-20, 20))
    fc = rng.uniform(0.5, 1.5) * cw
    cdist = rng.random() < 0.3
    ci = make_intrinsics(cw, ch, fc, fc * rng.uniform(0.98, 1.02), cw / 2 + rng.uniform(-20, 20), ch / 2 + rng.uniform(-20, 20),
                         model=1 if cdist else 0, coeffs=list(rng.normal(0, 0.02, 5)) if cdist else None)
    # small random rotation (Rodrigues), column-major
    ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
    ang = 0.0 if rng.random() < 0.3 else rng.normal(0, 0.6 if wild else 0.02)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    Rm = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
    t = rng.normal(0, 0.5 if wild else 0.02, 3)
    if rng.random() < 0.3:
        t[rng.integers(0, 3)] = 0.0
    scale = float(10.0 ** rng.uniform(-4, -2)) if wild else 0.001
    m = TRANSFORMS[rng.integers(0, 8)].copy()
    return make_stream_config(di, ci, cam_to_world=m, rotation=list(Rm.T.reshape(-1)), translation=list(t), depth_scale=scale)


@pytest.mark.parametrize("wild", [False, True])
def test_fuzzed_camera_configurations_both_policies(oracle, wild):
    """Random intrinsics / depth->colour extrinsics / scales: whatever policy the certificate picks, and the
    forced IEEE policy, must both equal the oracle bit for bit. `wild` includes configurations the
    certificate must refuse (large rotations, the colour camera behind the scene, tiny/huge scales)."""
    rng = np.random.default_rng(2024 + wild)
    picked = []
    for trial in range(24):
        w, h = [(64, 48), (128, 96), (104, 40), (200, 37)][trial % 4]
        cw, ch = [(64, 48), (192, 108), (100, 75), (320, 180)][(trial // 4) % 4]
        sc = _random_config(rng, w, h, cw, ch, wild)
        depth = S.synth_depth(w, h, trial, mode="random" if trial % 3 == 0 else "scene")
        color = S.synth_color(cw, ch, trial)
        want, _ = oracle.process_frames([sc], [depth], [color])
        for flags in (0, FLAG_FORCE_IEEE):
            with PcsContext([sc], flags=flags) as ctx:
                if flags == 0:
                    picked.append(ctx.stream_math(0))
                buf, _, _ = ctx.process_frames([depth], [color])
            d = first_diff(buf[2:2 + want.size].reshape(-1, 5), want)
            assert d is None, f"trial {trial} flags {flags} math {picked[-1]}: {d}"
    assert any(p > 0 for p in picked)                 # the certificate is not vacuous
    if wild:
        assert any(p == 0 for p in picked)            # ... and it does refuse


def test_lazy_convert_redo_path(oracle):
    """World coordinates beyond 2^31 mm: the saturating hardware convert and cvttss2si disagree there; the
    lane must notice and redo its points with the exact conversion."""
    cfgs, depth, color = S.synth_frame_set(1, 64, 48, single=True)
    m = np.array(list(cfgs[0].cam_to_world), np.float32)
    m[3] = 3.0e6      # +3000 km on x: x*1000 >= 2^31 for every point
    m[7] = -3.0e6     # and the negative side on y
    for k in range(16):
        cfgs[0].cam_to_world[k] = float(m[k])
    want, _ = oracle.process_frames(cfgs, depth, color)
    with PcsContext(cfgs) as ctx:
        assert ctx.stream_math(0) == 2      # certified, but NOT the no-overflow variant: the extrinsic rules it out
        buf, _, _ = ctx.process_frames(depth, color)
    assert_same(buf[2:2 + want.size].reshape(-1, 5), want)
    assert (want[:, 0] == 0).all()          # INT_MIN & 0xFFFF, the x86 answer (hardware alone would give -1)
    # and through pcs_set_cam_to_world: a context that starts in the no-overflow variant must leave it
    cfgs2, _, _ = S.synth_frame_set(1, 64, 48, single=True)
    with PcsContext(cfgs2) as ctx:
        assert ctx.stream_math(0) == 4
        ctx.set_cam_to_world(0, m)
        assert ctx.stream_math(0) == 2
        buf, _, _ = ctx.process_frames(depth, color)
    assert_same(buf[2:2 + want.size].reshape(-1, 5), want)


# ---------------------------------------------------------------------------------------------
# single-pass ordered compaction (published tile counts + direct-sum placement)
# ---------------------------------------------------------------------------------------------
@pytest.fixture(params=["three_pass", "single_pass"])
def compaction_path(request, monkeypatch):
    """Every compaction implementation must give the reference's `-c -m -t1` order bit for bit.
    three_pass = count + scan + emit (the default); single_pass = the one-launch kernel (opt-in by environment)."""
    for k in ("PCS_COMPACT_PATH", "PCS_COMPACT_SINGLE_PASS", "PCS_COMPACT_TICKETS"):
        monkeypatch.delenv(k, raising=False)
    if request.param == "single_pass":
        monkeypatch.setenv("PCS_COMPACT_PATH", "single")
    return request.param


@pytest.mark.parametrize("flags", [FLAG_DROP_INVALID, FLAG_CUTOFF | FLAG_DROP_INVALID, FLAG_CUTOFF])
def test_caller_provided_tile_counts_replace_the_count_pass(oracle, flags):
    """pcs_process_frames_device_counted: a producer that knows how many pixels of every 2048-pixel tile the predicate keeps
    hands the counts over; scan + emit alone must give the oracle's bytes. Garbage counts (any uint32) may garble the cloud
    but must neither fail nor write outside the payload (guard words behind it stay intact)."""
    shapes = [(640, 480), (200, 37), (1280, 720)]
    cfgs = [S.synth_stream_config(w, h, s) for s, (w, h) in enumerate(shapes)]
    depth = [S.synth_depth(w, h, s) for s, (w, h) in enumerate(shapes)]
    color = [S.synth_color(w, h, s) for s, (w, h) in enumerate(shapes)]
    for d in depth:
        d[10:30, :] //= 4          # bring part of the scene inside the 1.5 m cutoff
    want, wcounts = oracle.process_frames(cfgs, depth, color, flags)
    tile = 2048
    kept = []
    for cfg, d in zip(cfgs, depth):
        v, _ = oracle.deproject(cfg, d)
        k = np.ones(d.size, bool)
        if flags & FLAG_CUTOFF:
            k &= (v[:, 2] > 0) & (v[:, 2] <= np.float32(1.5)) & (v[:, 0] > -2) & (v[:, 0] <= 2)
        if flags & FLAG_DROP_INVALID:
            k &= d.reshape(-1) != 0
        pad = (-k.size) % tile
        kept.append(np.r_[k, np.zeros(pad, bool)].reshape(-1, tile).sum(1).astype(np.uint32))
    with PcsContext(cfgs, flags=flags) as ctx:
        assert [ctx.stream_tile_base(s) for s in range(4)] == list(np.r_[0, np.cumsum([k.size for k in kept])])
        counts_h = np.concatenate(kept)
        n_max = ctx.max_payload_shorts
        dd = [ctx.device_malloc(d.nbytes) for d in depth]; dc = [ctx.device_malloc(c.nbytes) for c in color]
        for ptr, a in zip(dd + dc, depth + color):
            ctx.memcpy_h2d(ptr, a)
        d_tc = ctx.device_malloc(counts_h.nbytes); ctx.memcpy_h2d(d_tc, counts_h)
        d_pay = ctx.device_malloc(n_max * 2 + 256); d_cnt = ctx.device_malloc(16)
        guard = np.full(128, 0x5A5A, np.int16)
        ctx.memcpy_h2d(d_pay + n_max * 2, guard)
        ctx.process_frames_device_counted(dd, dc, d_tc, d_pay, n_max, d_cnt)
        ctx.synchronize()
        cnt = np.empty(4, np.int32); ctx.memcpy_d2h(cnt, d_cnt)
        assert list(cnt[:3]) == wcounts and int(cnt[3]) == want.shape[0]
        got = np.empty(want.size, np.int16); ctx.memcpy_d2h(got, d_pay)
        assert_same(got.reshape(-1, 5), want)
        rng = np.random.default_rng(5)
        for trial in range(3):
            junk = rng.integers(0, 2**32, counts_h.size, dtype=np.uint64).astype(np.uint32)
            ctx.memcpy_h2d(d_tc, junk)
            ctx.process_frames_device_counted(dd, dc, d_tc, d_pay, n_max, d_cnt)
            ctx.synchronize()
            back = np.empty(128, np.int16); ctx.memcpy_d2h(back, d_pay + n_max * 2)
            assert (back == guard).all(), trial


def test_single_pass_compaction_chained_launches_and_mixed_sizes(oracle, compaction_path):
    # 21 streams -> two launches chained through stream_end; mixed raster sizes -> ticket->(stream, tile) search
    sizes = [(64, 48), (128, 96), (104, 40), (200, 37), (640, 480)]
    cfgs = [S.synth_stream_config(*sizes[i % len(sizes)], i) for i in range(21)]
    depth = [S.synth_depth(c.depth.width, c.depth.height, i) for i, c in enumerate(cfgs)]
    color = [S.synth_color(c.color.width, c.color.height, i) for i, c in enumerate(cfgs)]
    for flags in (FLAG_DROP_INVALID, FLAG_CUTOFF | FLAG_CUTOFF_COMPAT):
        got, counts = run_fused(cfgs, depth, color, flags)
        want, wcounts = oracle.process_frames(cfgs, depth, color, flags)
        assert counts == wcounts
        assert_same(got, want)


def test_single_pass_compaction_repeated_launches_device_api(oracle, compaction_path):
    # generations / tickets across many launches on one context; different inputs every time
    cfgs, _, _ = S.synth_frame_set(3, 640, 480)
    n_max = sum(c.n_points for c in cfgs) * 5
    with PcsContext(cfgs, flags=FLAG_DROP_INVALID) as ctx:
        out = ctx.device_malloc(n_max * 2 + 64)
        d_counts = ctx.device_malloc(4 * 4)
        dd = [ctx.device_malloc(c.n_points * 2) for c in cfgs]
        dc = [ctx.device_malloc(c.color_bytes) for c in cfgs]
        for it in range(40):
            depth = [S.synth_depth(640, 480, s, seed=1000 + it) for s in range(3)]
            color = [S.synth_color(640, 480, s, seed=1000 + it) for s in range(3)]
            if it % 7 == 3:
                depth[1][:] = 0                      # an entire stream dropped
            for p, a in zip(dd + dc, depth + color):
                ctx.memcpy_h2d(p, a)
            ctx.process_frames_device(dd, dc, out + 4, n_max, d_counts)     # +4: the reference's payload alignment
            ctx.synchronize()
            cnt = np.empty(4, np.int32)
            ctx.memcpy_d2h(cnt, d_counts)
            want, wcounts = oracle.process_frames(cfgs, depth, color, FLAG_DROP_INVALID)
            assert list(cnt[:3]) == wcounts and cnt[3] == want.shape[0], it
            got = np.empty(want.size, np.int16)
            if want.size:
                ctx.memcpy_d2h(got, out + 4)
            assert_same(got.reshape(-1, 5), want)


def test_three_pass_path_still_used_for_strided_compaction(oracle):
    cfgs, depth, color = S.synth_frame_set(2, 640, 480)
    got, counts = run_fused(cfgs, depth, color, FLAG_DROP_INVALID, 4)
    want, wcounts = oracle.process_frames(cfgs, depth, color, FLAG_DROP_INVALID, 4)
    assert counts == wcounts
    assert_same(got, want)


def test_drop_invalid_with_degenerate_depth_scale(oracle):
    """depth_scale so small that scale*d underflows to 0 for small d: `z != 0` is no longer `d != 0`, so the
    depth-only count shortcut must not be used (count and emit must agree)."""
    cfgs, depth, color = S.synth_frame_set(1, 64, 48)
    cfgs[0].depth_scale = 1e-45
    depth[0][:] = (np.arange(64 * 48).reshape(48, 64) % 7).astype(np.uint16)
    got, counts = run_fused(cfgs, depth, color, FLAG_DROP_INVALID)
    want, wcounts = oracle.process_frames(cfgs, depth, color, FLAG_DROP_INVALID)
    assert counts == wcounts
    assert_same(got, want)


def test_large_raster_multi_chunk_scan_and_dense(oracle):
    """2048x1100 = 1100 tiles per stream: more than one 1024-wide chunk in the per-stream scan; also the dense
    path on a raster larger than the reference's BUF_SIZE could ever hold."""
    cfgs = [S.synth_stream_config(2048, 1100, 0), S.synth_stream_config(2048, 1100, 1)]
    depth = [S.synth_depth(2048, 1100, i) for i in range(2)]
    color = [S.synth_color(2048, 1100, i) for i in range(2)]
    for flags in (0, FLAG_DROP_INVALID, FLAG_CUTOFF):
        got, counts = run_fused(cfgs, depth, color, flags)
        want, wcounts = oracle.process_frames(cfgs, depth, color, flags)
        assert counts == wcounts
        assert hashlib.sha256(got.tobytes()).hexdigest() == hashlib.sha256(want.tobytes()).hexdigest()


def test_last_colour_pixel_window_redo(oracle):
    """Tight RGB8 raster: only the very last colour pixel needs the dword window slid back; the fast path must
    notice (max index > limit) and redo the lane exactly. Force every point onto that pixel."""
    cfgs, depth, color = S.synth_frame_set(1, 64, 48, single=True)
    cfgs[0].color.ppx = 1.0e6        # projects everything far right/bottom -> clamps to (W-1, H-1)
    cfgs[0].color.ppy = 1.0e6
    want, _ = oracle.process_frames(cfgs, depth, color)
    with PcsContext(cfgs) as ctx:
        assert ctx.stream_math(0) in (2, 4)
        buf, _, _ = ctx.process_frames(depth, color)
    got = buf[2:2 + want.size].reshape(-1, 5)
    assert_same(got, want)
    valid = depth[0].reshape(-1) != 0
    last = color[0][-3:]
    assert (got[valid, 3].view(np.uint16) == (int(last[0]) | int(last[1]) << 8)).all() and (got[valid, 4] == last[2]).all()


def test_adversarial_configurations_near_certificate_thresholds(oracle):
    """Configurations pushed towards the edges of what certify_stream / certify_no_overflow accept or refuse:
    tiny and huge depth scales, colour plane almost through the camera, strong rotations, extreme focal
    lengths, extreme depth values. Whatever the certificates decide, the output must equal the oracle."""
    rng = np.random.default_rng(77)
    depth_vals = np.array([1, 2, 3, 65535, 65534, 32768, 1000, 0, 7, 500], np.uint16)
    seen = set()
    for trial in range(120):
        w, h = 64, 16
        cw, ch = [(64, 16), (48, 40), (128, 8)][trial % 3]
        fx = float(10.0 ** rng.uniform(0.5, 4.5))
        di = make_intrinsics(w, h, fx, fx * rng.uniform(0.5, 2.0), rng.uniform(-50, 150), rng.uniform(-50, 60))
        ci = make_intrinsics(cw, ch, float(10.0 ** rng.uniform(0.5, 4.5)), float(10.0 ** rng.uniform(0.5, 4.5)),
                             rng.uniform(-1e3, 1e3), rng.uniform(-1e3, 1e3))
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        ang = [0.0, 1e-4, 0.05, 0.6, 1.5, 3.0][trial % 6]
        K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        Rm = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
        scale = float([1e-3, 1e-6, 1e-12, 2.0 ** -39, 2.0 ** -41, 1.0, 300.0, 1e5][trial % 8])
        t = rng.normal(0, 1, 3) * [0.02, 0.02, 0.02]
        if trial % 5 == 0:
            t[2] = -scale * float(rng.integers(1, 5))          # P2 = Z + t2 crosses zero for small depths
        if trial % 7 == 0:
            t[:] = 0.0
        m = TRANSFORMS[trial % 8].copy()
        if trial % 9 == 0:
            m[3] = 2.0e6                                        # world x beyond 2^31 mm
        sc = make_stream_config(di, ci, cam_to_world=m, rotation=list(Rm.T.reshape(-1)), translation=list(t), depth_scale=scale)
        depth = depth_vals[rng.integers(0, depth_vals.size, w * h)].reshape(h, w)
        color = S.synth_color(cw, ch, trial)
        want, _ = oracle.process_frames([sc], [depth], [color])
        for flags in (0, FLAG_FORCE_IEEE):
            with PcsContext([sc], flags=flags) as ctx:
                if flags == 0:
                    seen.add(ctx.stream_math(0))
                buf, _, _ = ctx.process_frames([depth], [color])
            d = first_diff(buf[2:2 + want.size].reshape(-1, 5), want)
            assert d is None, f"trial {trial} flags {flags}: {d}"
    assert 0 in seen and (seen - {0})             # both refused and accepted configurations occurred


# ---------------------------------------------------------------------------------------------
# software-pipelined host form (pcs_submit_frames / pcs_collect_frames)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("flags", [0, FLAG_DROP_INVALID])
@pytest.mark.parametrize("pinned", [False, True])
def test_submit_collect_pipeline_matches_synchronous_calls(oracle, flags, pinned):
    cfgs, _, _ = S.synth_frame_set(3, 320, 240)
    frames = [([S.synth_depth(320, 240, s, seed=500 + k) for s in range(3)],
               [S.synth_color(320, 240, s, seed=500 + k) for s in range(3)]) for k in range(5)]
    with PcsContext(cfgs, flags=flags) as ctx:
        if pinned:                                             # what makes the two directions overlap
            staged = []
            for dep, col in frames:
                pd = [ctx.host_array(d.shape, np.uint16) for d in dep]
                pc = [ctx.host_array(c.shape, np.uint8) for c in col]
                for a, b in zip(pd + pc, dep + col):
                    a[...] = b
                staged.append((pd, pc))
            out = ctx.host_array((2 + ctx.max_payload_shorts,), np.int16)
        else:
            staged = frames
            out = np.zeros(2 + ctx.max_payload_shorts, np.int16)
        results = []
        t_prev = ctx.submit_frames(*staged[0])
        for k in range(1, len(staged) + 1):                   # submit(k+1); collect(k)
            t_next = ctx.submit_frames(*staged[k]) if k < len(staged) else None
            buf, counts, nbytes = ctx.collect_frames(t_prev, out)
            results.append((buf[2:2 + nbytes // 2].reshape(-1, 5).copy(), counts, int(buf[:2].view(np.int32)[0]), nbytes))
            t_prev = t_next
    for k, (dep, col) in enumerate(frames):
        want, wcounts = oracle.process_frames(cfgs, dep, col, flags)
        got, counts, header, nbytes = results[k]
        assert counts == wcounts and header == nbytes == want.nbytes, k
        assert_same(got, want)


def test_submit_collect_misuse_is_reported():
    cfgs, depth, color = S.synth_frame_set(1, 64, 48)
    with PcsContext(cfgs) as ctx:
        out = np.zeros(2 + ctx.max_payload_shorts, np.int16)
        with pytest.raises(PcsError, match="not in flight"):
            ctx.collect_frames(0, out)
        t0 = ctx.submit_frames(depth, color)
        t1 = ctx.submit_frames(depth, color)
        with pytest.raises(PcsError, match="in flight"):      # PCS_PIPELINE_DEPTH = 2
            ctx.submit_frames(depth, color)
        with pytest.raises(PcsError, match="submission order"):
            ctx.collect_frames(t1, out)
        ctx.collect_frames(t0, out)
        small = np.zeros(10, np.int16)
        with pytest.raises(PcsError, match="stitched buffer"):
            ctx.collect_frames(t1, small)
        t2 = ctx.submit_frames(depth, color)                   # the slot of the dropped frame-set is free again
        buf, counts, _ = ctx.collect_frames(t2, out)
        assert counts == [64 * 48]


# ---------------------------------------------------------------------------------------------
# PCS_FLAG_TEXCOORD_HALF_PIXEL: the older librealsense texcoord formula
# ---------------------------------------------------------------------------------------------
def test_half_pixel_texcoords_deproject_and_fused(oracle):
    H = FLAG_TEXCOORD_HALF_PIXEL
    cfgs, depth, color = S.synth_frame_set(3, 320, 240)
    with PcsContext(cfgs, flags=H) as ctx:
        for s in range(3):
            vtx, tex = ctx.deproject(s, depth[s])
            wv, wt = oracle.deproject(cfgs[s], depth[s], H)
            assert (vtx.view(np.uint32) == wv.view(np.uint32)).all() and (tex.view(np.uint32) == wt.view(np.uint32)).all()
    for flags in (H, H | FLAG_DROP_INVALID, H | FLAG_CUTOFF, H | FLAG_FORCE_IEEE):
        for ds in (1, 3):
            got, counts = run_fused(cfgs, depth, color, flags, ds)
            want, wcounts = oracle.process_frames(cfgs, depth, color, flags, ds)
            assert counts == wcounts
            assert_same(got, want)
    # and it really is a different result from the default convention
    base, _ = oracle.process_frames(cfgs, depth, color)
    half, _ = oracle.process_frames(cfgs, depth, color, H)
    assert (base != half).any()


def test_half_pixel_texcoords_fuzzed_configurations(oracle):
    rng = np.random.default_rng(77)
    for trial in range(16):
        w, h = [(64, 48), (128, 96), (104, 40), (200, 37)][trial % 4]
        cw, ch = [(64, 48), (192, 108), (100, 75), (320, 180)][(trial // 4) % 4]
        sc = _random_config(rng, w, h, cw, ch, wild=trial % 2 == 1)
        depth = S.synth_depth(w, h, trial, mode="random" if trial % 3 == 0 else "scene")
        color = S.synth_color(cw, ch, trial)
        want, _ = oracle.process_frames([sc], [depth], [color], FLAG_TEXCOORD_HALF_PIXEL)
        for extra in (0, FLAG_FORCE_IEEE):
            got, _ = run_fused([sc], [depth], [color], FLAG_TEXCOORD_HALF_PIXEL | extra)
            d = first_diff(got, want)
            assert d is None, f"trial {trial} extra {extra}: {d}"


@pytest.mark.parametrize("flags,stride", [(0, 1), (FLAG_DROP_INVALID, 1), (FLAG_CUTOFF | FLAG_CUTOFF_COMPAT, 1), (FLAG_DROP_INVALID, 3)])
@pytest.mark.parametrize("shapes", [[(640, 480)] * 3, [(1280, 720), (321, 243)]])
def test_host_api_zero_copy_with_pinned_buffers(oracle, flags, stride, shapes):
    """pcs_process_frames with page-locked rasters and stitched buffer (pcs_host_malloc): the kernels read and write the
    host memory directly, no staging copies. Same bytes, counts and header as the staged route (pageable numpy arrays,
    and pinned ones with one raster left pageable) and as the oracle."""
    cfgs = [S.synth_stream_config(w, h, s) for s, (w, h) in enumerate(shapes)]
    depth = [S.synth_depth(w, h, s) for s, (w, h) in enumerate(shapes)]
    color = [S.synth_color(w, h, s) for s, (w, h) in enumerate(shapes)]
    want, wcounts = oracle.process_frames(cfgs, depth, color, flags, stride)
    with PcsContext(cfgs, flags=flags, downsample=stride) as ctx:
        pd = [ctx.host_array(d.shape, np.uint16) for d in depth]
        pc = [ctx.host_array(c.shape, np.uint8) for c in color]
        for dst, src in zip(pd + pc, depth + color):
            dst[...] = src
        out = ctx.host_array((2 + ctx.max_payload_shorts,), np.int16)
        out[...] = 0x5555
        # a caller's own (numpy = malloc'ed) arrays, page-locked in place: zero copy as well
        rd = [np.array(d) for d in depth]; rc = [np.array(c) for c in color]
        rout = np.full(2 + ctx.max_payload_shorts, 0x5555, np.int16)
        for a in rd + rc + [rout]:
            ctx.host_register(a)
        buf, counts, size = ctx.process_frames(rd, rc, out=rout)
        assert counts == wcounts and size == want.size * 2
        assert_same(buf[2:2 + want.size].reshape(-1, 5), want)
        for a in rd + rc + [rout]:
            ctx.host_unregister(a)
        for trial, (dd, cc) in enumerate([(pd, pc), (pd[:-1] + [depth[-1]], pc), (depth, color)]):     # zero copy, staged, staged
            buf, counts, size = ctx.process_frames(dd, cc, out=out)
            assert counts == wcounts and size == want.size * 2, trial
            assert int(np.frombuffer(buf[:2].tobytes(), np.int32)[0]) == size
            assert_same(buf[2:2 + want.size].reshape(-1, 5), want)
            if trial == 0 and flags:
                assert (buf[2 + want.size:2 + want.size + 64] == 0x5555).all()     # nothing written past the kept points
            out[...] = 0x5555


def test_host_api_partially_registered_buffers_are_refused_not_faulted(oracle):
    """Zero copy needs the page-locked range to cover EVERYTHING the kernels touch. A stitched buffer of which only the first
    half is registered, an interior pointer too close to the end of a registration and a raster registered without its last
    pages are refused with PCS_ERR_INVALID_ARG and a reason (the HIP runtime also refuses staging copies that straddle the
    edge of a registration) — before any kernel could run off the end of a mapping and fault on the GPU. Fully registered
    and fully pageable buffers keep working on the same context."""
    shapes = [(640, 480)] * 2
    cfgs = [S.synth_stream_config(w, h, s) for s, (w, h) in enumerate(shapes)]
    depth = [S.synth_depth(w, h, s) for s, (w, h) in enumerate(shapes)]
    color = [S.synth_color(w, h, s) for s, (w, h) in enumerate(shapes)]
    want, wcounts = oracle.process_frames(cfgs, depth, color)
    with PcsContext(cfgs) as ctx:
        rd = [np.array(d) for d in depth]; rc = [np.array(c) for c in color]
        big = np.full(4 * (2 + ctx.max_payload_shorts), 0x5555, np.int16)
        rout = big[:2 + ctx.max_payload_shorts]
        for a in rd + rc:
            ctx.host_register(a)
        # (a) output registered for its first half only
        ctx.host_register(rout[:rout.size // 2])
        with pytest.raises(PcsError) as e:
            ctx.process_frames(rd, rc, out=rout)
        assert e.value.status == -1 and "page-locked for only part" in str(e.value)
        ctx.host_unregister(rout[:rout.size // 2])
        # (b) whole output registered, handed over through an interior pointer that leaves too little room behind it
        ctx.host_register(rout)
        tail = big[rout.size // 2:rout.size // 2 + 2 + ctx.max_payload_shorts]      # starts inside the registration, ends outside
        with pytest.raises(PcsError) as e:
            ctx.process_frames(rd, rc, out=tail)
        assert e.value.status == -1
        # (c) a raster registered without its last 8 KiB
        ctx.host_unregister(rd[0])
        ctx.host_register(rd[0].reshape(-1)[:rd[0].size - 4096])
        with pytest.raises(PcsError) as e:
            ctx.process_frames(rd, rc, out=rout)
        assert e.value.status == -1 and "depth raster 0" in str(e.value)
        ctx.host_unregister(rd[0].reshape(-1)[:rd[0].size - 4096])
        # everything registered in full: zero copy; everything pageable: staged — the same bytes
        ctx.host_register(rd[0])
        buf, counts, size = ctx.process_frames(rd, rc, out=rout)
        assert counts == wcounts and size == want.size * 2
        assert_same(buf[2:2 + want.size].reshape(-1, 5), want)
        for a in rd + rc + [rout]:
            ctx.host_unregister(a)
        buf, counts, size = ctx.process_frames(rd, rc, out=rout)
        assert counts == wcounts
        assert_same(buf[2:2 + want.size].reshape(-1, 5), want)


def test_zero_copy_verdicts_survive_registration_changes_made_through_another_context(oracle):
    """A context caches per buffer whether it is page-locked (zero copy) or pageable (staged). The buffers can change state
    behind its back — registered, then unregistered, through ANOTHER context (or plain HIP): a stale "zero copy" verdict would
    let the kernels dereference a dead device mapping (a GPU fault), a stale "pageable" one only costs speed. Every call
    re-validates with one attribute query, so the same bytes come out in every state."""
    shapes = [(320, 240)] * 2
    cfgs = [S.synth_stream_config(w, h, s) for s, (w, h) in enumerate(shapes)]
    depth = [S.synth_depth(w, h, s) for s, (w, h) in enumerate(shapes)]
    color = [S.synth_color(w, h, s) for s, (w, h) in enumerate(shapes)]
    want, wcounts = oracle.process_frames(cfgs, depth, color)
    with PcsContext(cfgs) as ctx, PcsContext(cfgs[:1]) as other:
        rd = [np.array(d) for d in depth]; rc = [np.array(c) for c in color]
        rout = np.zeros(2 + ctx.max_payload_shorts, np.int16)

        def run():
            rout[:] = 0x5555
            buf, counts, size = ctx.process_frames(rd, rc, out=rout)
            assert counts == wcounts and size == want.size * 2
            assert_same(buf[2:2 + want.size].reshape(-1, 5), want)
        run()                                                    # pageable: verdicts "staged" cached
        for a in rd + rc + [rout]:
            other.host_register(a)
        run()                                                    # now page-locked: zero copy
        for a in rd + rc + [rout]:
            other.host_unregister(a)
        run()                                                    # the cached device views are dead: must stage again, not fault
        run()


@pytest.mark.parametrize("flags", [0, FLAG_CUTOFF, FLAG_CUTOFF | FLAG_CUTOFF_COMPAT])
@pytest.mark.parametrize("n", [1, 7, 2048, 70001])
def test_a2_twin_zero_copy_with_pinned_arrays(oracle, flags, n):
    """pcs_copy_pointcloud_xyzrgb_to_buffer with page-locked vertices, texcoords, colour and pc_buffer: the kernel reads
    and writes them in place; same bytes and count as the staged route and the oracle."""
    sc, V, T, col = random_points(n, 4321 + n)
    want = oracle.pack(sc, V, T, col, flags)
    with PcsContext([sc], flags=flags) as ctx:
        pv = ctx.host_array(V.shape, np.float32); pv[...] = V
        pt = ctx.host_array(T.shape, np.float32); pt[...] = T
        pcol = ctx.host_array(col.shape, np.uint8); pcol[...] = col
        pout = ctx.host_array((n + 4, 5), np.int16); pout[...] = 0x5555
        out, cnt = ctx.copy_pointcloud_xyzrgb_to_buffer(0, pv, pt, pcol, pc_buffer=pout)
        assert cnt == want.shape[0]
        assert_same(out.reshape(-1, 5)[:cnt], want)
        assert (out.reshape(-1, 5)[cnt if flags else n:] == 0x5555).all()
        got2, cnt2 = ctx.copy_pointcloud_xyzrgb_to_buffer(0, V, T, col)               # staged
        assert cnt2 == cnt
        assert_same(got2, want)


# ---------------------------------------------------------------------------------------------
# throughput forms: K frame-sets per launch, all cameras' rs2::points per launch
# ---------------------------------------------------------------------------------------------
def _upload(ctx, arrays):
    ptrs = [ctx.device_malloc(max(a.nbytes, 16)) for a in arrays]
    for p, a in zip(ptrs, arrays):
        ctx.memcpy_h2d(p, a)
    return ptrs


@pytest.mark.parametrize("n_streams,shape,n_sets", [(3, (640, 480), 3), (8, (320, 240), 11), (1, (1280, 720), 5),
                                                    (40, (64, 48), 3)])
def test_batch_of_frame_sets_equals_single_calls(oracle, n_streams, shape, n_sets):
    """pcs_process_frames_device_batch: every frame-set's payload is byte-identical to a pcs_process_frames_device
    call on it (and to the oracle). 8 x 11 sets exceeds one launch's 64 entries -> two batched launches; 40 streams
    leaves room for one set per launch -> the set-by-set fallback."""
    cfgs = [S.synth_stream_config(*shape, s) for s in range(n_streams)]
    sets = [([S.synth_depth(*shape, s, seed=S.SEED + 101 * k) for s in range(n_streams)],
             [S.synth_color(*shape, s, seed=S.SEED + 101 * k) for s in range(n_streams)]) for k in range(n_sets)]
    n_sh = sum(c.n_points for c in cfgs) * POINT_SHORTS
    with PcsContext(cfgs) as ctx:
        dd = [_upload(ctx, d) for d, _ in sets]
        dc = [_upload(ctx, c) for _, c in sets]
        outs = [ctx.device_malloc(n_sh * 2 + 64) for _ in range(n_sets)]
        singles = []
        for k in range(n_sets):
            ctx.process_frames_device(dd[k], dc[k], outs[k], n_sh)
            ctx.synchronize()
            got = np.empty(n_sh, np.int16)
            ctx.memcpy_d2h(got, outs[k])
            singles.append(got)
            ctx.memcpy_h2d(outs[k], np.zeros(n_sh, np.int16))
        d_counts = [ctx.device_malloc(4 * (n_streams + 1)) for _ in range(n_sets)]
        ctx.process_frames_device_batch(dd, dc, outs, n_sh, d_counts)
        ctx.synchronize()
        for k in range(n_sets):
            got = np.empty(n_sh, np.int16)
            ctx.memcpy_d2h(got, outs[k])
            assert_same(got, singles[k])
            cnt = np.empty(n_streams + 1, np.int32)
            ctx.memcpy_d2h(cnt, d_counts[k])
            assert list(cnt[:-1]) == [c.n_points for c in cfgs] and cnt[-1] == n_sh // POINT_SHORTS
            if k in (0, n_sets - 1):
                want, _ = oracle.process_frames(cfgs, sets[k][0], sets[k][1])
                assert_same(got.reshape(-1, 5), want)


@pytest.mark.parametrize("flags,skew", [(FLAG_DROP_INVALID, 0), (0, 4)])
def test_batch_of_frame_sets_non_dense_configurations(oracle, flags, skew):
    """Compaction, or the reference's payload alignment (buffer + 4 bytes): the batch call runs set by set."""
    cfgs, _, _ = S.synth_frame_set(2, 640, 480)
    sets = [S.synth_frame_set(2, 640, 480, seed=S.SEED + 7 * k)[1:] for k in range(3)]
    n_sh = sum(c.n_points for c in cfgs) * POINT_SHORTS
    with PcsContext(cfgs, flags=flags) as ctx:
        dd = [_upload(ctx, d) for d, _ in sets]
        dc = [_upload(ctx, c) for _, c in sets]
        outs = [ctx.device_malloc(n_sh * 2 + 64) + skew for _ in range(3)]
        d_counts = [ctx.device_malloc(4 * 3) for _ in range(3)]
        ctx.process_frames_device_batch(dd, dc, outs, n_sh, d_counts)
        ctx.synchronize()
        for k in range(3):
            want, wcounts = oracle.process_frames(cfgs, sets[k][0], sets[k][1], flags)
            cnt = np.empty(3, np.int32)
            ctx.memcpy_d2h(cnt, d_counts[k])
            assert list(cnt[:2]) == wcounts
            got = np.empty(want.size, np.int16)
            ctx.memcpy_d2h(got, outs[k])
            assert_same(got.reshape(-1, 5), want)


@pytest.mark.parametrize("flags", [FLAG_DROP_INVALID, FLAG_CUTOFF, FLAG_CUTOFF | FLAG_CUTOFF_COMPAT,
                                   FLAG_CUTOFF | FLAG_DROP_INVALID])
@pytest.mark.parametrize("shapes,n_sets,skew", [([(640, 480)] * 3, 5, 0), ([(1280, 720), (321, 243), (64, 48), (640, 480)], 18, 2),
                                                ([(160, 120)] * 16, 5, 0)])
def test_batched_compaction_equals_the_oracle(oracle, flags, shapes, n_sets, skew):
    """pcs_process_frames_device_batch with a predicate: count, scan and emit each cover every frame-set of the launch
    (grid.z = set). Every set's payload and counts equal the oracle's; 4 streams x 18 sets = two launches of <= 16 sets;
    16 streams x 5 sets = launches of 4 sets; ragged rasters, payloads 2-byte aligned only, some sets without a counts
    pointer."""
    n = len(shapes)
    cfgs = [S.synth_stream_config(w, h, s) for s, (w, h) in enumerate(shapes)]
    sets = [([S.synth_depth(w, h, s, seed=S.SEED + 31 * k) for s, (w, h) in enumerate(shapes)],
             [S.synth_color(w, h, s, seed=S.SEED + 31 * k) for s, (w, h) in enumerate(shapes)]) for k in range(n_sets)]
    sets[1][0][0][:] = 0                                     # a stream with nothing kept
    if n_sets > 2:
        for d in sets[2][0]:
            d[:] = 0                                         # a whole frame-set with nothing kept
    n_sh = sum(c.n_points for c in cfgs) * POINT_SHORTS
    with PcsContext(cfgs, flags=flags) as ctx:
        dd = [_upload(ctx, d) for d, _ in sets]
        dc = [_upload(ctx, c) for _, c in sets]
        outs = [ctx.device_malloc(n_sh * 2 + 64) + skew for _ in range(n_sets)]
        d_counts = [ctx.device_malloc(4 * (n + 1)) if k % 3 != 1 else None for k in range(n_sets)]
        for rep in range(2):                                 # the scratch rows are reused by the second call
            ctx.process_frames_device_batch(dd, dc, outs, n_sh, d_counts)
        ctx.synchronize()
        for k in range(n_sets):
            want, wcounts = oracle.process_frames(cfgs, sets[k][0], sets[k][1], flags)
            if d_counts[k] is not None:
                cnt = np.empty(n + 1, np.int32)
                ctx.memcpy_d2h(cnt, d_counts[k])
                assert list(cnt[:n]) == wcounts and cnt[n] == sum(wcounts), f"set {k}"
            got = np.empty(max(want.size, 1), np.int16)
            ctx.memcpy_d2h(got, outs[k])
            assert_same(got[:want.size].reshape(-1, 5), want)


def test_batched_compaction_with_a_stride_runs_set_by_set(oracle):
    cfgs, _, _ = S.synth_frame_set(2, 320, 240)
    sets = [S.synth_frame_set(2, 320, 240, seed=S.SEED + 3 * k)[1:] for k in range(3)]
    n_sh = sum(c.n_points for c in cfgs) * POINT_SHORTS
    with PcsContext(cfgs, flags=FLAG_DROP_INVALID, downsample=3) as ctx:
        dd = [_upload(ctx, d) for d, _ in sets]
        dc = [_upload(ctx, c) for _, c in sets]
        outs = [ctx.device_malloc(n_sh * 2 + 64) for _ in range(3)]
        ctx.process_frames_device_batch(dd, dc, outs, n_sh, None)
        ctx.synchronize()
        for k in range(3):
            want, _ = oracle.process_frames(cfgs, sets[k][0], sets[k][1], FLAG_DROP_INVALID, downsample=3)
            got = np.empty(want.size, np.int16)
            ctx.memcpy_d2h(got, outs[k])
            assert_same(got.reshape(-1, 5), want)


@pytest.mark.parametrize("skew", [0, 4])
def test_batched_pack_equals_single_clouds(oracle, skew):
    """pcs_copy_pointclouds_xyzrgb_to_buffer_device: 19 cameras (two launches), ragged point counts incl. 0, each
    cloud byte-identical to the oracle's copyPointCloudXYZRGBToBufferSIMD; skew 4 = the reference's buffer + 2 shorts."""
    sizes = [5000, 0, 1, 2048, 2049, 8, 307200, 7, 4095, 64, 100, 12345, 3, 2047, 4096, 9, 640 * 48, 17, 921600]
    clouds = [random_points(n, 500 + i, spread=3.0) for i, n in enumerate(sizes)]
    cfgs = [c[0] for c in clouds]
    with PcsContext(cfgs) as ctx:
        descs, outs = [], []
        for i, (sc, V, T, col) in enumerate(clouds):
            dv, dt, dcol = _upload(ctx, [V, T, col])
            out = ctx.device_malloc(max(V.shape[0], 1) * 10 + 64) + skew
            outs.append(out)
            descs.append((i, V.shape[0], dv, dt, dcol, out))
        d_cnt = ctx.device_malloc(4 * len(sizes))
        ctx.copy_pointclouds_xyzrgb_to_buffer_device(descs, d_cnt)
        ctx.synchronize()
        cnt = np.empty(len(sizes), np.int32)
        ctx.memcpy_d2h(cnt, d_cnt)
        assert list(cnt) == sizes
        for i, (sc, V, T, col) in enumerate(clouds):
            if sizes[i] == 0:
                continue
            got = np.empty(sizes[i] * 5, np.int16)
            ctx.memcpy_d2h(got, outs[i])
            assert_same(got.reshape(-1, 5), oracle.pack(sc, V, T, col))


def test_batched_pack_with_cutoff_runs_cloud_by_cloud(oracle):
    clouds = [random_points(n, 900 + i, spread=1.5) for i, n in enumerate([3000, 10007, 5])]
    for _, V, _, _ in clouds:
        V[:, 2] = np.abs(V[:, 2])
    cfgs = [c[0] for c in clouds]
    with PcsContext(cfgs, flags=FLAG_CUTOFF) as ctx:
        descs, outs = [], []
        for i, (sc, V, T, col) in enumerate(clouds):
            dv, dt, dcol = _upload(ctx, [V, T, col])
            out = ctx.device_malloc(V.shape[0] * 10 + 64)
            outs.append(out)
            descs.append((i, V.shape[0], dv, dt, dcol, out))
        d_cnt = ctx.device_malloc(12)
        ctx.copy_pointclouds_xyzrgb_to_buffer_device(descs, d_cnt)
        ctx.synchronize()
        cnt = np.empty(3, np.int32)
        ctx.memcpy_d2h(cnt, d_cnt)
        for i, (sc, V, T, col) in enumerate(clouds):
            want = oracle.pack(sc, V, T, col, FLAG_CUTOFF)
            assert cnt[i] == want.shape[0]
            got = np.empty(want.size, np.int16)
            if want.size:
                ctx.memcpy_d2h(got, outs[i])
            assert_same(got.reshape(-1, 5), want)


@pytest.mark.parametrize("flags", [FLAG_CUTOFF, FLAG_CUTOFF | FLAG_CUTOFF_COMPAT, FLAG_CUTOFF | FLAG_CUTOFF_COMPAT | FLAG_DROP_INVALID])
@pytest.mark.parametrize("fx", [100.0, 448.0])
def test_cutoff_count_from_depth_words_and_its_refusal(oracle, flags, fx, compaction_path):
    """-c on the count pass: with a narrow lens (fx 448: 1.5 * max|mx| < 2) the host certifies that the x test cannot
    fail inside the z range and the count pass tests the raw Z16 word against a precomputed d_max; with a 145-degree lens
    (fx 100) the x test rejects points at the image border and the count pass must deproject. Depths straddle 1.5 m
    (d = 1499..1502 at scale 0.001) so the d_max boundary itself is exercised."""
    w, h = 640, 480
    it = make_intrinsics(w, h, fx, fx, w / 2 - 0.5, h / 2 - 0.5)
    cfgs = [make_stream_config(it, it, cam_to_world=TRANSFORMS[s]) for s in range(2)]
    rng = np.random.default_rng(int(fx))
    depth = [rng.integers(0, 3000, (h, w)).astype(np.uint16) for _ in range(2)]
    for d in depth:
        d[::7, ::3] = rng.integers(1497, 1504, d[::7, ::3].shape)
        d[5:40, :] = 0
    color = [S.synth_color(w, h, s) for s in range(2)]
    got, counts = run_fused(cfgs, depth, color, flags)
    want, wcounts = oracle.process_frames(cfgs, depth, color, flags)
    assert counts == wcounts and 0 < sum(counts) < 2 * w * h
    assert_same(got, want)
