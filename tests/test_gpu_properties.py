"""Size-independent properties of the hot path at BASELINE.json's full sizes (8 x 1280x720 and 16 x 1920x1080),
checked on the GPU output alone — no oracle involved, so they hold the path to account where a full CPU
comparison would take minutes:

  * a7 concatenation: the stitched payload of N cameras is the cameras' single-stream payloads in camera order
  * stride:           `downsample = d` keeps exactly every d-th record of each camera's kept sequence
  * compaction:       drop-invalid keeps exactly the pixels with non-zero depth, in raster order, and the kept
                      records are the dense records at those pixels
  * invalid pixels:   dense records of zero-depth pixels are the transform's translation (camera origin) with
                      the colour of texel (0,0)'s clamp target — identical for every invalid pixel of a camera
  * idempotence / determinism: two runs give identical bytes
"""
import numpy as np
import pytest

from pointcloud_stitching_amd import synthetic as S
from pointcloud_stitching_amd.api import PcsContext
from pointcloud_stitching_amd.types import FLAG_DROP_INVALID

pytestmark = pytest.mark.gpu


def run(cfgs, depth, color, flags=0, ds=1):
    with PcsContext(cfgs, flags=flags, downsample=ds) as ctx:
        buf, counts, nbytes = ctx.process_frames(depth, color)
    return buf[2:2 + nbytes // 2].reshape(-1, 5).copy(), counts


@pytest.fixture(scope="module", params=[(8, 1280, 720), (16, 1920, 1080)], ids=["8x720p", "16x1080p"])
def workload(request):
    n, W, H = request.param
    cfgs, depth, color = S.synth_frame_set(n, W, H)
    dense, counts = run(cfgs, depth, color)
    assert counts == [W * H] * n
    return cfgs, depth, color, dense


def test_concatenation_in_camera_order(workload):
    cfgs, depth, color, dense = workload
    N = cfgs[0].n_points
    for s in (0, len(cfgs) // 2, len(cfgs) - 1):                    # three cameras, each alone on the GPU
        alone, _ = run([cfgs[s]], [depth[s]], [color[s]])
        assert np.array_equal(alone, dense[s * N:(s + 1) * N]), s


def test_runs_are_deterministic(workload):
    cfgs, depth, color, dense = workload
    again, _ = run(cfgs, depth, color)
    assert np.array_equal(again, dense)


@pytest.mark.parametrize("ds", [2, 7])
def test_stride_keeps_every_dth_record_per_camera(workload, ds):
    cfgs, depth, color, dense = workload
    N = cfgs[0].n_points
    got, counts = run(cfgs, depth, color, ds=ds)
    per = -(-N // ds)
    assert counts == [per] * len(cfgs)
    want = np.concatenate([dense[s * N:(s + 1) * N][::ds] for s in range(len(cfgs))])
    assert np.array_equal(got, want)


def test_drop_invalid_is_a_stable_filter_of_the_dense_output(workload):
    cfgs, depth, color, dense = workload
    N = cfgs[0].n_points
    got, counts = run(cfgs, depth, color, FLAG_DROP_INVALID)
    keep = np.concatenate([d.reshape(-1) != 0 for d in depth])
    assert counts == [int((d != 0).sum()) for d in depth]
    assert np.array_equal(got, dense[keep])
    # and with a stride on top: every 3rd KEPT record of each camera (a7 strides the kept sequence)
    got3, counts3 = run(cfgs, depth, color, FLAG_DROP_INVALID, ds=3)
    want3 = np.concatenate([dense[s * N:(s + 1) * N][keep[s * N:(s + 1) * N]][::3] for s in range(len(cfgs))])
    assert np.array_equal(got3, want3)


def test_invalid_pixels_collapse_to_one_record_per_camera(workload):
    cfgs, depth, color, dense = workload
    N = cfgs[0].n_points
    for s in range(len(cfgs)):
        inv = dense[s * N:(s + 1) * N][depth[s].reshape(-1) == 0]
        assert inv.shape[0] > 0
        assert (inv == inv[0]).all()
        # vertex (0,0,0) -> world = translation column, in truncated millimetres (low 16 bits)
        m = np.array(list(cfgs[s].cam_to_world), np.float32).reshape(4, 4)
        t_mm = (m[:3, 3] * np.float32(1000.0)).astype(np.int64) & 0xFFFF
        assert list(inv[0, :3].astype(np.int64) & 0xFFFF) == list(t_mm)
        # texcoord (0,0) -> colour pixel (0,0)
        rgb = color[s].reshape(-1)[:3]
        assert (int(inv[0, 3]) & 0xFFFF) == int(rgb[0]) | (int(rgb[1]) << 8) and (int(inv[0, 4]) & 0xFFFF) == int(rgb[2])
