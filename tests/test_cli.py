"""The C++ host CLI (pcs-camera-optimized work-alike): flag surface and loud failure without a GPU here;
on the GPU box, its output against the oracle (which also pins the C++ synthetic generator to the
Python one)."""
import os
import re
import subprocess

import numpy as np
import pytest

from pointcloud_stitching_amd import synthetic as S
from pointcloud_stitching_amd.types import FLAG_CUTOFF, FLAG_CUTOFF_COMPAT, FLAG_DROP_INVALID

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI_DIR = os.path.join(ROOT, "pointcloud_stitching_amd", "cli")
BIN = os.path.join(ROOT, "pointcloud_stitching_amd", "bin", "pcs-camera-optimized")


@pytest.fixture(scope="module")
def cli():
    subprocess.run(["make", "-C", CLI_DIR], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert os.path.exists(BIN)
    return BIN


def run(cli, *args, timeout=120):
    return subprocess.run([cli, *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_help_lists_the_reference_flags(cli):
    r = run(cli, "-h")
    assert r.returncode == 0
    for flag in ("-f", "-s", "-m", "-t", "-c"):          # readme.md:45-49 + getopt string :122
        assert flag in r.stdout


def test_live_mode_is_refused_with_a_reason(cli):
    r = run(cli, "-m")
    assert r.returncode == 2 and "librealsense" in r.stderr


def test_fails_loudly_without_a_gpu(cli, gpu_present):
    if gpu_present:
        pytest.skip("GPU present")
    r = run(cli, "-f", "synth:64x48", "-m")
    assert r.returncode == 1
    assert "no CPU fallback" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("extra,flags,ds", [([], 0, 1), (["-i"], FLAG_DROP_INVALID, 1),
                                            # -c is the reference's -c -m payload, lane-reversed mask included (:501-502, 519);
                                            # -C gates every point by its own range test
                                            (["-c", "-d", "3"], FLAG_CUTOFF | FLAG_CUTOFF_COMPAT, 3), (["-C"], FLAG_CUTOFF, 1)])
def test_cli_synthetic_matches_oracle(cli, oracle, tmp_path, extra, flags, ds):
    out = str(tmp_path / "stitched.bin")
    r = run(cli, "-f", "synth:640x480", "-m", "-t", "4", "-n", "3", "-r", "3", "-o", out, *extra)
    assert r.returncode == 0, r.stderr
    assert len(re.findall(r"^Frame Time: .* ms FPS: .*Buffer size: .* MBytes$", r.stdout, flags=re.M)) == 3
    for line in ("### Video Frames H x W : 480 x 640", "### # Points : 921600", "### Total Frames = 3",
                 "### AVG Frame Time:", "### AVG FPS:", "### OpenMP Threads : 4", "### AVG Bytes/Frame:",
                 "### AVG Filter Compress Ratio"):
        assert line in r.stdout, line
    frame = 2                                             # the dump is the last frame
    cfgs = [S.synth_stream_config(640, 480, s) for s in range(3)]
    depth = [S.synth_depth(640, 480, s, seed=S.SEED + 7919 * frame) for s in range(3)]
    color = [S.synth_color(640, 480, s, seed=S.SEED + 7919 * frame) for s in range(3)]
    want, _ = oracle.process_frames(cfgs, depth, color, flags, ds)
    raw = np.fromfile(out, dtype=np.uint8)
    assert int.from_bytes(raw[:4].tobytes(), "little") in (0, want.nbytes)     # header only written under -s
    got = raw[4:4 + want.nbytes].view(np.int16).reshape(-1, 5)
    assert got.shape == want.shape and (got == want).all()


@pytest.mark.gpu
def test_cli_raw_dump_single_stream_uses_tf_mat(cli, oracle, tmp_path):
    cfgs, depth, color = S.synth_frame_set(1, 128, 96, single=True)
    path = str(tmp_path / "f.pcsraw")
    S.write_pcsraw(path, cfgs, [(depth, color), (depth, color)])
    out = str(tmp_path / "o.bin")
    r = run(cli, "-f", path, "-m", "-o", out)
    assert r.returncode == 0, r.stderr
    assert "### Total Frames = 2" in r.stdout
    want, _ = oracle.process_frames(cfgs, depth, color)
    got = np.fromfile(out, dtype=np.uint8)[4:4 + want.nbytes].view(np.int16).reshape(-1, 5)
    assert (got == want).all()


@pytest.mark.gpu
def test_cli_loads_extrinsics_file(cli, oracle, tmp_path):
    from pointcloud_stitching_amd import calibration as cal
    from pointcloud_stitching_amd.types import TRANSFORMS
    mats = [TRANSFORMS[5].reshape(4, 4), TRANSFORMS[2].reshape(4, 4)]
    ext = str(tmp_path / "ext.txt")
    cal.write_extrinsics(ext, mats)
    out = str(tmp_path / "o.bin")
    r = run(cli, "-f", "synth:128x96", "-m", "-n", "2", "-r", "1", "-e", ext, "-o", out)
    assert r.returncode == 0, r.stderr
    cfgs = [S.synth_stream_config(128, 96, s) for s in range(2)]
    for s in range(2):
        for k in range(16):
            cfgs[s].cam_to_world[k] = float(np.float32(mats[s].reshape(-1)[k]))
    depth = [S.synth_depth(128, 96, s) for s in range(2)]
    color = [S.synth_color(128, 96, s) for s in range(2)]
    want, _ = oracle.process_frames(cfgs, depth, color)
    got = np.fromfile(out, dtype=np.uint8)[4:4 + want.nbytes].view(np.int16).reshape(-1, 5)
    assert (got == want).all()


def test_cli_rejects_short_extrinsics_file(cli, tmp_path):
    ext = tmp_path / "e.txt"
    ext.write_text("1 0 0 0 0 1 0 0 0 0 1 0 0 0 0 1\n")
    r = run(cli, "-f", "synth:64x48", "-m", "-n", "2", "-e", str(ext))
    assert r.returncode == 2 and "matrices" in r.stderr
