"""Calibration tool against vectors produced by the reference's own Python script
(tests/golden/calibration_golden.json, made by tests/golden/make_calibration_golden.py) and against the
expected matrices the reference records in calibration/pcs4.csv:14-24."""
import json
import os

import numpy as np
import pytest

from pointcloud_stitching_amd import calibration as cal

GOLD = os.path.join(os.path.dirname(__file__), "golden", "calibration_golden.json")


@pytest.fixture(scope="module")
def gold():
    with open(GOLD) as f:
        return json.load(f)


def points_of(case):
    pts = {}
    for x, y, z, label in case["points"]:
        pts.setdefault(label, np.array([x, y, z], np.float64))
    return pts


def test_transforms_match_the_reference_script(gold):
    n = 0
    for case in gold["cases"]:
        pts = points_of(case)
        for cam in case["cameras"]:
            m = cal.camera_transform(pts, cam["label"], cam["prefix"])
            assert np.array_equal(m, np.array(cam["transform"])), (case["file"], cam["label"])   # same float64 ops
            assert cal.format_transform(0, m) == cam["printed"]
            n += 1
    assert n == 10


def test_pcs4_recorded_expectations(gold):
    case = [c for c in gold["cases"] if c["file"] == "pcs4.csv"][0]
    pts = points_of(case)
    for name, label, prefix in (("dextro", "DEXTRO", "D"), ("levo", "LEVO", "L")):
        m = cal.camera_transform(pts, label, prefix)
        assert np.allclose(m, np.array(gold["pcs4_recorded"][name]), atol=5e-9, rtol=0)


def test_rotations_are_orthonormal_and_files_round_trip(gold, tmp_path):
    case = gold["cases"][0]
    pts = points_of(case)
    mats = [cal.camera_transform(pts, c["label"], c["prefix"]) for c in case["cameras"]]
    for m in mats:
        R = m[:3, :3]
        assert np.allclose(R.T @ R, np.eye(3), atol=1e-12)
        assert abs(np.linalg.det(R) - 1) < 1e-12
    p = str(tmp_path / "ext.txt")
    cal.write_extrinsics(p, mats)
    back = cal.read_extrinsics(p)
    assert len(back) == len(mats)
    for a, b in zip(back, mats):
        assert np.allclose(a, b, atol=1e-7)


def test_cli_and_csv_loader(tmp_path, capsys, gold):
    case = gold["cases"][1]
    csvp = tmp_path / "s.csv"
    with open(csvp, "w") as f:
        f.write("1,0.000,0.000,0.000,\n")
        for i, (x, y, z, label) in enumerate(case["points"]):
            f.write(f"{i + 2},{x:.3f},{y:.3f},{z:.3f},{label}\n")
        f.write("\n\ndextro\n-0.99 0.02 -0.08 0.02\n")            # trailing junk like pcs4.csv has
    out = tmp_path / "e.txt"
    assert cal.main([str(csvp), "--cameras", "DEXTRO:D", "LEVO:L", "-o", str(out)]) == 0
    text = capsys.readouterr().out
    assert "transform[0] << [[-0.99574067" in text and "transform[1] << [[ 0.99056815" in text
    assert len(cal.read_extrinsics(str(out))) == 2
    with pytest.raises(KeyError):
        cal.camera_transform(cal.load_survey_csv(str(csvp)), "Z")
