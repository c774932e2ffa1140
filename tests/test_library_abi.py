"""The C-ABI library builds, loads and exports exactly what include/pcs_hip.h declares.
No compute calls here (no GPU in this tier)."""
import os
import re
import subprocess

import pytest

from pointcloud_stitching_amd import lib as L
from pointcloud_stitching_amd.api import PcsContext, PcsError
from pointcloud_stitching_amd import synthetic as S

HEADER = os.path.join(L.INCLUDE_DIR, "pcs_hip.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pcs_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_loads():
    path = L.build()
    assert os.path.exists(path)
    lib = L.load()
    assert lib.pcs_abi_version() == 1


def test_every_declared_symbol_is_exported_and_bound():
    declared = header_functions()
    assert len(declared) >= 25
    nm = subprocess.run(["nm", "-D", "--defined-only", L.LIB_PATH], stdout=subprocess.PIPE, text=True, check=True).stdout
    exported = set(re.findall(r" T (pcs_[a-z0-9_]+)", nm))
    assert set(declared) <= exported, sorted(set(declared) - exported)
    bound = {name for name, _, _ in L.SYMBOLS}
    assert set(declared) == bound, (sorted(set(declared) - bound), sorted(bound - set(declared)))


def test_no_torch_types_or_cxx_in_the_header():
    src = open(HEADER).read()
    assert "torch" not in src and "at::" not in src and "std::" not in src
    assert 'extern "C"' in src


def test_product_does_not_use_the_oracle():
    """Nothing under pointcloud_stitching_amd/ may include, import, link or execute oracle/."""
    pkg = os.path.dirname(os.path.dirname(L.LIB_PATH))
    pats = [r'#\s*include\s*[<"][^>"]*oracle', r'^\s*(from|import)\s+oracle\b', r'libpcs_oracle', r'-lpcs_oracle',
            r'pcs_oracle_[a-z_]+\s*\(', r'oracle/']
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".c")) or f == "Makefile":
                text = open(os.path.join(root, f), errors="ignore").read()
                code = "\n".join(l for l in text.splitlines()
                                 if not l.lstrip().startswith(("//", "#  ", "* ", "/*")) or "include" in l)
                for p in pats:
                    assert not re.search(p, code, flags=re.M), (os.path.join(root, f), p)
    nm = subprocess.run(["nm", "-D", L.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
    assert "oracle" not in nm


def test_strerror_and_create_without_gpu(gpu_present):
    lib = L.load()
    assert lib.pcs_strerror(0) == b"ok"
    assert b"no CPU fallback" in lib.pcs_strerror(-2) or b"fallback" in lib.pcs_strerror(-2)
    if gpu_present:
        pytest.skip("a GPU is present; the no-device path cannot be shown")
    cfgs, _, _ = S.synth_frame_set(1, 64, 48, single=True)
    with pytest.raises(PcsError) as e:
        PcsContext(cfgs)
    assert e.value.status == -2          # fails loudly: no CPU fallback in the product


def test_create_rejects_bad_configs_before_touching_the_device():
    cfgs, _, _ = S.synth_frame_set(1, 64, 48, single=True)
    with pytest.raises(PcsError) as e:
        PcsContext(cfgs, downsample=0)
    assert e.value.status == -1
    with pytest.raises(PcsError) as e:
        PcsContext(cfgs, flags=0x2)       # COMPAT without CUTOFF
    assert e.value.status == -1
    cfgs[0].color_bpp = 2
    with pytest.raises(PcsError) as e:
        PcsContext(cfgs)
    assert e.value.status == -4
    cfgs, _, _ = S.synth_frame_set(1, 64, 48, single=True)
    cfgs[0].depth.model = 4
    cfgs[0].depth.coeffs[0] = 0.1
    with pytest.raises(PcsError) as e:
        PcsContext(cfgs)
    assert e.value.status == -4


def test_node_library_exports_its_header():
    node_h = os.path.join(L.INCLUDE_DIR, "pcs_node.h")
    src = re.sub(r"/\*.*?\*/", "", open(node_h).read(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(pcs_node_[a-z0-9_]+)\s*\(", src)))
    lib = os.path.join(os.path.dirname(L.LIB_PATH), "libpcs_node.so")
    L.build()
    assert os.path.exists(lib)
    nm = subprocess.run(["nm", "-D", "--defined-only", lib], stdout=subprocess.PIPE, text=True, check=True).stdout
    exported = set(re.findall(r" T (pcs_node_[a-z0-9_]+)", nm))
    assert set(declared) == exported and len(declared) >= 6


def test_build_falls_back_to_the_shipped_library_when_make_cannot_run(monkeypatch):
    """A read-only install (no lock file can be opened) or a box without make: the shipped library is loaded with a warning; with
    no library either, the error is PcsBuildError — not a bare OSError."""
    import builtins
    import subprocess
    from pointcloud_stitching_amd import lib as L
    L.build()                                                        # make sure the library exists

    def no_make(*a, **k):
        raise FileNotFoundError(2, "No such file or directory: 'make'")
    monkeypatch.setattr(subprocess, "run", no_make)
    with pytest.warns(RuntimeWarning, match="as shipped"):
        assert L.build() == L.LIB_PATH
    with pytest.raises(L.PcsBuildError):                             # a forced rebuild cannot be satisfied by what is there
        L.build(force=True)
    monkeypatch.undo()

    real_open = builtins.open

    def read_only(path, mode="r", *a, **k):
        if str(path).endswith(".build.lock"):
            raise PermissionError(13, "Read-only file system")
        return real_open(path, mode, *a, **k)
    monkeypatch.setattr(builtins, "open", read_only)
    with pytest.warns(RuntimeWarning, match="as shipped"):
        assert L.build() == L.LIB_PATH
    monkeypatch.setattr(L, "LIB_PATH", L.LIB_PATH + ".absent")
    with pytest.raises(L.PcsBuildError):
        L.build()
