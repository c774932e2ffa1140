#!/usr/bin/env python3
"""Copy what is kept of one tools/final_run.sh record (gpurun_out/final_<tag>/, gpurun_out/prof_<tag>/) into profiles/ under
<prefix>_*: kernel stats, PMC summary, traffic.json, the bench lines of every route, one soak log (GPU test count, soaks, voxel
pipeline by leaf, stale-splitter sequence).      python tools/keep_record.py <tag> [prefix=r05]"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
prefix = sys.argv[2] if len(sys.argv) > 2 else "r05"
fin = os.path.join(ROOT, "gpurun_out", f"final_{tag}")
prof = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
out = os.path.join(ROOT, "profiles")
noise = ("amdgpu.ids", "RCCL version", "HIP version", "ROCm version", "Hostname", "Librccl")


def lines(path):
    return [l.rstrip("\n") for l in open(path) if not any(n in l for n in noise)] if os.path.exists(path) else []


shutil.copy(os.path.join(prof, "kernel_stats.csv"), os.path.join(out, f"{prefix}_kernel_stats.csv"))
shutil.copy(os.path.join(prof, "pmc_summary.json"), os.path.join(out, f"{prefix}_pmc_summary.json"))
shutil.copy(os.path.join(prof, "traffic.json"), os.path.join(out, "traffic.json"))
for f in sorted(os.listdir(fin)):
    if f.startswith("bench") and f.endswith(".json"):
        body = [l for l in lines(os.path.join(fin, f)) if l.startswith("{")]
        if body:
            open(os.path.join(out, f"{prefix}_{f}"), "w").write(body[-1] + "\n")
head = subprocess.run(["git", "log", "-1", "--format=%h"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
log = [f"# tools/final_run.sh {tag} on the round's final kernels (commit {head})"]
log += lines(os.path.join(fin, "pytest.txt")) + lines(os.path.join(fin, "soak.log"))
log += ["# tools/voxel_probe.py, one call from the rasters, 16 x 1080p: as shipped / PCS_VOXEL_REGIONS=0 (bucket tail held to its "
        "cold chain) / PCS_VOXEL_TAIL=lsd"] + lines(os.path.join(fin, "voxel_by_leaf.txt"))
log += ["# tools/lab/bkt_cliff.py (40 mm, scattered points): cloud A, A, B (inside one of A's key ranges), B, B, A, A"]
log += lines(os.path.join(fin, "bkt_cliff.txt"))
open(os.path.join(out, f"{prefix}_soak.log"), "w").write("\n".join(log) + "\n")
print("kept", tag, "as", prefix, "at", head)
