#!/usr/bin/env python3
"""Print VGPR / spill / occupancy / LDS of every kernel in pcs_kernels.hip (hipcc -Rpass-analysis=kernel-resource-usage).
Usage: python tools/lab/kernel_resources.py [file.hip] [--voxel-tu] [substring ...]
(--voxel-tu: pcs_kernels.hip as its second translation unit, the voxel readers, with the flags the Makefile gives it)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "pointcloud_stitching_amd", "csrc")
src = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".hip") else "pcs_kernels.hip"
voxel_tu = "--voxel-tu" in sys.argv
pats = [a for a in sys.argv[1:] if not a.endswith(".hip") and a != "--voxel-tu"]
flags = "-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt".split()
if src == "pcs_kernels.hip" and voxel_tu:
    flags += ["-mllvm", "-amdgpu-kernarg-preload-count=16", "-DPCS_TU_VOXEL=1"]
elif src == "pcs_kernels.hip":
    flags += ["-fno-slp-vectorize", "-mllvm", "-amdgpu-kernarg-preload-count=16"]
r = subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"],
                   cwd=CSRC, capture_output=True, text=True)
blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
names = [b.split("\n")[0].strip(" []") for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for b, dn in zip(blocks, dem):
    if pats and not any(p in dn for p in pats):
        continue
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    dn = dn.replace("pcs::(anonymous namespace)::", "").replace("void ", "")
    dn = re.sub(r"\(pcs::StreamParams.*", "", dn)
    print(f"{dn[:100]:100s} VGPR {g('VGPRs'):>3} AGPR {g('AGPRs'):>3} spill {g('VGPRs Spill'):>3} scratch {g('ScratchSize .bytes/lane.'):>4} "
          f"occ {g('Occupancy .waves/SIMD.'):>2} LDS {g('LDS Size .bytes/block.'):>6}")
