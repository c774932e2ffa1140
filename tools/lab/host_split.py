"""Host-pointer API (pcs_process_frames, PCIe both ways) by buffer kind — run on the GPU box:
    python tools/host_split.py
Shows that long-lived pageable buffers cost the same as page-locked ones (2.25 vs 2.19 ms per 8x720p
frame-set on MI355X / ROCm 7.2) and that a freshly allocated output buffer per call is what is slow
(first-touch page faults, ~7.5 ms). DESIGN.md section 11."""
import time, numpy as np, sys
sys.path.insert(0, '.')
from pointcloud_stitching_amd.api import PcsContext
from pointcloud_stitching_amd import synthetic as S
cfgs, depth, color = S.synth_frame_set(8, 1280, 720)
ctx = PcsContext(cfgs)
n = sum(c.n_points for c in cfgs)
pd = [ctx.host_array(d.shape, np.uint16) for d in depth]
pc = [ctx.host_array(c.shape, np.uint8) for c in color]
for a, b in zip(pd + pc, depth + color): a[...] = b
po = ctx.host_array((2 + n * 5,), np.int16)
pg = np.empty(2 + n * 5, np.int16)
def t(dep, col, out, reps=5):
    ctx.process_frames(dep, col, out=out)
    t0 = time.perf_counter()
    for _ in range(reps): ctx.process_frames(dep, col, out=out)
    return (time.perf_counter() - t0) / reps * 1e3
print("pageable in, pageable out: %.2f ms" % t(depth, color, pg))
print("pageable in, pinned out  : %.2f ms" % t(depth, color, po))
print("pinned in,   pageable out: %.2f ms" % t(pd, pc, pg))
print("pinned in,   pinned out  : %.2f ms" % t(pd, pc, po))
# raw numpy memcpy speed single thread
big = np.empty(73_728_000, np.uint8); src = np.frombuffer(po[2:].tobytes(), np.uint8)[:73_728_000]
t0 = time.perf_counter(); big[:] = src; dt = time.perf_counter() - t0
print("numpy memcpy 73.7MB: %.2f ms (%.1f GB/s)" % (dt * 1e3, 73.728e-3 / dt))
def t_fresh(dep, col, reps=5):
    ctx.process_frames(dep, col, out=None)
    t0 = time.perf_counter()
    for _ in range(reps): ctx.process_frames(dep, col, out=None)
    return (time.perf_counter() - t0) / reps * 1e3
print("pageable in, FRESH pageable out each call: %.2f ms" % t_fresh(depth, color))
