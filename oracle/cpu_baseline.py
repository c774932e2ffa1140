"""The timed CPU sample of bench.py's `cpu_baseline` leg, run as a CHILD PROCESS.  TEST INFRASTRUCTURE ONLY.

    python -m oracle.cpu_baseline --width 1280 --height 720 --streams 8 --seconds 12

prints one JSON object. What is timed: oracle/pcs_oracle_simd.c — the SSE/FMA + OpenMP port of the reference's
`-m -t<N>` path — in the reference's own bracket (`sendXYZRGBPointcloud` only: memset + pack, deprojection excluded,
src/pcs-camera-optimized.cpp:291-293; work-sharing :413), `streams` frames back to back = one frame-set.

Why a child process, and what makes the figure reproducible (round 4's driver lines moved 5 384 -> 4 036 Mpoints/s on
one CPU model with a flat -t1 figure):
  * the parent puts OMP_PROC_BIND=close / OMP_PLACES=cores into the child's environment, so libgomp binds the team
    when it initialises (an environment variable set after some other library loaded libgomp would be ignored);
  * the child imports numpy and the oracle only — no torch thread pools, no HIP runtime threads beside the team;
  * it runs BEFORE any GPU leg of bench.py;
  * vertices, texcoords, colour and the output buffer are freshly mapped per thread count and FIRST TOUCHED BY THE
    TEAM with the pack loop's own work-sharing (pcs_oracle_place_omp), so pages live on the node of the core that uses them;
  * `value` is the MEDIAN over >= 30 passes at the best thread count; best / p10 / p90 ride beside it.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _physical_cores(allowed=None):
    """Distinct (socket, core) pairs of /proc/cpuinfo among the CPUs this process may run on."""
    try:
        pairs, cpu, phys, core = set(), None, None, None

        def flush():
            if phys is not None and core is not None and (allowed is None or cpu in allowed):
                pairs.add((phys, core))
        for line in open("/proc/cpuinfo"):
            if line.startswith("processor"):
                cpu = int(line.split(":", 1)[1])
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                flush()
                cpu = phys = core = None
        flush()
        return len(pairs) or None
    except (OSError, ValueError):
        return None


def _pct(sorted_vals, q):
    return sorted_vals[min(len(sorted_vals) - 1, max(0, int(round(q * (len(sorted_vals) - 1)))))]


def sample(width, height, streams, seconds, min_passes=30, camera0=0):
    # the CPUs this process may use — read BEFORE libgomp initialises: with OMP_PROC_BIND set it binds the calling (master) thread
    # to the first place, and sched_getaffinity would then report that one core
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:
        allowed = set(range(os.cpu_count() or 1))
    from oracle import pcs_oracle as O
    from pointcloud_stitching_amd import synthetic as Syn
    L = O.lib()
    avail = len(allowed)
    phys = _physical_cores(allowed) or avail
    S, npts = streams, width * height
    cfgs = [Syn.synth_stream_config(width, height, camera0 + s) for s in range(S)]
    depth = [Syn.synth_depth(width, height, camera0 + s) for s in range(S)]
    color = [np.ascontiguousarray(Syn.synth_color(width, height, camera0 + s)).reshape(-1) for s in range(S)]
    src_vt = [O.deproject(cfgs[s], depth[s]) for s in range(S)]
    depth_flat = [np.ascontiguousarray(d).reshape(-1) for d in depth]
    # the reference's 5 000 000-short buffer (:157) holds 999 999 points; larger frames (1080p) would overflow it there
    buf_shorts = max(5_000_000, 2 + 5 * npts)
    bpp = cfgs[0].color_bpp
    color_px = color[0].size // bpp

    def placed(threads):
        """Fresh mappings, first touched by a team of `threads` with the pack loop's work-sharing."""
        v, t, c = [], [], []
        for s in range(S):
            a = np.empty((npts, 3), np.float32); L.pcs_oracle_place_omp(a.ctypes.data, src_vt[s][0].ctypes.data, npts, 12, threads)
            b = np.empty((npts, 2), np.float32); L.pcs_oracle_place_omp(b.ctypes.data, src_vt[s][1].ctypes.data, npts, 8, threads)
            k = np.empty(color[s].size, np.uint8); L.pcs_oracle_place_omp(k.ctypes.data, color[s].ctypes.data, color_px, bpp, threads)
            v.append(a); t.append(b); c.append(k)
        buf = np.empty(buf_shorts, np.int16)
        L.pcs_oracle_place_omp(buf.ctypes.data, None, buf_shorts // 5, 10, threads)
        vv = np.empty((npts, 3), np.float32); L.pcs_oracle_place_omp(vv.ctypes.data, None, npts, 12, threads)
        tt = np.empty((npts, 2), np.float32); L.pcs_oracle_place_omp(tt.ctypes.data, None, npts, 8, threads)
        return v, t, c, buf, vv, tt

    def make_runs(threads):
        v, t, c, buf, vv, tt = placed(threads)

        def send(s, vs, ts):
            if L.pcs_oracle_send_simd_omp(C.byref(cfgs[s]), vs.ctypes.data, ts.ctypes.data, npts, c[s].ctypes.data,
                                          buf.ctypes.data, buf_shorts, threads) < 0:
                raise RuntimeError("cpu baseline: buffer too small")

        def run_a():                      # bracket A: the reference's timed region, S frames back to back
            for s in range(S):
                send(s, v[s], t[s])

        def run_b():                      # bracket B: with the CPU deprojection = what the fused GPU kernel does
            for s in range(S):
                L.pcs_oracle_deproject_omp(C.byref(cfgs[s]), depth_flat[s].ctypes.data, vv.ctypes.data, tt.ctypes.data, threads)
                send(s, vv, tt)
        return run_a, run_b, (v, t, c, buf, vv, tt)

    def passes(fn, share, least):
        fn(); fn()                                           # warm
        t_end = time.perf_counter() + share
        out = []
        while time.perf_counter() < t_end or len(out) < least:
            t0 = time.perf_counter(); fn(); out.append(time.perf_counter() - t0)
        return sorted(out)

    # The reference's schedule(static,10000) over its four-point iterations yields 24 chunks per 720p frame
    # (230 400 / 10 000), so more than 24 threads cannot help it; the sweep runs up to the PHYSICAL core count.
    cand = (1, 2, 4, 8, 12, 16, 24, 32, 48, 64)
    sweep = sorted({t for t in cand if t <= max(min(avail, phys), 1)})
    pts = S * npts
    # 1/3 of the budget: a short sweep that only PICKS the thread count (median of >= 5 passes each, both brackets)
    share = seconds / 3.0 / (2.0 * len(sweep))
    med_a, med_b = {}, {}
    for th in sweep:
        ra, rb, keep = make_runs(th)
        pa = passes(ra, share, 5); med_a[th] = _pct(pa, 0.5)
        pb = passes(rb, share, 5); med_b[th] = _pct(pb, 0.5)
        del keep
    ta = min(med_a, key=med_a.get); tb = min(med_b, key=med_b.get)
    # 2/3: the reported samples
    ra, _, keep_a = make_runs(ta)
    cpus = (C.c_int * ta)(); L.pcs_oracle_team_cpus(cpus, ta)
    A = passes(ra, seconds / 3.0, min_passes)
    r1, rb1, keep_1 = make_runs(1)
    A1 = passes(r1, seconds / 9.0, 5)
    B1 = passes(rb1, seconds / 9.0, 5)
    del keep_1
    _, rb, keep_b = make_runs(tb)
    B = passes(rb, seconds / 9.0, min_passes)
    med = _pct(A, 0.5)
    return {
        "value": round(pts / med / 1e6, 2), "unit": "Mpoints/s", "cores": ta, "kind": "port",
        "sample": f"{S} x {width}x{height} frames back-to-back; MEDIAN of {len(A)} passes at -t{ta} (picked by a sweep over "
                  f"-t{sweep}); bracket A = the reference's timed region (memset + pack, deprojection excluded), SSE/FMA + "
                  f"OpenMP port of the -m -t<N> path; child process, team bound (OMP_PROC_BIND="
                  f"{os.environ.get('OMP_PROC_BIND', 'unset')}, OMP_PLACES={os.environ.get('OMP_PLACES', 'unset')}), buffers "
                  f"first-touched by the team, run before any GPU leg",
        "statistic": "median",
        "passes": len(A),
        "ms_per_frame_set": round(med * 1e3, 3),
        "best_value": round(pts / A[0] / 1e6, 2),
        "p10_value": round(pts / _pct(A, 0.9) / 1e6, 2),          # slow tail: 90th percentile of TIME
        "p90_value": round(pts / _pct(A, 0.1) / 1e6, 2),
        "theoretical_fps_per_stream": round(S / med, 1),
        "t1_value": round(pts / _pct(A1, 0.5) / 1e6, 2),
        "by_threads": {str(t): round(pts / med_a[t] / 1e6, 1) for t in sweep},
        "with_deprojection_value": round(pts / _pct(B, 0.5) / 1e6, 2),
        "with_deprojection_cores": tb,
        "with_deprojection_t1_value": round(pts / _pct(B1, 0.5) / 1e6, 2),
        "host_physical_cores": phys,
        "host_logical_cpus": avail,
        "team_cpus": sorted(int(x) for x in cpus),
        "cpu_model": _cpu_model(),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--streams", type=int, default=8)
    ap.add_argument("--seconds", type=float, default=12.0)
    ap.add_argument("--min-passes", type=int, default=30)
    a = ap.parse_args()
    print(json.dumps(sample(a.width, a.height, a.streams, a.seconds, a.min_passes)), flush=True)


if __name__ == "__main__":
    sys.exit(main() or 0)
