"""The give-up path of the voxel pipeline's bucket tail, walked by fault injection.

A bucket workgroup waits (bounded) for the voxel counts of the buckets before it; on a stalled or preempted device it gives up
and the call ends FLAGGED: the device count is -1 and the bytes are not valid (include/pcs_hip.h, pcs_voxel_grid_device).
pcs_inject_voxel_stall / PCS_BKT_INJECT_STALL makes a launch end exactly so — its first workgroup sleeps, the others' wait
bound is 20 us. Every form of the library that reads the count itself must then run the frame-set again on the LSD tail,
latch LSD for that context, and still end on the oracle's bytes; a negative length must never reach a caller's size
arithmetic or the wire (the reference sends `size` as it is: src/pcs-multicamera-client.cpp:394-403)."""
import os
import subprocess

import numpy as np
import pytest

from pointcloud_stitching_amd import synthetic as S
from pointcloud_stitching_amd.api import PcsContext
from pointcloud_stitching_amd.types import FLAG_DROP_INVALID

from test_voxel_grid import random_payload, _upload_rasters
from test_wire import CENTRAL, built, connect, frame_inputs, free_port, read_frame, retry_server_start      # noqa: F401 (built: fixture)

LATCHED = 3      # PCS_VOXEL_TAIL_LSD_LATCHED


@pytest.fixture(autouse=True)
def bucket_tail_and_no_leftover_injection(monkeypatch):
    monkeypatch.setenv("PCS_VOXEL_TAIL", "bucket")          # the latch must beat the environment
    monkeypatch.delenv("PCS_BKT_INJECT_STALL", raising=False)
    yield
    try:
        from pointcloud_stitching_amd import lib
        lib.load().pcs_inject_voxel_stall(0)
    except Exception:
        pass


@pytest.mark.gpu
def test_host_form_runs_a_flagged_call_again_on_the_lsd_tail(oracle):
    p = random_payload(150000, 11, 2500)
    want = oracle.voxel_grid(p, 50)
    cfgs, _, _ = S.synth_frame_set(1, 64, 48)
    with PcsContext(cfgs) as ctx:
        assert ctx.voxel_tail_reruns() == 0
        got = ctx.voxel_grid(p, 50)                          # healthy (cold bucket call)
        assert (got == want).all() and ctx.voxel_tail_reruns() == 0
        for warm in (True, False):                           # a warm (regions) and, on a fresh latch, a cold bucket call flagged
            ctx.inject_voxel_stall(1)
            got = ctx.voxel_grid(p, 50)
            assert got.shape == want.shape and (got == want).all()
            assert ctx.voxel_tail_reruns() == (1 if warm else 2)
            # latched: the injection armed now is not consumed by this context's calls (they take the LSD tail) ...
            ctx.inject_voxel_stall(1)
            got = ctx.voxel_grid(p, 50)
            assert (got == want).all() and ctx.voxel_tail_reruns() == (1 if warm else 2)
            # ... until the caller lifts the latch: the armed stall hits the next (cold again) bucket call
            ctx.set_voxel_tail(0)


@pytest.mark.gpu
def test_device_form_reports_minus_one_and_the_caller_latches(oracle):
    p = random_payload(90000, 12, 2500)
    want = oracle.voxel_grid(p, 60)
    cfgs, _, _ = S.synth_frame_set(1, 64, 48)
    with PcsContext(cfgs) as ctx:
        d_in = ctx.device_malloc(p.nbytes + 64); ctx.memcpy_h2d(d_in, p)
        d_out = ctx.device_malloc(p.nbytes + 64); d_n = ctx.device_malloc(64)
        nv = np.zeros(1, np.int32)
        ctx.inject_voxel_stall(1)
        ctx.voxel_grid_device(d_in, p.shape[0], 60, d_out, p.size, d_n)
        ctx.synchronize(); ctx.memcpy_d2h(nv, d_n)
        assert int(nv[0]) == -1                              # flagged, and it came back (no hang)
        ctx.set_voxel_tail(LATCHED)
        assert ctx.voxel_tail_reruns() == 1
        ctx.voxel_grid_device(d_in, p.shape[0], 60, d_out, p.size, d_n)
        ctx.synchronize(); ctx.memcpy_d2h(nv, d_n)
        assert int(nv[0]) == want.shape[0]
        got = np.empty(want.size, np.int16); ctx.memcpy_d2h(got, d_out)
        assert (got.reshape(-1, 5) == want).all()


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [[0], [0, 0, 0, 0], "one peer, partials pipeline", "four peers, exchange"])
def test_node_wait_runs_a_flagged_frame_set_again(oracle, devices, monkeypatch):
    """submit(k+1); wait(k) with the stall injected into frame-set 2's tail: every frame-set still equals the oracle, the wait
    never returns a negative count, the node says it ran one frame-set again."""
    from pointcloud_stitching_amd.node import PcsNode, VOXEL_PAYLOADS
    from test_node import _upload, _fetch
    if devices == "four peers, exchange":
        devices = [0, 0, 0, 0]
        monkeypatch.setenv("PCS_NODE_VOXEL_SINK", "0")        # ([0, 0, 0, 0] by itself: through the sinks, every peer run again)
    elif isinstance(devices, str):
        devices = [0]
        monkeypatch.setenv("PCS_NODE_ONE_CALL", "0")
    n, w, h, frames, leaf = 4, 320, 240, 6, 40
    cfgs = [S.synth_stream_config(w, h, s) for s in range(n)]
    sets = [([S.synth_depth(w, h, s, seed=S.SEED + 17 * f) for s in range(n)],
             [S.synth_color(w, h, s, seed=S.SEED + 17 * f) for s in range(n)]) for f in range(frames)]
    want = [oracle.voxel_grid(oracle.process_frames(cfgs, d, c, FLAG_DROP_INVALID, 1)[0], leaf) for d, c in sets]
    with PcsNode(cfgs, devices=devices, flags=FLAG_DROP_INVALID) as node, PcsContext(cfgs[:1]) as mem:
        cap = node.max_payload_shorts
        dev_sets = [_upload(mem, d, c) for d, c in sets]
        vox = [mem.device_malloc(cap * 2 + 64) for _ in range(2)]
        tickets = [node.submit_voxel_device(*dev_sets[0], leaf, vox[0], cap)]
        for k in range(1, frames + 1):
            if k == 2:
                mem.inject_voxel_stall(1)        # consumed by the next bucket-tail launch: frame-set 2's (one call: at its submit;
            if k < frames:                       # partials route: when its reduce is enqueued, by submit(3) or wait(2))
                tickets.append(node.submit_voxel_device(*dev_sets[k], leaf, vox[k & 1], cap))
            nv = node.wait_voxel(tickets[k - 1])
            assert nv == want[k - 1].shape[0], k - 1
            assert (_fetch(mem, vox[(k - 1) & 1], nv) == want[k - 1]).all(), k - 1
        assert node.voxel_reruns() == 1
        # the synchronous PAYLOADS route on the root's own context (one-call nodes latched it above; the others flag it here)
        mem.inject_voxel_stall(1)
        nv, _ = node.process_voxel_device(*dev_sets[1], leaf, vox[0], cap, VOXEL_PAYLOADS)
        assert nv == want[1].shape[0] and (_fetch(mem, vox[0], nv) == want[1]).all()
        assert node.voxel_reruns() == (1 if len(devices) == 1 and os.environ.get("PCS_NODE_ONE_CALL") != "0" else 2)


@pytest.mark.gpu
def test_node_host_form_never_returns_a_negative_size(oracle):
    from pointcloud_stitching_amd.node import PcsNode
    cfgs, depth, color = S.synth_frame_set(4, 256, 144)
    want = oracle.voxel_grid(oracle.process_frames(cfgs, depth, color, FLAG_DROP_INVALID, 1)[0], 50)
    with PcsNode(cfgs, devices=[0, 0], flags=FLAG_DROP_INVALID) as node, PcsContext(cfgs[:1]) as mem:
        mem.inject_voxel_stall(1)
        got, stats = node.process_voxel(depth, color, 50)
        assert got.shape == want.shape and (got == want).all() and stats["voxels"] == want.shape[0]
        assert node.voxel_reruns() == 1


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["device call", "node"])
@retry_server_start
def test_central_cli_never_puts_a_negative_length_on_the_wire(oracle, mode):
    """PCS_BKT_INJECT_STALL=1: the first bucket-tail launch of the process ends flagged; the consumer still receives the voxel grid
    of every frame (length prefix = 10 x voxels), and the program says what it did."""
    port = free_port()
    args = [CENTRAL, "-i", "synth:160x120", "-N", "3", "-V", "50", "-p", str(port), "-r", "3", "-Z"] + (["-G", "1"] if mode == "node" else [])
    env = dict(os.environ, PCS_BKT_INJECT_STALL="1", PCS_VOXEL_TAIL="bucket")
    p = subprocess.Popen(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    try:
        sock = connect(port, procs=[p])
        for frame in range(3):
            sock.sendall(b"Z")
            got = read_frame(sock)
            cfgs, depth, color = frame_inputs(3, 160, 120, frame, single=False)
            stitched, _ = oracle.process_frames(cfgs, depth, color, FLAG_DROP_INVALID, 1)
            want = oracle.voxel_grid(stitched, 50)
            assert got.shape == want.shape and (got == want).all(), frame
        sock.close()
        _, err = p.communicate(timeout=60)
        assert p.returncode == 0, err
        if mode == "device call":
            assert "again on the LSD tail" in err
    finally:
        if p.poll() is None:
            p.kill()


@pytest.mark.gpu
def test_pipelined_cli_reports_the_rerun(tmp_path):
    """-P (submit(k+1); wait(k) over libpcs_node) with the first bucket tail of the process flagged: the loop runs to its end and the
    dump of the last frame-set has a positive length that matches its header."""
    dump = tmp_path / "last.bin"
    env = dict(os.environ, PCS_BKT_INJECT_STALL="1", PCS_VOXEL_TAIL="bucket")
    r = subprocess.run([CENTRAL, "-i", "synth:320x240", "-N", "4", "-G", "1", "-P", "-V", "50", "-Z", "-r", "4", "-o", str(dump)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "run again on the LSD tail after a flagged bucket tail: 1" in r.stdout, r.stdout
    raw = dump.read_bytes()
    (size,) = np.frombuffer(raw[:4], np.int32)
    assert size > 0 and size % 10 == 0 and len(raw) == size + 4
