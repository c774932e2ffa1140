"""The bench rig of the single-GPU (and ranks-route) legs: contexts, the ring of frame-sets resident in HBM, the launch forms.

Everything a leg needs hangs on ONE object, so that a leg is a function of (rig) that returns its dict (benchlegs/legs_*.py) and
bench.py only assembles the line. ALL launch forms take their ring slot from one monotonically increasing counter that runs
through pre-heat, warm-up and every timed region: a slot is never re-read before at least 2 x 256 MiB of other inputs went by."""
import ctypes as C
import time

import numpy as np

from .common import ALGO_BYTES_PER_POINT, INFINITY_CACHE_BYTES

VP = C.c_void_p


def up(nbytes):
    """slab carving granularity: 256 bytes (power-of-two aligned per-raster allocations alias in the Infinity Cache)"""
    return (nbytes + 16 + 255) & ~255


class Rig:
    def __init__(self, args, rank, world, local_rank):
        import torch
        from pointcloud_stitching_amd import synthetic as Syn
        from pointcloud_stitching_amd.api import PcsContext
        from pointcloud_stitching_amd.types import POINT_SHORTS, FLAG_CUTOFF, FLAG_DROP_INVALID

        self.args, self.rank, self.world, self.local_rank = args, rank, world, local_rank
        self.torch, self.Syn, self.PcsContext = torch, Syn, PcsContext
        self.dev = dev = torch.device("cuda", local_rank)
        W, H = args.width, args.height
        self.W, self.H = W, H
        self.strong = args.scaling == "strong"
        if self.strong:
            if args.streams % world:
                raise SystemExit(f"--scaling strong shards {args.streams} streams over {world} GPUs: not divisible")
            S = args.streams // world            # cameras on this GPU
            self.total_streams = args.streams
        else:
            S = args.streams
            self.total_streams = args.streams * world
        self.S = S
        self.in_bytes_per_set = S * (W * H * 2 + W * H * 3)
        # a slot is re-read after R-1 other sets: (R-1) * inputs > 2 x Infinity Cache
        R = max(args.ring, 2) if args.ring else max(4, -(-2 * INFINITY_CACHE_BYTES // self.in_bytes_per_set) + 2)
        self.R = R
        self.ring_cold = (R - 1) * self.in_bytes_per_set >= 2 * INFINITY_CACHE_BYTES
        if not args.ring:
            assert self.ring_cold, "default ring must keep every re-read >= 2 x 256 MiB of input traffic apart"
        self.npts = npts = W * H
        self.set_points = S * npts
        # global camera index = rank*S + s  -> extrinsic transform[(rank*S+s) % 8], distinct seeds per camera
        self.cfgs = cfgs = [Syn.synth_stream_config(W, H, rank * S + s) for s in range(S)]
        self.mode_flags = {"drop_invalid": FLAG_DROP_INVALID, "batch_drop_invalid": FLAG_DROP_INVALID,
                           "cutoff": FLAG_CUTOFF}.get(args.mode, 0)
        self.ctx = ctx = PcsContext(cfgs, device=local_rank, flags=self.mode_flags)
        # One explicit HIP stream for everything this rank enqueues: the library's kernels (pcs_set_stream) and torch's own
        # work — the RCCL gather orders itself against torch's CURRENT stream. (torch's default stream has the handle 0, which
        # pcs_set_stream reads as "use the context's own stream": the kernels and the gather would then be unordered.)
        self.stream = stream = torch.cuda.Stream(dev)
        torch.cuda.set_stream(stream)
        assert stream.cuda_stream != 0
        ctx.set_stream(stream.cuda_stream)

        # Ring of frame-sets resident in HBM, carved from ONE slab at 256-byte granularity.
        self.payload_shorts = payload_shorts = self.set_points * POINT_SHORTS
        depth_b, color_b, out_b = up(npts * 2), up(cfgs[0].color_bytes), up(payload_shorts * 2 + 256)
        self.slab = slab = torch.empty(R * (S * (depth_b + color_b) + out_b) + 256, dtype=torch.uint8, device=dev)
        off = (-slab.data_ptr()) % 256
        self.d_depth, self.d_color, self.d_out, self.host0, self.host1 = [], [], [], None, None
        DISTINCT = 4          # frame-sets generated on the host; further ring slots are device copies of these (distinct
        for slot in range(R):  # ADDRESSES are what defeats the caches; generating 16 sets in numpy would only cost start-up time)
            if slot < DISTINCT:
                dep = [Syn.synth_depth(W, H, rank * S + s, seed=Syn.SEED + 7919 * slot) for s in range(S)]
                col = [Syn.synth_color(W, H, rank * S + s, seed=Syn.SEED + 7919 * slot) for s in range(S)]
            if slot == 0:
                self.host0 = (dep, col)
            if slot == 1:
                self.host1 = (dep, col)
            dd, dc = [], []
            for s in range(S):
                v = slab[off:off + npts * 2]
                v.copy_(torch.from_numpy(dep[s].reshape(-1).view(np.uint8)) if slot < DISTINCT else self.d_depth[slot % DISTINCT][s])
                dd.append(v); off += depth_b
                v = slab[off:off + cfgs[0].color_bytes]
                v.copy_(torch.from_numpy(col[s]) if slot < DISTINCT else self.d_color[slot % DISTINCT][s])
                dc.append(v); off += color_b
            self.d_depth.append(dd); self.d_color.append(dc)
            sk = args.payload_skew & ~1
            self.d_out.append(slab[off + sk:off + sk + payload_shorts * 2].view(torch.int16)); off += out_b
        if self.host1 is None:
            self.host1 = self.host0
        self.ring_bytes = R * (self.set_points * ALGO_BYTES_PER_POINT)
        self.kept_frac = float(np.mean([(d != 0).mean() for d in self.host0[0]]))      # rho of the invalid-drop compaction

        self.lib = ctx._lib
        self.h = ctx._h
        self.call_args = []
        for slot in range(R):
            dp = (VP * S)(*[t.data_ptr() for t in self.d_depth[slot]])
            cp = (VP * S)(*[t.data_ptr() for t in self.d_color[slot]])
            self.call_args.append((dp, cp, VP(self.d_out[slot].data_ptr())))
        self.pack_ring = None
        # a context without predicate flags for the legs that need the plain configuration (and for deproject)
        self.ctx0 = ctx if self.mode_flags == 0 else PcsContext(cfgs, device=local_rank)
        if self.ctx0 is not ctx:
            self.ctx0.set_stream(stream.cuda_stream)
        self.counter = 0
        self.d_cnt = torch.zeros(S + 1, dtype=torch.int32, device=dev)
        self.variable = False       # ranks route with a predicate: the launch writes its counts to d_cnt (set by bench.py)

        self.KB = KB = max(1, min(args.batch_sets, R // 2))
        self.batch_args = []
        for g in range(R // KB):
            slots = [g * KB + k for k in range(KB)]
            dp = (VP * (KB * S))(*[t.data_ptr() for sl in slots for t in self.d_depth[sl]])
            cp = (VP * (KB * S))(*[t.data_ptr() for sl in slots for t in self.d_color[sl]])
            pp = (VP * KB)(*[self.d_out[sl].data_ptr() for sl in slots])
            self.batch_args.append((dp, cp, pp))
        self.launch = {"dense": self.launch_dense, "drop_invalid": self.launch_dense, "cutoff": self.launch_dense,
                       "pack": self.launch_pack_single, "pack_batch": self.launch_pack_batch, "batch": self.launch_batch,
                       "batch_drop_invalid": self.launch_batch}[args.mode]
        self.sets_per_launch = KB if args.mode in ("batch", "batch_drop_invalid") else 1

    # ---- plumbing ----------------------------------------------------------------------------------------------------------
    def check(self, rc, handle=None):
        if rc:
            raise RuntimeError(self.lib.pcs_last_error(handle or self.h).decode())

    def next_slot(self, ring=None):
        k = self.counter
        self.counter = k + 1
        return k % (ring or self.R)

    def sync(self):
        self.torch.cuda.synchronize(self.dev)

    def preheat(self, fn, ms):
        t_pre = time.perf_counter()
        while (time.perf_counter() - t_pre) * 1e3 < ms:      # untimed: settle clocks
            for _ in range(50):
                fn()
            self.sync()

    def timed(self, fn, n, c=None):
        """n launches of fn bracketed by a hipEvent pair on the launch stream -> ms per launch."""
        c = c or self.ctx
        c.timer_begin()
        for _ in range(n):
            fn()
        c.timer_end()
        return c.timer_elapsed_ms() / n

    def new_context(self, cfgs=None, flags=0, own_stream=False):
        c = self.PcsContext(cfgs or self.cfgs, device=self.local_rank, flags=flags)
        if not own_stream:
            c.set_stream(self.stream.cuda_stream)
        return c

    # ---- the a2 twin's inputs (vertices 12 B + texcoords 8 B per point), one copy per ring slot -------------------------------
    def build_pack_ring(self):
        if self.pack_ring is not None:
            return self.pack_ring
        from pointcloud_stitching_amd.types import CloudDesc
        torch, S, npts, R = self.torch, self.S, self.npts, self.R
        set_b = S * (up(npts * 12) + up(npts * 8))
        # re-read distance as for the rasters, over everything the kernel reads (vertices + texcoords + colour)
        Rp = max(3, -(-2 * INFINITY_CACHE_BYTES // (set_b + S * self.cfgs[0].color_bytes)) + 2)
        Rp = min(Rp, R)
        vt_slab = torch.empty(Rp * set_b + 256, dtype=torch.uint8, device=self.dev)
        vo = (-vt_slab.data_ptr()) % 256
        per_slot = [[] for _ in range(Rp)]
        for s in range(S):
            v, t = self.ctx0.deproject(s, self.host0[0][s])
            hv = torch.from_numpy(v.reshape(-1).view(np.uint8)); ht = torch.from_numpy(t.reshape(-1).view(np.uint8))
            for slot in range(Rp):
                o = vo + slot * set_b
                dv = vt_slab[o:o + npts * 12]; dv.copy_(hv if slot == 0 else vt_slab[vo:vo + npts * 12])
                dt = vt_slab[o + up(npts * 12):o + up(npts * 12) + npts * 8]
                dt.copy_(ht if slot == 0 else vt_slab[vo + up(npts * 12):vo + up(npts * 12) + npts * 8])
                per_slot[slot].append((dv.data_ptr(), dt.data_ptr()))
            vo += up(npts * 12) + up(npts * 8)
        descs = []
        for slot in range(Rp):
            arr = (CloudDesc * S)()
            for s in range(S):
                arr[s].stream, arr[s].n_points = s, npts
                arr[s].vertices, arr[s].texcoords = per_slot[slot][s]
                arr[s].color = self.d_color[slot][s].data_ptr()
                arr[s].pc_buffer = self.d_out[slot].data_ptr() + s * npts * 10
            descs.append(arr)
        self.pack_ring = {"R": Rp, "slab": vt_slab, "per_slot": per_slot, "descs": descs}
        return self.pack_ring

    # ---- launch forms --------------------------------------------------------------------------------------------------------
    def launch_dense(self, handle=None):
        dp, cp, out = self.call_args[self.next_slot()]
        self.check(self.lib.pcs_process_frames_device(handle or self.h, dp, cp, out, self.payload_shorts,
                                                      VP(self.d_cnt.data_ptr()) if (self.variable and handle is None) else None), handle)

    def launch_pack_single(self):
        pr = self.build_pack_ring()
        slot = self.next_slot(pr["R"])
        h0 = self.ctx0._h
        for s in range(self.S):
            self.check(self.lib.pcs_copy_pointcloud_xyzrgb_to_buffer_device(
                h0, s, VP(pr["per_slot"][slot][s][0]), VP(pr["per_slot"][slot][s][1]), self.npts,
                VP(self.d_color[slot][s].data_ptr()), VP(self.d_out[slot].data_ptr() + s * self.npts * 10), None), h0)

    def launch_pack_batch(self):
        pr = self.build_pack_ring()
        slot = self.next_slot(pr["R"])
        self.check(self.lib.pcs_copy_pointclouds_xyzrgb_to_buffer_device(self.ctx0._h, self.S, pr["descs"][slot], None), self.ctx0._h)

    def launch_batch(self, c=None):
        c = c or (self.ctx if self.args.mode == "batch_drop_invalid" else self.ctx0)
        dp, cp, pp = self.batch_args[self.next_slot(len(self.batch_args))]
        self.check(self.lib.pcs_process_frames_device_batch(c._h, self.KB, dp, cp, pp, self.payload_shorts, None), c._h)

    def close(self):
        if self.ctx0 is not self.ctx:
            self.ctx0.close()
        self.ctx.close()
