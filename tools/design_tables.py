#!/usr/bin/env python3
"""Prints DESIGN.md section 0's kernel table from the round's committed records — `profiles/<tag>_bench.json` (one `python bench.py`
line), `<tag>_kernel_stats.csv` (rocprofv3 --kernel-trace --stats of the same commands), `<tag>_pmc_summary.json` (separate --pmc passes),
`<tag>_bench_config5_node1.json` / `_node8v.json` — so the document cannot quote a figure no record holds. DESIGN.md keeps the output
between the kernels markers; tests/test_docs.py regenerates it and compares.

    python tools/design_tables.py [tag=r06] [--write]      # --write: replace the block in DESIGN.md
"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN, END = "<!-- kernels:begin -->", "<!-- kernels:end -->"
PEAK = 8000.0


def load_line(path):
    try:
        return json.loads([l for l in open(path) if l.startswith("{")][-1])
    except (OSError, IndexError, ValueError):
        return None


class Records:
    def __init__(self, tag):
        p = os.path.join(ROOT, "profiles")
        self.tag = tag
        self.bench = load_line(os.path.join(p, f"{tag}_bench.json"))
        self.node1 = load_line(os.path.join(p, f"{tag}_bench_config5_node1.json"))
        self.node8 = load_line(os.path.join(p, f"{tag}_bench_config5_node8v.json"))
        self.stats = {}
        try:
            for r in csv.DictReader(open(os.path.join(p, f"{tag}_kernel_stats.csv"))):
                self.stats.setdefault(r["run"], []).append(r)
        except OSError:
            pass
        try:
            self.pmc = json.load(open(os.path.join(p, f"{tag}_pmc_summary.json")))
        except (OSError, ValueError):
            self.pmc = {}

    def prof_us(self, run, pattern):
        """rocprofv3 average duration (us) of the most-called kernel of `run` whose name matches `pattern`; None if absent."""
        rows = [r for r in self.stats.get(run, []) if re.search(pattern, r["kernel"])]
        if not rows:
            return None
        r = max(rows, key=lambda x: int(x["calls"]))
        return float(r["avg_ns"]) / 1e3

    def traffic(self, run, pattern):
        """HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, KiB units) summed over the kernels of `run` that match; None if absent."""
        tot, hit = 0.0, False
        for k, cs in self.pmc.get(run, {}).items():
            if re.search(pattern, k) and "traffic_bytes_per_launch" in cs:
                tot += cs["traffic_bytes_per_launch"]; hit = True
        return tot if hit else None


def f1(x):
    return "—" if x is None else f"{x:.1f}"


def mb(x):
    return "—" if x is None else f"{x / 1e6:.1f} MB"


def ratio(t, a):
    return "—" if (t is None or not a) else f"{t / a:.2f}×"


def frac(nbytes, us):
    return "—" if not us else f"{nbytes / (us * 1e-6) / 1e9 / PEAK:.2f}"


def table(R):
    d = R.bench
    rows = []
    S, W, H = d["config"]["streams_per_gpu"], d["config"]["width"], d["config"]["height"]
    n = S * W * H
    rf = d["roofline"]
    us = rf["avg_launch_ms"] * 1e3
    rows.append(("`pcs_fused_dense_kernel` **(headline, `value`)**", f"{S} × {W}×{H}, one frame-set per launch", f"**{us:.1f}**",
                 f1(R.prof_us("dense", r"fused_dense_kernel<")), f"15 B × {n / 1e6:.2f} M = {mb(15 * n)}", f"**{rf['frac']:.2f}** (wall clock: {rf.get('frac_wall', rf['frac']):.2f})",
                 ratio(R.traffic("dense", r"fused_dense_kernel<false, false, pcs::CertMath<true"), 15 * n),
                 "HBM at the copy line: a plain 16-byte copy of the same volume takes 22.4 µs; only more work per launch moves it"))
    ss = d.get("single_stream")
    if ss:
        a = ss["roofline"]["algorithmic_bytes_per_launch"]
        rows.append(("same kernel, ONE stream per launch (BASELINE `configs[1]`)", f"1 × {W}×{H}, cold ring of {ss['ring_frame_sets']} frames",
                     f"{ss['ms_per_frame'] * 1e3:.2f} (512-point tiles: {ss['tile_512_points_ms'] * 1e3:.2f}"
                     + (f"; frame loop over two contexts: **{ss['frame_loop_two_contexts']['ms_per_frame'] * 1e3:.2f}**" if "frame_loop_two_contexts" in ss else "") + ")",
                     f1(R.prof_us("single", r"fused_dense_kernel<")),
                     mb(a), f"{ss['roofline']['frac']:.2f}" + (f"; loop {ss['frame_loop_two_contexts']['frac']:.2f}" if "frame_loop_two_contexts" in ss else ""), ratio(R.traffic("single", r"fused_dense_kernel<"), a),
                     "latency: a lone launch is a chain of dependent round trips (constants, Z16 + LUT, colour gather, store drain) — ≈ 5 µs before "
                     "the first stream's bytes count, ≈ 2.3 µs per further stream; the tile size does not shorten it (Appendix A)"))
    bd = d.get("batched_dense")
    if bd:
        k = bd["frame_sets_per_launch"]
        pu = R.prof_us("batch", r"fused_dense_batch_kernel<")
        rows.append(("`pcs_fused_dense_batch_kernel`", f"the same, {k} frame-sets per launch", f"{bd['ms_per_frame_set'] * 1e3:.1f} / set",
                     "—" if pu is None else f"{pu:.1f} / {k} sets", f"{mb(15 * n)} / set", f"{bd['frac']:.2f}",
                     ratio(R.traffic("all_legs", r"fused_dense_batch_kernel<"), 15 * n * k), "HBM; fill and drain of the launch amortised"))
    for key, label, note in (("general_rotation", "same kernel, 1° depth→colour rotation", "+15 individually rounded flops per pixel"),
                             ("color_1080p", "same kernel, colour 1920×1080 + rotation + colour distortion (what a D400 rig records)",
                              "gather into a 2.25× larger raster + 40 more rounded flops per pixel; closed (Appendix A)")):
        g = d.get(key)
        if g:
            extra = f"; by the colour lines it must touch: {g['frac_touched_bytes']:.2f}" if "frac_touched_bytes" in g else ""
            rows.append((label, f"{S} × 720p depth" + (" + 1080p colour" if key == "color_1080p" else ""), f"{g['ms_per_step'] * 1e3:.1f}", "—",
                         mb(15 * n), f"{g['frac']:.2f}{extra}",
                         ratio(g.get("pmc_traffic_bytes_per_launch"), 15 * n) if key == "color_1080p" else "—", note))
    c = d.get("compaction")
    if c:
        ab = c["algorithmic_bytes_per_point"] * n
        p3 = [R.prof_us("drop_invalid", p) for p in (r"fused_count_kernel", r"pcs_scan_kernel", r"fused_emit_kernel")]
        rows.append(("`pcs_fused_count` → `pcs_scan` → `pcs_fused_emit` (default ordered compaction)", f"{S} × 720p, {100 * (1 - c['kept_fraction']):.1f} % invalid",
                     f"{c['ms_per_step'] * 1e3:.1f} (counts handed in: {c['caller_counts']['ms_per_step'] * 1e3:.1f}; {c['batched']['frame_sets_per_call']} sets per call: "
                     f"{c['batched']['ms_per_frame_set'] * 1e3:.1f} / set)", " + ".join(f1(x) for x in p3), f"(5 + 10ρ) B × {n / 1e6:.2f} M = {mb(ab)}",
                     f"{c['frac']:.2f} ({c['caller_counts']['frac']:.2f}; {c['batched']['frac']:.2f})",
                     ratio(R.traffic("all_legs", r"fused_count_kernel|pcs_scan_kernel|fused_emit_kernel<true, true"), ab),
                     "three dependent launches, Z16 read twice; closed after three rounds of measured negatives (Appendix A)"))
    pk = d.get("pack_twin")
    if pk:
        one = pk.get("single", {})
        rows.append(("`pcs_pack_batch_kernel` / `pcs_pack_dense_kernel` (a2 twin)", f"{S} clouds per launch / one 720p cloud per call",
                     f"{pk['batched_ms_per_frame_set'] * 1e3:.1f} / {one.get('ms_per_cloud', pk['per_stream_launches_ms_per_frame_set'] / S) * 1e3:.2f}",
                     f"{f1(R.prof_us('pack_batch', r'pack_batch_kernel'))} / {f1(R.prof_us('twin_single', r'pack_dense_kernel'))}", "33 B/pt",
                     f"{pk['batched_frac']:.2f} / {one.get('roofline', {}).get('frac', pk['per_stream_launches_frac']):.2f}",
                     ratio(R.traffic("all_legs", r"pack_batch_kernel"), 33 * n), "HBM / launch latency"))
    ct = d.get("centre_transform")
    if ct:
        rows.append(("`pcs_transform_payload_kernel` (the centre's decode / PCL-order affine / re-encode)", f"{S} packed 720p payloads, one launch",
                     f"{ct['ms_per_frame_set'] * 1e3:.1f}", "—", f"20 B × {n / 1e6:.2f} M = {mb(20 * n)}", f"{ct['frac']:.2f}", "—",
                     "HBM: 10 B in + 10 B out per record through one LDS buffer used twice"))
    c5 = d.get("config5_one_gpu")
    if c5:
        oc = c5["one_call"]
        fe = R.prof_us("voxel_one_call", r"fused_voxel_partials_kernel")
        fe0 = R.prof_us("voxel_one_call_norowc", r"fused_voxel_partials_kernel")
        rd = R.prof_us("voxel_one_call", r"vox_bkt_reduce_kernel")
        rows.append(("`pcs_fused_voxel_partials_kernel` (config 5 front end, warm: partials placed in the buckets' regions)", "16 × 1920×1080 → voxel partials (50 mm)",
                     "—", f1(fe) + ("" if fe0 is None else f" (colour row per pixel: {fe0:.1f})"), mb(5 * c5["points_in"]) + " in", "—",
                     "—", "**VALU** (deprojection + affine + voxel key + LDS hash table per pixel): 2 workgroups per CU at ≤ 128 VGPRs, 72 KiB LDS"))
        rows.append(("`pcs_vox_bkt_reduce_kernel` (warm bucket tail: ONE launch)", "the partials → voxel records in (z, y, x) order", "—", f1(rd), "—", "—", "—",
                     "a latency chain per workgroup; in a frame loop over two contexts it runs beside the next pre-aggregation"))
        rows.append(("**config 5 in one call** (`pcs_process_frames_voxel_device`)", "16 × 1080p → voxel cloud, digest-checked",
                     f"**{oc['ms_per_frame_set'] * 1e3:.0f}**" + (f"; frame loop over two contexts **{oc['frame_loop_two_contexts_ms_per_frame_set'] * 1e3:.0f}**"
                                                                 if "frame_loop_two_contexts_ms_per_frame_set" in oc else "")
                     + f" (cold chain {oc['cold_chain_ms_per_frame_set'] * 1e3:.0f}, LSD {oc['lsd_tail_ms_per_frame_set'] * 1e3:.0f})", "—",
                     mb(oc["algorithmic_bytes"]), f"{oc['frac']:.2f}" + (f"; loop {oc['frame_loop_frac']:.2f}" if "frame_loop_frac" in oc else ""),
                     ratio(R.traffic("voxel_one_call", r"fused_voxel_partials_kernel|vox_bkt_reduce_kernel"), oc["algorithmic_bytes"]),
                     "the two rows above: serial on one context, overlapped on two"))
    if R.node1 or R.node8:
        a = f"{R.node1['ms_per_step'] * 1e3:.0f}" if R.node1 else "—"
        b = f"{R.node8['ms_per_step'] * 1e3:.0f}" if R.node8 else "—"
        op = (R.node1 or {}).get("one_peer", {})
        sink = bool((R.node8 or {}).get("voxel_sink"))
        xr = ((R.node8 or {}).get("same_gpu_peers") or {}).get("partials_exchange", {})
        rows.append(("**config 5 through the node** (`pcs_node_submit_voxel_device` / `pcs_node_wait_voxel`)",
                     "one peer / 8 virtual peers of ONE GPU" + (" (through the voxel sinks; the partials exchange on RCCL self send/recv beside it)" if sink
                                                                 else " (RCCL self send/recv)"),
                     f"**{a}** / **{b}**" + (f" (exchange route {xr['ms_per_step'] * 1e3:.0f})" if xr else "") + " per frame-set"
                     + (f" (one peer on one context {op['one_context_ms_per_step'] * 1e3:.0f}, partials pipeline {op['partials_pipeline_ms_per_step'] * 1e3:.0f})"
                        if "one_context_ms_per_step" in op else ""), "—", "—", "—", "—",
                     "one peer: the one call above, the two slots on two contexts in turn. 8 virtual peers: 8 contexts pre-aggregate into a sink of the "
                     "GPU they share (§9), the tail of k beside the pre-aggregations of k+1; the exchange route runs every peer's kernels, RCCL's "
                     "self-copies and the root's place + reduce one after the other on that GPU. Neither says anything about a node of 8 (never "
                     "measured across GPUs)"))
    head = ["kernel(s)", "workload", "µs, un-profiled (`bench.py`)", "µs, `rocprofv3` avg", "algorithmic bytes", "of 8 TB/s", "PMC traffic", "what bounds it"]
    out = [BEGIN, f"(generated by `python tools/design_tables.py {R.tag} --write` from `profiles/{R.tag}_bench.json`, `{R.tag}_kernel_stats.csv`, "
                  f"`{R.tag}_pmc_summary.json`, `{R.tag}_bench_config5_node*.json`)", "", "| " + " | ".join(head) + " |", "|" + "---|" * len(head)]
    out += ["| " + " | ".join(r) + " |" for r in rows]
    out.append(END)
    return "\n".join(out)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    tag = args[0] if args else "r06"
    text = table(Records(tag))
    if "--write" in sys.argv:
        p = os.path.join(ROOT, "DESIGN.md")
        s = open(p).read()
        a, b = s.index(BEGIN), s.index(END) + len(END)
        open(p, "w").write(s[:a] + text + s[b:])
    else:
        print(text)


if __name__ == "__main__":
    main()
