#!/usr/bin/env python3
"""Prints README.md's "Results at a glance" table from ONE recorded bench line (default: profiles/r06_bench.json, the `python
bench.py` line of the round's final build on one MI355X). README holds the output between the results markers;
tests/test_bench_contract.py regenerates it and compares, so the README cannot quote a number no record holds.

    python tools/readme_results.py [bench.json] [--write]      # --write: replace the block in README.md
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN, END = "<!-- results:begin -->", "<!-- results:end -->"


def load(path):
    return json.loads([l for l in open(path) if l.startswith("{")][-1])


def us(ms):
    return f"{ms * 1e3:.1f} µs"


def pct(f):
    return f"{100 * f:.1f} %"


def table(d, src):
    r, cb, c5 = d["roofline"], d["cpu_baseline"], d["config5_one_gpu"]
    comp, pk, bd, ha = d["compaction"], d["pack_twin"], d["batched_dense"], d["host_api"]
    hs = d.get("host_api_single")
    rows = [
        ("8 × 1280×720 streams, fused dense kernel, one frame-set per launch (**headline `value`**)",
         us(r["avg_launch_ms"]), f"{d['value'] / 1e3:.0f} k Mpoints/s", pct(r["frac"]) + (f" (by the wall clock `value` uses: {pct(r['frac_wall'])})" if "frac_wall" in r else "")),
        *([("ONE 1280×720 stream per launch (BASELINE `configs[1]`), cold ring, same kernel"
            + (" / as a frame loop over two contexts" if "frame_loop_two_contexts" in d["single_stream"] else ""),
            us(d["single_stream"]["ms_per_frame"]) + (f" / {us(d['single_stream']['frame_loop_two_contexts']['ms_per_frame'])}" if "frame_loop_two_contexts" in d["single_stream"] else ""),
            f"{d['single_stream']['value'] / 1e3:.0f} k Mpoints/s" + (f" / {d['single_stream']['frame_loop_two_contexts']['value'] / 1e3:.0f} k" if "frame_loop_two_contexts" in d["single_stream"] else ""),
            pct(d["single_stream"]["roofline"]["frac"]) + " (latency-bound)"
            + (f" / {pct(d['single_stream']['frame_loop_two_contexts']['frac'])}" if "frame_loop_two_contexts" in d["single_stream"] else ""))] if "single_stream" in d else []),
        (f"same tiles, {bd['frame_sets_per_launch']} frame-sets per launch (`pcs_process_frames_device_batch`)",
         us(bd["ms_per_frame_set"]) + " / frame-set", f"{bd['value'] / 1e3:.0f} k Mpoints/s", pct(bd["frac"])),
        ("same, 1° depth→colour rotation", us(d["general_rotation"]["ms_per_step"]), f"{d['general_rotation']['value'] / 1e3:.0f} k Mpoints/s",
         pct(d["general_rotation"]["frac"])),
        ("the geometry a D400 rig records (colour 1920×1080, rotation, colour distortion)", us(d["color_1080p"]["ms_per_step"]),
         f"{d['color_1080p']['value'] / 1e3:.0f} k Mpoints/s", f"{pct(d['color_1080p']['frac'])} (by the colour lines it must touch: {pct(d['color_1080p']['frac_touched_bytes'])})"),
        ("a2 twin from resident `rs2::points` arrays (33 B/point): 8 cameras in one launch / one launch per camera" + (" / ONE cloud per call" if "single" in pk else ""),
         f"{us(pk['batched_ms_per_frame_set'])} / {us(pk['per_stream_launches_ms_per_frame_set'])}" + (f" / {us(pk['single']['ms_per_cloud'])}" if "single" in pk else ""), "",
         f"{pct(pk['batched_frac'])} / {pct(pk['per_stream_launches_frac'])}" + (f" / {pct(pk['single']['roofline']['frac'])}" if "single" in pk else "")),
        ("invalid-depth compaction, order-preserving (count + scan + emit) / per-tile counts handed in / 4 frame-sets per call ((5 + 10ρ) B/point)",
         f"{us(comp['ms_per_step'])} / {us(comp['caller_counts']['ms_per_step'])} / {us(comp['batched']['ms_per_frame_set'])}", "",
         f"{pct(comp['frac'])} / {pct(comp['caller_counts']['frac'])} / {pct(comp['batched']['frac'])}"),
        ("centre-side re-transform of 8 packed 720p payloads (`pcs_transform_payloads_device`, 20 B/record)",
         us(d["centre_transform"]["ms_per_frame_set"]), "", pct(d["centre_transform"]["frac"])),
        ("BASELINE config 5 on one GPU, 16 × 1080p, 50 mm: compaction → stitch → voxel grid (two calls) / ONE call from the rasters"
         + (" / that call as a frame loop over two contexts" if "frame_loop_two_contexts_ms_per_frame_set" in c5["one_call"] else ""),
         f"{c5['pipeline_ms_per_frame_set']:.3f} ms / **{c5['one_call']['ms_per_frame_set']:.3f} ms**"
         + (f" / **{c5['one_call']['frame_loop_two_contexts_ms_per_frame_set']:.3f} ms**" if "frame_loop_two_contexts_ms_per_frame_set" in c5["one_call"] else ""),
         f"{c5['one_call']['value'] / 1e3:.0f} k Mpixels/s in", "—"),
        ("8 cameras, host pointers in and out (PCIe both ways): staged / zero copy / software-pipelined",
         f"{ha['ms_per_step']:.2f} / {ha['pinned_ms_per_step']:.2f} / {ha['pipelined_ms_per_step']:.2f} ms", "", f"link-bound (CPU port: {cb['ms_per_frame_set']:.2f} ms)"),
        *([("ONE camera, host pointers (the reference's deployment): a2 twin pageable / page-locked · fused staged / zero copy / pipelined",
            f"{hs['a2_twin_ms_per_frame']['pageable']:.2f} / {hs['a2_twin_ms_per_frame']['page_locked']:.2f} · {hs['fused_ms_per_frame']['staged_pageable']:.2f} / "
            f"{hs['fused_ms_per_frame']['zero_copy_page_locked']:.2f} / {hs['fused_ms_per_frame']['pipelined_submit_collect']:.2f} ms per frame", "",
            (f"link-bound (CPU port, one frame: `-t1` {hs['cpu_port_ms_per_frame']['t1']:.2f}, `-t8` {hs['cpu_port_ms_per_frame']['t8']:.2f}, best "
             f"`-t{hs['cpu_port_ms_per_frame']['best_threads']}` {hs['cpu_port_ms_per_frame']['best']:.2f} ms; with its deprojection {hs['cpu_port_ms_per_frame']['with_deprojection_best']:.2f} ms)")
            if "cpu_port_ms_per_frame" in hs else "link-bound")] if hs else []),
        (f"CPU baseline, same box ({cb['cpu_model']}, {cb['host_physical_cores']} physical cores): SSE/FMA+OpenMP port of `-m -t{cb['cores']}`, "
         f"median of {cb['passes']} passes, team bound, reference's own bracket",
         f"{cb['ms_per_frame_set']:.2f} ms", f"{cb['value'] / 1e3:.2f} k Mpoints/s", f"GPU = {d['speedup_vs_cpu_baseline']:.1f} ×"),
    ]
    out = [BEGIN, f"(from `{src}`: one `python bench.py` line, inputs cold in HBM; regenerate with `python tools/readme_results.py --write`)", "",
           "| workload | time | rate | of 8 TB/s HBM |", "|---|---|---|---|"]
    out += ["| " + " | ".join(x) + " |" for x in rows]
    if "parity" in d:
        out += ["", "Parity: " + d["parity"] + "."]
    out.append(END)
    return "\n".join(out)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    src = args[0] if args else os.path.join("profiles", "r06_bench.json")
    text = table(load(os.path.join(ROOT, src) if not os.path.isabs(src) else src), src)
    if "--write" in sys.argv:
        p = os.path.join(ROOT, "README.md")
        s = open(p).read()
        a, b = s.index(BEGIN), s.index(END) + len(END)
        open(p, "w").write(s[:a] + text + s[b:])
    else:
        print(text)


if __name__ == "__main__":
    main()
