#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of EVERY kernel family.
#   tools/profile_gpu.sh <tag> [steps]
# Summaries land in gpurun_out/prof_<tag>/ ; copy the ones to keep into profiles/ :
#   <tag>_kernel_stats.csv   all kernels of all runs below, one row per (run, kernel): calls, avg/min/max ns
#   <tag>_pmc_summary.json   per-kernel means of the PMC counters (one rocprofv3 pass per counter group, --pmc only)
#   traffic.json             HBM bytes per launch of the headline kernel (2 x FETCH_SIZE + WRITE_SIZE, KiB units)
# Runs: the headline line (dense, --no-extra-legs so the average is over cold launches of ONE kernel), then one
# run per other kernel family (bench.py --mode ...), the config-5 tail (tools/voxel_bench.py) and the config-5 workload
# (bench.py --workload config5: partials + sort + segmented mean from caller-held partials).
TAG=${1:-r06}
STEPS=${2:-200}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps $STEPS --warmup 20 --no-cpu-baseline --no-host-api --no-extra-legs"
declare -A RUNS
RUNS[dense]="$BENCH"
RUNS[drop_invalid]="$BENCH --mode drop_invalid"
RUNS[drop_invalid_single]="env PCS_COMPACT_PATH=single $BENCH --mode drop_invalid"
RUNS[cutoff]="$BENCH --mode cutoff"
RUNS[pack]="$BENCH --mode pack"
RUNS[pack_batch]="$BENCH --mode pack_batch"
RUNS[batch]="$BENCH --mode batch"
RUNS[batch_drop_invalid]="$BENCH --mode batch_drop_invalid"
RUNS[voxel]="python $PWD/tools/voxel_bench.py 16 1920 1080 50,200"
RUNS[config5]="python $PWD/bench.py --workload config5 --steps 60 --warmup 5 --no-cpu-baseline"
# the one-process node route (libpcs_node) on this box's one GPU: 8 virtual peers, the exchange = RCCL self send/recv pairs
RUNS[node_stitch]="python $PWD/bench.py --gpus 8 --node-devices 0,0,0,0,0,0,0,0 --steps $STEPS --warmup 20"
RUNS[node_config5]="python $PWD/bench.py --workload config5 --gpus 8 --node-devices 0,0,0,0,0,0,0,0 --steps 60 --warmup 5"
# BASELINE configs[4] in ONE call from the rasters (pcs_process_frames_voxel_device, 50 mm): front end + the warm bucket tail (2 launches
# per call); the bucket tail held to its cold chain (5); the LSD tail (13)
RUNS[voxel_one_call]="python $PWD/tools/voxel_probe.py 50 80"
RUNS[voxel_one_call_cold]="env PCS_VOXEL_REGIONS=0 python $PWD/tools/voxel_probe.py 50 80"
RUNS[voxel_one_call_lsd]="env PCS_VOXEL_TAIL=lsd python $PWD/tools/voxel_probe.py 50 80"
# BASELINE configs[1]: ONE 1280x720 stream per launch over a cold ring (fused kernel; the a2 twin's one-cloud call), both tile shapes
RUNS[single]="python $PWD/tools/single_probe.py 1 3000"
RUNS[single_512]="env PCS_SMALL_TILES=1 python $PWD/tools/single_probe.py 1 3000"
RUNS[twin_single]="python $PWD/tools/single_probe.py 1 3000 twin"
# BASELINE configs[4] as a frame loop over two contexts used in turn (the tail of k beside the pre-aggregation of k+1); the same with
# the colour row computed per pixel (PCS_ROW_CONST=0)
RUNS[voxel_two_ctx]="python $PWD/tools/voxel_overlap_probe.py 2 50 80"
RUNS[voxel_one_call_norowc]="env PCS_ROW_CONST=0 python $PWD/tools/voxel_probe.py 50 80"
# every leg of the default line (incl. centre_transform, config5_one_gpu, color_1080p): one row per kernel of the library
RUNS[all_legs]="python $PWD/bench.py --steps $STEPS --warmup 20 --no-cpu-baseline --no-host-api"
ORDER=${PROFILE_RUNS:-"dense all_legs single single_512 twin_single drop_invalid cutoff pack pack_batch batch batch_drop_invalid voxel voxel_one_call voxel_one_call_norowc voxel_one_call_cold voxel_one_call_lsd voxel_two_ctx config5 node_stitch node_config5"}
PMC_ORDER=${PROFILE_PMC_RUNS:-"dense all_legs single voxel_one_call voxel_one_call_norowc voxel_one_call_cold voxel_one_call_lsd config5"}
cd /tmp
for R in $ORDER; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$R -- ${RUNS[$R]} > $OUT/stats_$R.log 2>&1
  echo "stats $R rc=$?"
done
# PMC passes (their own runs, --pmc + --kernel-trace only): HBM traffic for every family, the instruction mix for the headline
for R in $PMC_ORDER; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${R}_$C -- ${RUNS[$R]} > $OUT/pmc_${R}_$C.log 2>&1
    echo "pmc $R $C rc=$?"
  done
done
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  for R in dense voxel_one_call voxel_one_call_norowc; do      # the headline kernel (HBM-bound) and the voxel pipeline (VALU-bound front end), with and without the colour-row table
    timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${R}_$N -- ${RUNS[$R]} > $OUT/pmc_${R}_$N.log 2>&1
    echo "pmc $R $N rc=$?"
  done
done
cd - > /dev/null
python - <<PY
import csv, glob, os, collections, json, re
out = "$OUT"
def short(k):
    k = re.sub(r"\(anonymous namespace\)::", "", k)
    k = re.sub(r"^void ", "", k)
    k = re.sub(r"\(.*", "", k)
    return k[:140]
rows = []
for d in sorted(glob.glob(out + "/stats_*")):
    if not os.path.isdir(d):
        continue
    run = os.path.basename(d)[6:]
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = short(r["Name"])
            if not ("pcs" in name or "rocprim" in name or "ccl" in name.lower()):
                continue
            rows.append({"run": run, "kernel": name, "calls": r["Calls"], "avg_ns": r["AverageNs"], "min_ns": r["MinNs"],
                         "max_ns": r["MaxNs"], "total_ns": r["TotalDurationNs"], "pct_of_run": r["Percentage"]})
with open(out + "/kernel_stats.csv", "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=["run", "kernel", "calls", "avg_ns", "min_ns", "max_ns", "total_ns", "pct_of_run"])
    w.writeheader(); w.writerows(rows)
summ = {}
for d in sorted(glob.glob(out + "/pmc_*")):
    if not os.path.isdir(d):
        continue
    run = os.path.basename(d)[4:]
    run = re.sub(r"_(FETCH_SIZE|WRITE_SIZE|SQ_.*)$", "", run)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            if not ("pcs" in k or "rocprim" in k):
                continue
            for c, v in cs.items():
                summ.setdefault(run, {}).setdefault(k, {})[c] = {"n": len(v), "mean": sum(v) / len(v)}
# HBM traffic per launch: FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports half the bytes of wide
# coalesced reads (MI355X_MICROARCH.md, HBM) -> double it.
for run, ks in summ.items():
    for k, cs in ks.items():
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            cs["traffic_bytes_per_launch"] = (2.0 * cs["FETCH_SIZE"]["mean"] + cs["WRITE_SIZE"]["mean"]) * 1024.0
json.dump(summ, open(out + "/pmc_summary.json", "w"), indent=1, sort_keys=True)
for k, cs in summ.get("dense", {}).items():
    # the headline launch is the identity-R instantiation
    if "fused_dense_kernel" in k and "CertMath<true" in k and "traffic_bytes_per_launch" in cs:
        json.dump({"tag": "$TAG", "workload": "8x1280x720", "kernel": k, "fetch_size_kib": cs["FETCH_SIZE"]["mean"],
                   "write_size_kib": cs["WRITE_SIZE"]["mean"], "fetch_correction": 2.0,
                   "traffic_bytes_per_launch": cs["traffic_bytes_per_launch"],
                   "algorithmic_bytes_per_launch": 8 * 1280 * 720 * 15}, open(out + "/traffic.json", "w"), indent=1)
import shutil
for d in glob.glob(out + "/stats_*") + glob.glob(out + "/pmc_*"):     # raw rocprofv3 output: hundreds of MB; the summaries above are what is kept
    if os.path.isdir(d):
        shutil.rmtree(d, ignore_errors=True)
for r in rows:
    print(f"{r['run']:22s} {r['kernel'][:90]:90s} calls {r['calls']:>6s} avg {float(r['avg_ns'])/1e3:9.2f} us")
PY
