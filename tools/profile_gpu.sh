#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of bench.py.
# Summaries land in gpurun_out/prof_<tag>/ ; copy the ones to keep into profiles/.
TAG=${1:-r01}
STEPS=${2:-200}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $PWD/bench.py --steps $STEPS --warmup 20 --no-cpu-baseline --no-cache-leg --no-host-api"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
echo "stats rc=$?"
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -- $CMD > $OUT/pmc_$N.log 2>&1
  echo "pmc $N rc=$?"
done
cd - > /dev/null
python - <<PY
import csv, glob, os, collections, json
out = "$OUT"
summ = {}
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    summ["kernel_stats"] = rows[:12]
for d in glob.glob(out + "/pmc_*"):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:200]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            for c, v in cs.items():
                summ.setdefault("pmc", {}).setdefault(k, {})[c] = {"n": len(v), "mean": sum(v) / len(v)}
json.dump(summ, open(out + "/summary.json", "w"), indent=1)
# HBM traffic per launch of the fused kernel: FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports
# half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM) -> double it.
for k, cs in summ.get("pmc", {}).items():
    # the headline launch is the identity-R instantiation; the general-rotation leg's kernel is reported separately
    if "fused_dense" in k and "CertMath<true" in k and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        f, w = cs["FETCH_SIZE"]["mean"], cs["WRITE_SIZE"]["mean"]
        json.dump({"tag": "$TAG", "workload": "8x1280x720", "kernel": k, "fetch_size_kib": f, "write_size_kib": w,
                   "fetch_correction": 2.0, "traffic_bytes_per_launch": (2.0 * f + w) * 1024.0,
                   "algorithmic_bytes_per_launch": 8 * 1280 * 720 * 15}, open(out + "/traffic.json", "w"), indent=1)
print(json.dumps(summ, indent=1)[:6000])
PY
