#!/bin/bash
# Runs on the GPU box: bench.py --mode drop_invalid / cutoff under both compaction implementations (8 x 720p and the
# 16 x 1080p half of config 5); prints one line each.   tools/compaction_sweep.sh [outdir]
OUT=${1:-gpurun_out/sweep}
mkdir -p $OUT
B="python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-host-api"
for geo in "8 1280 720 300" "16 1920 1080 100"; do
  set -- $geo; S=$1; W=$2; H=$3; K=$4
  for mode in drop_invalid cutoff; do
    for path in ${PATHS:-three single}; do
      f=$OUT/${S}x${H}_${mode}_$path.json
      PCS_COMPACT_PATH=$path $B --mode $mode --streams $S --width $W --height $H --steps $K > $f 2>>$OUT/err.log
      python - "$f" "${S}x${W}x${H} $mode $path" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    print(f"{sys.argv[2]:40s} {r['avg_launch_ms']*1e3:7.2f} us  frac {r['frac']:.4f}  median {r.get('per_launch_ms',{}).get('median')}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
    done
  done
done
