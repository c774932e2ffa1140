#!/bin/bash
# Runs on the GPU box: bench.py --mode drop_invalid / cutoff under every compaction implementation; prints one line each.
OUT=${1:-gpurun_out/sweep}
mkdir -p $OUT
B="python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-host-api"
run() {  # name, env...
  name=$1; shift
  for mode in drop_invalid cutoff; do
    env "$@" $B --mode $mode > $OUT/${mode}_$name.json 2>>$OUT/err.log
    python - "$OUT/${mode}_$name.json" "$mode $name" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d["roofline"]
    print(f"{sys.argv[2]:40s} {r['avg_launch_ms']*1e3:7.2f} us  frac {r['frac']:.4f}  median {r.get('per_launch_ms',{}).get('median')}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  done
}
run three PCS_COMPACT_PATH=three
run single PCS_COMPACT_PATH=single
run stream_w6 PCS_COMPACT_PATH=stream
run stream_w7 PCS_COMPACT_PATH=stream PCS_COMPACT_WPS=7
run stream_w5 PCS_COMPACT_PATH=stream PCS_COMPACT_WPS=5
run stream_w6_g1200 PCS_COMPACT_PATH=stream PCS_COMPACT_GRID=1200
run stream_w6_g900 PCS_COMPACT_PATH=stream PCS_COMPACT_GRID=900
run stream_w7_g1200 PCS_COMPACT_PATH=stream PCS_COMPACT_WPS=7 PCS_COMPACT_GRID=1200
run chunk4 PCS_COMPACT_PATH=chunk
run chunk2 PCS_COMPACT_PATH=chunk PCS_COMPACT_CHUNK_VARIANT=1
