#!/bin/bash
# Runs on the GPU box: per-kernel launch statistics of one command.   tools/kstats.sh <outdir> <label> <command...>
# Prints "label kernel calls avg_us" for every kernel with >= 20 calls; the CSV stays in <outdir>/<label>/.
OUT=$1; LABEL=$2; shift 2
mkdir -p $OUT
export TMPDIR=/tmp
ABS=$(cd $OUT && pwd)
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ABS/$LABEL -- "$@" > $ABS/$LABEL.log 2>&1 )
python - "$ABS/$LABEL" "$LABEL" <<'PY'
import csv, glob, re, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); k = re.sub(r"^void ", "", k); k = re.sub(r"\(.*", "", k)[:90]
        if int(r["Calls"]) >= 20:
            print(f"{sys.argv[2]:14s} {k:92s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.2f} us")
PY
