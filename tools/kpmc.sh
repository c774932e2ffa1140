#!/bin/bash
# GPU box: per-kernel means of one group of PMC counters for one command (its own rocprofv3 pass: --pmc + --kernel-trace only).
#   tools/kpmc.sh <outdir> <label> "<COUNTER ...>" <command...>
OUT=$1; LABEL=$2; CTRS=$3; shift 3
mkdir -p $OUT
export TMPDIR=/tmp
ABS=$(cd $OUT && pwd)
( cd /tmp && timeout 900 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $ABS/$LABEL -- "$@" > $ABS/$LABEL.log 2>&1 )
python - "$ABS/$LABEL" "$LABEL" <<'PY'
import csv, glob, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); k = re.sub(r"^void ", "", k); k = re.sub(r"\(.*", "", k)[:70]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    n = max(len(v) for v in cs.values())
    if n >= 10:
        print(f"{sys.argv[2]:10s} {k:72s} n={n:5d} " + "  ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(cs.items())))
PY
