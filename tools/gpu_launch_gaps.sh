#!/bin/bash
# Developer probe (GPU box): gaps between consecutive solver launches of the bench loop (kernel trace: end of one launch to the start of the next).
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/gaps
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --workload ${1:-config2} --steps 20 --warmup 3 --no-cpu-baseline --secondary none > $OUT/trace.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "cmpc_solve_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("dur %.1f us" % ((e - s) / 1e3), "" if prev is None else "gap %.1f us" % ((s - prev) / 1e3), "scratch", r.get("Scratch_Size", r.get("Private_Segment_Size", "?")), "lds", r.get("LDS_Block_Size", "?"))
    prev = e
PY
