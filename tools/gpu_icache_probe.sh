#!/bin/bash
# Developer probe (GPU box): instruction-cache behaviour of the solve kernel (one rocprofv3 counter pass, counters alone).  Argument: workload.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
WL=${1:-config2}
OUT=$REPO/gpurun_out/icache_$WL
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PARGS="$REPO/bench.py --workload $WL --steps 5 --warmup 1 --no-cpu-baseline --secondary none"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc -- python3 $PARGS > $OUT/pmc.log 2>&1 || { tail -5 $OUT/pmc.log; exit 1; }
python3 - <<PY
import csv, glob
fn = sorted(glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True))[-1]
acc, n = {}, {}
for r in csv.DictReader(open(fn)):
    if "cmpc_solve_kernel" not in r["Kernel_Name"]: continue
    acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"]); n[r["Counter_Name"]] = n.get(r["Counter_Name"], 0) + 1
nl = max(n.values()) // 1
for k in sorted(acc): print("%-32s %.4g per launch" % (k, acc[k] / n[k]))
if "SQC_ICACHE_REQ" in acc: print("hit rate %.4f" % (acc["SQC_ICACHE_HITS"] / acc["SQC_ICACHE_REQ"]))
if "SQ_IFETCH_LEVEL" in acc and acc.get("SQ_IFETCH"): print("mean instruction-fetch latency %.1f (quad?)cycles" % (acc["SQ_IFETCH_LEVEL"] / acc["SQ_IFETCH"]))
PY
