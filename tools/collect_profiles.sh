#!/bin/bash
# GPU box: rocprofv3 passes over the bench command (argument: workload, default config2 = B 256, N 20).  Kernel trace + stats in one run,
# every hardware-counter group in a run of its own (FETCH_SIZE and WRITE_SIZE do not fit one pass).  Raw output
# under gpurun_out/prof/, summarised into profiles/ by tools/summarise_profiles.py.
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_${1:-config2}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
WL=${1:-config2}
ARGS="$REPO/bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline --secondary none"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
echo "trace done"
PARGS="$REPO/bench.py --workload $WL --steps 5 --warmup 1 --no-cpu-baseline --secondary none"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc$i -- python3 $PARGS > $OUT/pmc$i.log 2>&1
  echo "pmc pass $i ($grp) done"
done
