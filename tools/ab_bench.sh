#!/bin/bash
# Developer tool (GPU box): A/B of two builds of the library inside ONE gpurun call (box-to-box spread is ~2 %, a call's own
# spread ~0.1 %).  Build the variant as <package>/libcmpc_hip_exp.so (git-ignored, travels with gpurun), then
#   gpurun -- 'bash tools/ab_bench.sh config2 3'
set -e
P=paper_romualdi_2022_icra_centroidal-mpc-walking_amd
WL=${1:-config2}; REPS=${2:-3}
cp $P/libcmpc_hip.so /tmp/base.so
trap 'cp /tmp/base.so $P/libcmpc_hip.so' EXIT
for rep in $(seq $REPS); do
  for v in base exp; do
    if [ $v = base ]; then cp /tmp/base.so $P/libcmpc_hip.so; else cp $P/libcmpc_hip_exp.so $P/libcmpc_hip.so; fi
    python bench.py --workload $WL --no-cpu-baseline --secondary none --steps 30 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v','$WL',d['value'],d['ms_per_step'])"
  done
done
