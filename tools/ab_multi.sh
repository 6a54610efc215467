#!/bin/bash
# Developer tool (GPU box): several builds of the library (files in the package directory, selected through CMPC_LIB) benchmarked in turn
# inside ONE gpurun call, REPS rounds.   gpurun -- 'bash tools/ab_multi.sh config2 3 libcmpc_hip.so libcmpc_hip_exp.so ...'
WL=${1:-config2}; REPS=${2:-3}; shift 2
for rep in $(seq $REPS); do
  for lib in "$@"; do
    CMPC_LIB=$lib python bench.py --workload $WL --no-cpu-baseline --secondary none --steps 30 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib','$WL',d['value'],d['ms_per_step'])"
  done
done
