#!/bin/bash
# One rocprofv3 counter pass per quoted argument over a short timing run; development aid.
# usage: [RT_HOTPATH_LIB=...] [RT_PMC_SCENE=s1m] [RT_PMC_ORDER=near_first_exact] tools/pmc_pass.sh <pipeline> <limit> "<counters pass 1>" "<counters pass 2>" ...
cd /tmp && export TMPDIR=/tmp
pipe=$1; limit=$2; shift 2
i=0
for pass in "$@"; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $pass --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${pipe}_$i -- python3 $GRAFT_REPO_ROOT/tools/time_render.py --scenes ${RT_PMC_SCENE:-bunny} --limits $limit --pipelines $pipe --subframes 1 --reps 3 --order ${RT_PMC_ORDER:-reference} > /dev/null 2>&1 || exit 1
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $GRAFT_REPO_ROOT/gpurun_out/pmc_${pipe}_$i
done
