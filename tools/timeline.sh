#!/bin/bash
# on the GPU box: one steady-state step as a timeline -> gpurun_out/<name>_timeline.txt
#   tools/timeline.sh <name> [bench.py args...]      (environment switches are inherited)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
NAME=${1:-step}; shift
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_$NAME
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$NAME -o t -- python $R/bench.py --steps 10 --warmup 5 --repeats 1 --launch-mode plan --no-cpu-baseline "$@" > /dev/null 2> $R/gpurun_out/${NAME}_timeline.err
python $R/tools/step_timeline.py /tmp/tl_$NAME > $R/gpurun_out/${NAME}_timeline.txt 2>> $R/gpurun_out/${NAME}_timeline.err
tail -3 $R/gpurun_out/${NAME}_timeline.txt
