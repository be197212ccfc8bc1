#!/bin/bash
# on the GPU box: the warm-start free-run tests N times over (VERDICT r04 "next" #4: knife-edge retries must be 0 in every run)
#   tools/repeat_warm.sh [N=20]  ->  gpurun_out/warm_repeats.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-20}
mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/warm_repeats.txt
: > $OUT
cd $R
for i in $(seq 1 $N); do
  python -m pytest tests/test_step_gpu.py -q -m gpu -k "test_free_run_from_warm_start_matches_reference" -p no:cacheprovider 2>&1 \
    | grep -E "passed|failed|knife-edge|fp32-floor" | sed "s/^/run $i: /" >> $OUT
done
echo "runs: $N" >> $OUT
echo "runs with 0 knife-edge retries: $(grep -c 'knife-edge retries.*: 0$' $OUT)" >> $OUT
echo "failed runs: $(grep -c 'failed' $OUT)" >> $OUT
tail -4 $OUT
