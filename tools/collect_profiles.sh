#!/bin/bash
# run on the GPU box (through gpurun): collects the rocprofv3 evidence for profiles/.
#   tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>/...
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r01}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. the default bench command (launch mode chosen during warm-up; 4 streams): kernel trace + stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/graph -o g -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_graph.json 2> $OUT/bench_graph.err
# 2. the same step issued eagerly on ONE compute stream + one SN stream (per-kernel durations without overlap inflation)
MMDGAN_SIDE_WGRAD=0 MMDGAN_SN_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/eager -o e -- python $R/bench.py --steps 20 --warmup 5 --no-graph --no-cpu-baseline > $OUT/bench_eager.json 2> $OUT/bench_eager.err
# 3. the dominant kernel alone: stats row == the launches roofline.dominant_kernel times
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/probe -o p -- python $R/bench.py --probe-only --probe-reps 50 > $OUT/probe.json 2> $OUT/probe.err
# 4. PMC passes (own runs, kernel-trace only): HBM read / write bytes, MFMA busy
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python $R/bench.py --probe-only --probe-reps 50 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python $R/bench.py --probe-only --probe-reps 50 > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o q -- python $R/bench.py --probe-only --probe-reps 50 > /dev/null 2> $OUT/pmc_sq.err
ls -R $OUT | head -50
BENCH_DGRAD_3B=1 python $R/tools/bench_conv.py 64 > $OUT/conv_layers.txt 2>&1
MMDGAN_WINO=0 BENCH_DGRAD_3B=1 python $R/tools/bench_conv.py 64 'D l' > $OUT/conv_layers_direct.txt 2>&1
python $R/tools/issue_time.py > $OUT/launch_modes.txt 2>&1
for c in stl celeba; do python $R/bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_$c.json 2>/dev/null; done
cd $R && python bench.py > $OUT/bench.json 2> $OUT/bench.err
