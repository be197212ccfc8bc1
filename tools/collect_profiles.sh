#!/bin/bash
# run on the GPU box (through gpurun): collects the rocprofv3 evidence for profiles/.
#   tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>/...     then, here: python tools/make_profiles.py <tag>
#   tools/collect_profiles.sh <tag> benches   only the bench lines: run it after make_profiles.py has written the
#                                             r03_dominant_kernel_*.json of the same build (bench.py takes `traffic` from them)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r05}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
benches() {
cd $R
MMDGAN_DP_FORCE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --no-cpu-baseline > $OUT/bench_dp_one_rank.json 2> /dev/null
for c in stl celeba lsun_resnet; do python bench.py --config $c --steps 20 --warmup 5 > $OUT/bench_$c.json 2>/dev/null; done
python bench.py > $OUT/bench.json 2> $OUT/bench.err
}
if [ "${2:-all}" = "benches" ]; then benches; exit 0; fi
if [ "${2:-all}" != "probes" ]; then
# 1. the default bench command (launch mode chosen during warm-up; 4 streams): kernel trace + stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/default -o g -- $B --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline > $OUT/bench_default_profiled.json 2> $OUT/bench_default.err
# 2. the same step issued eagerly on ONE compute stream + one SN stream (per-kernel durations without overlap inflation)
MMDGAN_SIDE_WGRAD=0 MMDGAN_SN_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/single -o e -- $B --steps 20 --warmup 5 --repeats 1 --launch-mode eager --no-cpu-baseline > $OUT/bench_single_profiled.json 2> $OUT/bench_single.err
# 3. one steady-state step as a timeline (plan replay, 4 streams)
rocprofv3 --kernel-trace --output-format csv -d $OUT/timeline -o t -- $B --steps 10 --warmup 5 --repeats 1 --launch-mode plan --no-cpu-baseline > /dev/null 2> $OUT/timeline.err
fi
# 4. the dominant kernel alone, per config: kernel trace (the probe launches are isolated by grid size in make_profiles.py)
#    and the PMC passes (own runs, kernel-trace only): HBM read / write bytes, MFMA busy
for c in cifar stl celeba lsun_resnet; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/probe_$c -o p -- $B --config $c --probe-only --probe-reps 50 > $OUT/probe_$c.json 2> $OUT/probe_$c.err
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$c -o f -- $B --config $c --probe-only --probe-reps 50 > /dev/null 2> $OUT/pmc_fetch_$c.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$c -o w -- $B --config $c --probe-only --probe-reps 50 > /dev/null 2> $OUT/pmc_write_$c.err
  $B --config $c --probe-only --probe-reps 50 > $OUT/probe_unprofiled_$c.json 2> /dev/null
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq_cifar -o q -- $B --probe-only --probe-reps 50 > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --kernel-trace --pmc TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_ta_cifar -o a -- $B --probe-only --probe-reps 50 > /dev/null 2> $OUT/pmc_ta.err
if [ "${2:-all}" = "probes" ]; then exit 0; fi
# 5. whole-step MFMA busy: the SQ counters over every kernel of 8 eagerly issued single-stream steps
MMDGAN_SIDE_WGRAD=0 MMDGAN_SN_STREAMS=1 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_step -o s -- $B --steps 5 --warmup 3 --repeats 1 --launch-mode eager --no-cpu-baseline > /dev/null 2> $OUT/pmc_step.err
# 6. the ResNet-SN config: kernel stats of its bench command
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/resnet -o r -- $B --config lsun_resnet --steps 10 --warmup 3 --repeats 1 --launch-mode eager --no-cpu-baseline > $OUT/bench_resnet_profiled.json 2> $OUT/bench_resnet.err
ls $OUT
cd $R
BENCH_DGRAD_3B=1 python tools/bench_conv.py 64 > $OUT/conv_layers.txt 2>&1
BENCH_OWN_TRANSFORM=1 BENCH_DGRAD_3B=1 python tools/bench_conv.py 64 > $OUT/conv_layers_own_transform.txt 2>&1
MMDGAN_WINO=0 BENCH_DGRAD_3B=1 python tools/bench_conv.py 64 'D l' > $OUT/conv_layers_direct.txt 2>&1
python tools/wino_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/winograd_kernels.txt
MMDGAN_WINO2=0 MMDGAN_WINO=0 python tools/wino_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/winograd_kernels_direct.txt
(python tools/step_clock.py; MMDGAN_SIDE_WGRAD=0 MMDGAN_SN_STREAMS=1 python tools/step_clock.py; python tools/step_clock.py celeba) 2>&1 | grep "average shader clock" > $OUT/step_clock.txt
python tools/bench_mmd.py 2>&1 | grep -v amdgpu.ids > $OUT/pairwise_kernel_sweep.txt
python tools/issue_time.py > $OUT/launch_modes.txt 2>&1
python tools/issue_time.py celeba >> $OUT/launch_modes.txt 2>&1
benches
