mkdir -p gpurun_out
python tools/debug_conv_geoms.py folded 2>&1 | grep -v amdgpu.ids
SHIPPED_STEP_REPORT=1 python tests/shipped_step.py lsun_resnet rep 32 plan 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); pt=d['per_tensor']
for k,v in sorted(pt.items(), key=lambda kv:-kv[1][0])[:6]: print('  %-50s L2 %.3e max %.3e'%(k,v[0],v[1]))"
python -m pytest tests/test_production_gpu.py tests/test_api_gpu.py "tests/test_step_gpu.py::test_step_matches_oracle_mfma_path" "tests/test_step_gpu.py::test_library_owned_rccl_exchange_is_part_of_the_plan" "tests/test_step_gpu.py::test_data_parallel_exchange_runs_over_rccl" "tests/test_step_gpu.py::test_data_parallel_step_equals_the_mean_gradient_step" "tests/test_ops_gpu.py::test_conv_full_size_adjointness_and_linearity" -q -m gpu 2>&1 | tail -15
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 5 --repeats 1 --launch-mode plan --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/tl.err
cd $GRAFT_REPO_ROOT && python tools/step_timeline.py gpurun_out/tl > gpurun_out/r04_step_timeline_a.txt; rm -rf gpurun_out/tl
python tools/ab_env.py --modes eager MMDGAN_WGRAD_CUS=448 base MMDGAN_WGRAD_CUS=128 MMDGAN_WGRAD_CUS=256 2>&1 | tee gpurun_out/ab_wgrad_cus.txt
