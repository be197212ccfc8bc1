mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -q -m gpu -k "addend or every_config" 2>&1 | tail -6
python -m pytest tests/test_res_gpu.py tests/test_api_gpu.py tests/test_net_golden_gpu.py "tests/test_production_gpu.py" "tests/test_step_gpu.py::test_free_run_from_warm_start_matches_reference" "tests/test_step_gpu.py::test_library_owned_rccl_exchange_is_part_of_the_plan" -q -m gpu 2>&1 | tail -8
for v in 1 0; do MMDGAN_TAPE_FUSE_ADD=$v python bench.py --config lsun_resnet --no-cpu-baseline --steps 50 --repeats 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('FUSE_ADD=$v', d['ms_per_step'], d['ms_per_step_regions'], d['config']['launch_mode'])"; done
for v in 1 0; do MMDGAN_TAPE_FUSE_ADD=$v python bench.py --config lsun_resnet --no-cpu-baseline --steps 50 --repeats 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('FUSE_ADD=$v', d['ms_per_step'], d['ms_per_step_regions'], d['config']['launch_mode'])"; done
