mkdir -p gpurun_out
python tools/record_production_kernels.py > gpurun_out/record.log 2>&1; cp gpurun_out/production_kernels.json tests/golden/
(time python -m pytest tests -m gpu -q --durations=8) > gpurun_out/pytest_gpu.log 2>&1; tail -25 gpurun_out/pytest_gpu.log
python bench.py > gpurun_out/bench_b.json 2> gpurun_out/bench_b.err; python -c "
import json; d=json.load(open('gpurun_out/bench_b.json')); print(d['ms_per_step'], d['value'], d['config']['launch_mode'], d['config']['kernel_set'], d['roofline']['frac'], d['cpu_baseline']['thread_sweep'])"
