mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -q -m gpu -k "gemm" 2>&1 | tail -4
python -m pytest tests/test_step_gpu.py -q -m gpu -k "golden or warm_start or shipped" -x 2>&1 | tail -4
python tools/ab_env.py MMDGAN_GEMM_SHORTK=0 base 2>&1 | tee gpurun_out/ab_shortk.txt
