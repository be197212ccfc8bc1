mkdir -p gpurun_out
python -m pytest "tests/test_ops_gpu.py" -q -m gpu -k "thin or conv2d_fwd or dgrad_and_wgrad or every_config or adjointness" 2>&1 | tail -8
python tools/bench_conv.py 64 thin 2>&1 | grep -v amdgpu.ids
python tools/bench_conv.py 64 "(B)" 2>&1 | grep -v amdgpu.ids
MMDGAN_N2W_BLOCKS=768 python tools/bench_conv.py 64 thin 2>&1 | grep "thin"
MMDGAN_N2W_BLOCKS=1024 python tools/bench_conv.py 64 thin 2>&1 | grep "thin"
MMDGAN_W2N_THREADS=256 python tools/bench_conv.py 64 thin 2>&1 | grep "thin"
python tools/ab_env.py --modes eager base MMDGAN_N2W_BLOCKS=768 MMDGAN_W2N_THREADS=256 2>&1 | tee gpurun_out/ab_thin.txt
