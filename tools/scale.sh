#!/bin/bash
# the scaling curve in one command, on a node with 8 MI355X: bench.py for N in {1,2,4,8} x {cifar, celeba}, one rank per
# GPU over RCCL / xGMI (weak scaling: the per-GPU batch is fixed), then one table.
#   tools/scale.sh [steps] [warmup]        env: CONFIGS="cifar celeba lsun_resnet"  NS="1 2 4 8"  BACKENDS="capi torch"
# Every N > 1 runs once per exchange backend (capi = the library's own RCCL communicator, collectives as launch-plan nodes;
# torch = ProcessGroupNCCL): the table is the A/B.  bench.py's config.exchange carries G's last bucket: its duration and how
# much of it the main stream waited for - the exposed part of the exchange, the rest travels under the backward kernels.
set -u
cd "$(dirname "$0")/.."
STEPS=${1:-200}; WARM=${2:-20}
CONFIGS=${CONFIGS:-"cifar celeba lsun_resnet"}; NS=${NS:-"1 2 4 8"}; BACKENDS=${BACKENDS:-"capi torch"}
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=${OUT:-gpurun_out/scale}; mkdir -p $OUT
PORT=29600
for c in $CONFIGS; do
  for n in $NS; do
    PORT=$((PORT + 1))
    if [ "$n" = 1 ]; then
      python bench.py --gpus 1 --config $c --steps $STEPS --warmup $WARM --no-cpu-baseline > $OUT/${c}_${n}_single.json 2> $OUT/${c}_${n}_single.err
    else
      for be in $BACKENDS; do
        PORT=$((PORT + 1))
        MMDGAN_DP_BACKEND=$be python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $PORT \
          bench.py --gpus $n --config $c --steps $STEPS --warmup $WARM --no-cpu-baseline > $OUT/${c}_${n}_$be.json 2> $OUT/${c}_${n}_$be.err
      done
    fi
  done
done
python - "$OUT" "$BACKENDS" $CONFIGS <<'PY'
import json, sys
out, backends, configs = sys.argv[1], sys.argv[2].split(), sys.argv[3:]
print('%-12s %4s %-7s %12s %10s %10s %6s  %-6s %s' % ('config', 'GPUs', 'backend', 'images/s', 'ms/step', 'x 1 GPU', 'eff', 'mode', 'last bucket: MB, ms, exposed ms'))
for c in configs:
    base = None
    for n in (1, 2, 4, 8):
        for be in (['single'] if n == 1 else backends):
            f = '%s/%s_%d_%s' % (out, c, n, be)
            try:
                r = json.loads([l for l in open(f + '.json') if l.startswith('{')][-1])
            except Exception:
                print('%-12s %4d %-7s  (no result: see %s.err)' % (c, n, be, f))
                continue
            base = base or r['value'] / r['n_gpus']
            sp = r['value'] / base
            ex = r['config'].get('exchange') or {}
            exs = '%.1f, %.3f, %.3f' % (ex['last_bucket_bytes'] / 1e6, ex['last_bucket_ms'], ex['exposed_ms']) if ex else '-'
            print('%-12s %4d %-7s %12.0f %10.3f %10.2f %5.0f%%  %-6s %s' % (c, n, be, r['value'], r['ms_per_step'], sp, 100 * sp / n,
                  r['config'].get('launch_mode'), exs))
PY
