#!/bin/bash
# the scaling curve in one command, on a node with 8 MI355X: bench.py for N in {1,2,4,8} x {cifar, celeba}, one rank per
# GPU over RCCL / xGMI (weak scaling: the per-GPU batch is fixed), then one table.
#   tools/scale.sh [steps] [warmup]        env: CONFIGS="cifar celeba"  NS="1 2 4 8"  MMDGAN_DP_BACKEND=capi|torch
set -u
cd "$(dirname "$0")/.."
STEPS=${1:-200}; WARM=${2:-20}
CONFIGS=${CONFIGS:-"cifar celeba"}; NS=${NS:-"1 2 4 8"}
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=${OUT:-gpurun_out/scale}; mkdir -p $OUT
PORT=29600
for c in $CONFIGS; do
  for n in $NS; do
    PORT=$((PORT + 1))
    if [ "$n" = 1 ]; then
      python bench.py --gpus 1 --config $c --steps $STEPS --warmup $WARM --no-cpu-baseline > $OUT/${c}_$n.json 2> $OUT/${c}_$n.err
    else
      python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $PORT \
        bench.py --gpus $n --config $c --steps $STEPS --warmup $WARM --no-cpu-baseline > $OUT/${c}_$n.json 2> $OUT/${c}_$n.err
    fi
  done
done
python - "$OUT" $CONFIGS <<'PY'
import json, sys
out, configs = sys.argv[1], sys.argv[2:]
print('%-8s %4s %12s %10s %10s %8s  %s' % ('config', 'GPUs', 'images/s', 'ms/step', 'x 1 GPU', 'eff', 'launch mode / exchange'))
for c in configs:
    base = None
    for n in (1, 2, 4, 8):
        try:
            line = [l for l in open('%s/%s_%d.json' % (out, c, n)) if l.startswith('{')][-1]
            r = json.loads(line)
        except Exception:
            print('%-8s %4d   (no result: see %s/%s_%d.err)' % (c, n, out, c, n))
            continue
        base = base or r['value'] / r['n_gpus']
        sp = r['value'] / base
        print('%-8s %4d %12.0f %10.3f %10.2f %7.0f%%  %s / %s' % (c, n, r['value'], r['ms_per_step'], sp, 100 * sp / n,
              r['config'].get('launch_mode'), r['config'].get('dp_backend')))
PY
