#!/bin/bash
# per-tensor gradient errors of the ResNet-SN step at several batches / launch modes / kernel selections
cd "$(dirname "$0")/.."
export SHIPPED_STEP_REPORT=1
for spec in "32 plan" "32 eager" "8 plan" "4 eager"; do
  set -- $spec
  echo "== default env  B=$1 mode=$2"
  python tests/shipped_step.py lsun_resnet rep $1 $2 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
pt=d.get('per_tensor',{})
bad=sorted(pt.items(), key=lambda kv:-kv[1][0])[:12]
for k,v in bad: print('  %-50s L2 %.3e max %.3e'%(k,v[0],v[1]))
print('  images',d['err_images'],'scores',d['err_scores'],'losses',d['loss_gen'],d['loss_dis'])
"
done
echo "== test env (MMDGAN_WINO_MIN_TILES=32 MMDGAN_WINO2=2) B=32 plan"
MMDGAN_WINO_MIN_TILES=32 MMDGAN_WINO2=2 python tests/shipped_step.py lsun_resnet rep 32 plan 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
pt=d.get('per_tensor',{})
bad=sorted(pt.items(), key=lambda kv:-kv[1][0])[:12]
for k,v in bad: print('  %-50s L2 %.3e max %.3e'%(k,v[0],v[1]))
"
