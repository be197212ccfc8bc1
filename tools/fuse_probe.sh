#!/bin/bash
# A/B of the fused two-tile-block variant of wino2_kernel (MMDGAN_WINO2_FUSE) on the dominant launch: duration and the
# L1 -> L2 request counters.   gpurun -- tools/fuse_probe.sh   -> gpurun_out/fuse_probe.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out; OUT=$R/gpurun_out/fuse_probe.txt; : > $OUT
cd /tmp && export TMPDIR=/tmp
for f in 0 1; do
  rm -rf /tmp/fp$f
  MMDGAN_WINO2_FUSE=$f rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TA_BUSY_avr GRBM_GUI_ACTIVE --output-format csv -d /tmp/fp$f -o a -- python $R/bench.py --probe-only --probe-reps 50 > /dev/null 2>&1
  MMDGAN_WINO2_FUSE=$f python $R/bench.py --probe-only --probe-reps 50 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['dominant_kernel']; print('MMDGAN_WINO2_FUSE=$f  HIP events %.2f us  (%s)' % (d['ms']*1e3, d['kernel']))" >> $OUT
  python - $f >> $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob('/tmp/fp%s/**/a_counter_collection.csv' % sys.argv[1], recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if 'wino2_kernel' in r['Kernel_Name']:
        acc[r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
g, c = max(acc.items(), key=lambda kv: len(next(iter(kv[1].values()))))
print('   grid %s: ' % g + ', '.join('%s %.4g' % (k, sum(v[-50:]) / len(v[-50:])) for k, v in sorted(c.items())))
PY
done
cat $OUT
