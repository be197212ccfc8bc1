#!/bin/bash
# A/B of the XCD-aware workgroup map of wino2_kernel on the dominant launch (D l2 3B-row input-gradient):
# kernel duration (rocprofv3 kernel trace), HBM read bytes (FETCH_SIZE), and the whole step.
#   tools/xcd_probe.sh <tag>   -> gpurun_out/<tag>/
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${1:-xcd}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for M in 0 1; do
  export MMDGAN_XCD_REMAP=$M
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace$M -o t -- python $R/bench.py --probe-only --probe-reps 50 > $OUT/probe$M.json 2> $OUT/probe$M.err
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch$M -o f -- python $R/bench.py --probe-only --probe-reps 50 > /dev/null 2> $OUT/fetch$M.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write$M -o w -- python $R/bench.py --probe-only --probe-reps 50 > /dev/null 2> $OUT/write$M.err
  python $R/bench.py --no-cpu-baseline --launch-mode plan > $OUT/bench$M.json 2> $OUT/bench$M.err
  BENCH_DGRAD_3B=1 python $R/tools/bench_conv.py 64 'l' > $OUT/conv$M.txt 2>&1
done
python - <<PY
import csv, glob, json, collections
for m in (0, 1):
    out = '$OUT'
    tr = glob.glob(out + '/trace%d/**/t_kernel_trace.csv' % m, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(tr)) if r['Kernel_Name'].startswith('mmdgan::wino2_kernel') or 'wino2_kernel' in r['Kernel_Name']]
    by = collections.defaultdict(list)
    for r in rows:
        by[(r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size'), )].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    grid, d = max(by.items(), key=lambda kv: len(kv[1]))
    res = {'remap': m, 'probe_grid': grid, 'launches': len(d), 'avg_us_profiled': sum(d) / len(d)}
    for tag, name in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
        f = glob.glob(out + '/%s%d/**/*_counter_collection.csv' % (tag, m), recursive=True)[0]
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'wino2_kernel' in r['Kernel_Name'] and r['Counter_Name'] == name:
                acc[r['Grid_Size']].append(float(r['Counter_Value']))
        g, v = max(acc.items(), key=lambda kv: len(kv[1]))
        res[name + '_KiB'] = sum(v) / len(v)
    res['hbm_read_MB (2x FETCH_SIZE)'] = 2 * res['FETCH_SIZE_KiB'] * 1024 / 1e6
    res['unprofiled_us'] = json.load(open(out + '/probe%d.json' % m))['dominant_kernel']['ms'] * 1e3
    b = json.loads(open(out + '/bench%d.json' % m).read().strip().splitlines()[-1])
    res['step_ms'] = b['ms_per_step']
    print(json.dumps(res))
PY
