#!/bin/bash
# Stand-alone duration of every stage launch of the spectral-norm power iterations (csrc/sn_chain.hip: sn_phase_kernel,
# one launch per stage for all kernels of a net) in the bench step, from a kernel trace of the single-stream schedule.
#   on the GPU box:  tools/sn_stage_times.sh [config]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sn_stages
MMDGAN_SIDE_WGRAD=0 MMDGAN_SN_STREAMS=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/sn_stages -o e -- \
    python $R/bench.py --config ${1:-cifar} --steps 20 --warmup 5 --repeats 1 --launch-mode eager --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import collections, csv, glob
f = glob.glob('/tmp/sn_stages/**/*kernel_trace.csv', recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if 'sn_phase' in r['Kernel_Name']:
        agg.setdefault(int(r['Grid_Size_X']) // 256, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
steps = min(len(v) for v in agg.values())
chain = 0.0
for g, v in agg.items():                                  # (stages with equal grids - the two norm pairs - share a row)
    v = sorted(v)
    print('%6d workgroups  %d launch(es) per step  median %6.1f us' % (g, len(v) // steps, v[len(v) // 2]))
    chain += v[len(v) // 2] * (len(v) // steps)
print('chain: %.0f us per step' % chain)
PY
