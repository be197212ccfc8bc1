#!/usr/bin/env python3
"""One steady-state training step as the GPU ran it: every kernel launch of the LAST complete step in a rocprofv3
kernel trace, in start order, with its hardware queue, grid, duration and the gap to the previous launch on the
same queue - the view that shows what is on the critical path and what hides under what.
    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python bench.py --steps 10 --warmup 5 --launch-mode plan --no-cpu-baseline
    python tools/step_timeline.py DIR > profiles/rNN_step_timeline.txt"""
import collections
import csv
import glob
import sys

tr = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(tr)))
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    r['name'] = r['Kernel_Name'].replace('mmdgan::', '').replace('void ', '').split('(')[0]
rows.sort(key=lambda r: r['s'])
loss = [i for i, r in enumerate(rows) if r['name'].startswith('mmd_kernel') or r['name'].startswith('score_loss')]
assert len(loss) >= 3, 'need a few steps in the trace'
# a step ends with G's Adam: the last adam_kernel before the next step's first kernel; steps are serialised on it
adams = [i for i, r in enumerate(rows) if r['name'].startswith(('adam_kernel', 'adam_segments_kernel'))]
ends = []
for a, b in zip(loss[:-1], loss[1:]):
    cand = [i for i in adams if a < i < b]
    ends.append(max(cand, key=lambda i: rows[i]['e']))
# the last complete ORDINARY step: after the Adam of step n-1 up to the Adam of step n.  (bench.py's untimed extras at the end
# of a run - the dominant-kernel probe's launches, the loss check - fall between two steps too: a candidate must have the
# launch count most steps have)
def between(lo, hi):
    t_lo = rows[lo]['e']
    return [r for r in rows if r['s'] >= t_lo and r['e'] <= rows[hi]['e'] and r is not rows[lo]]
cands = [(lo, hi, len(between(lo, hi))) for lo, hi in zip(ends[:-1], ends[1:])]
usual = collections.Counter(c[2] for c in cands).most_common(1)[0][0]
lo, hi = [(a, b) for a, b, n in cands if n == usual][-1]
step = between(lo, hi)
t0 = min(r['s'] for r in step)
span = (rows[hi]['e'] - t0) / 1e3
busy = sum(r['e'] - r['s'] for r in step) / 1e3
print('# last complete step of %s: %d launches, wall %.1f us (first launch -> end of G Adam), sum of kernel durations %.1f us'
      % (tr.split('/')[-1], len(step), span, busy))
queues = sorted({r['Queue_Id'] for r in step})
print('# queues: ' + ', '.join('q%s %d launches %.0f us' % (q, sum(1 for r in step if r['Queue_Id'] == q),
                                                            sum(r['e'] - r['s'] for r in step if r['Queue_Id'] == q) / 1e3) for q in queues))
print('%9s %8s %7s %4s %-18s %s' % ('start us', 'dur us', 'gap us', 'q', 'grid(wg)', 'kernel'))
last_end = {}
for r in step:
    gap = (r['s'] - last_end[r['Queue_Id']]) / 1e3 if r['Queue_Id'] in last_end else 0.0
    last_end[r['Queue_Id']] = r['e']
    wgs = (int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
    print('%9.1f %8.1f %7.1f %4s %-18s %s' % ((r['s'] - t0) / 1e3, (r['e'] - r['s']) / 1e3, gap, r['Queue_Id'],
                                               '%d(%s)' % (wgs, r['Workgroup_Size_X']), r['name'][:70]))
# time during which NO kernel of the step is running
ev = sorted([(r['s'], 1) for r in step] + [(r['e'], -1) for r in step])
idle, depth, prev = 0, 0, t0
for t, d in ev:
    if depth == 0:
        idle += t - prev
    depth += d
    prev = t
print('# GPU idle inside the step (no kernel running): %.1f us of %.1f' % (idle / 1e3, span))
by = collections.defaultdict(float)
for r in step:
    by[r['name'].split('<')[0]] += (r['e'] - r['s']) / 1e3
print('# by kernel: ' + ', '.join('%s %.0f' % kv for kv in sorted(by.items(), key=lambda kv: -kv[1])[:14]))
