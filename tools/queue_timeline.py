#!/usr/bin/env python3
"""per-hardware-queue view of the last step of a rocprofv3 --kernel-trace CSV: which kernels run where, and the idle
gaps of the busiest (main) queue.  usage: tools/queue_timeline.py <dir> <prefix> <steps>"""
import csv, re, sys, collections
d, pre, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
def short(n):
    n = re.sub(r'\(.*', '', n).replace('void ', '').replace('mmdgan::', '')
    if 'at::native' in n: n = 'torch:' + re.sub(r'.*::', '', n.split('<')[0])
    return n[:44]
tr = list(csv.DictReader(open('%s/%s_kernel_trace.csv' % (d, pre))))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
# one step = from one mmd_kernel to the next
idx = [i for i, r in enumerate(tr) if 'mmd_kernel' in r['Kernel_Name']]
lo, hi = idx[-2] + 1, idx[-1] + 1            # kernels after the previous loss kernel up to this one's ... shifted window
# use adam as the step boundary instead: the last two adam launches end a step
ad = [i for i, r in enumerate(tr) if 'adam_kernel' in r['Kernel_Name']]
end = ad[-1] + 1
start = ad[-3] + 1
step = tr[start:end]
t0 = int(step[0]['Start_Timestamp'])
print('last step: %d launches, %.3f ms wall' % (len(step), (int(step[-1]['End_Timestamp']) - t0) / 1e6))
byq = collections.defaultdict(list)
for r in step:
    byq[r['Queue_Id']].append(r)
for q, rows in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows)
    print('queue %s: %3d launches, busy %.3f ms, first +%.3f last +%.3f  e.g. %s' % (
        q, len(rows), busy / 1e6, (int(rows[0]['Start_Timestamp']) - t0) / 1e6, (int(rows[-1]['End_Timestamp']) - t0) / 1e6,
        ', '.join(sorted({short(r['Kernel_Name']) for r in rows})[:4])))
main = max(byq.values(), key=len)
print('\nidle gaps > 8 us on the busiest queue:')
prev_end, prev = int(main[0]['End_Timestamp']), main[0]
tot = 0
for r in main[1:]:
    s = int(r['Start_Timestamp'])
    if s - prev_end > 8000:
        print('  +%.3f ms: %6.1f us idle between %-36s and %s' % ((prev_end - t0) / 1e6, (s - prev_end) / 1e3, short(prev['Kernel_Name']), short(r['Kernel_Name'])))
    if s > prev_end: tot += s - prev_end
    if int(r['End_Timestamp']) > prev_end: prev_end, prev = int(r['End_Timestamp']), r
print('total idle on it: %.3f ms' % (tot / 1e6))
