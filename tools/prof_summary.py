#!/usr/bin/env python3
"""Summarise rocprofv3 --kernel-trace --stats CSV output into a short text table.
usage: tools/prof_summary.py <dir> <prefix> [steps]   (writes to stdout)"""
import csv, re, sys, collections
d, pre = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
def short(n):
    n = re.sub(r'\(.*', '', n)
    n = n.replace('void ', '').replace('mmdgan::', '')
    if 'at::native' in n: n = 'torch:' + re.sub(r'.*::', '', n.split('<')[0])
    return n[:60]
rows = list(csv.DictReader(open('%s/%s_kernel_stats.csv' % (d, pre))))
tot = sum(int(r['TotalDurationNs']) for r in rows)
print('total kernel time %.3f ms over %d steps = %.3f ms/step' % (tot / 1e6, steps, tot / 1e6 / steps))
print('%-62s %7s %10s %10s %6s' % ('kernel', 'calls', 'avg_us', 'ms/step', '%'))
for r in rows[:40]:
    print('%-62s %7s %10.1f %10.3f %6.2f' % (short(r['Name']), r['Calls'], float(r['AverageNs']) / 1e3,
                                          int(r['TotalDurationNs']) / 1e6 / steps, float(r['Percentage'])))
# per-launch view of the last step (by grid size)
tr = list(csv.DictReader(open('%s/%s_kernel_trace.csv' % (d, pre))))
if tr and '--trace' in sys.argv:
    tr.sort(key=lambda r: int(r['Start_Timestamp']))
    n = len(tr) // steps
    last = tr[-n:]
    t0 = int(last[0]['Start_Timestamp'])
    print('\nlast step, launch order (%d launches):' % len(last))
    for r in last:
        dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        print('%9.1f us  +%8.1f  grid %-18s wg %-5s %s' % (dur, (int(r['Start_Timestamp']) - t0) / 1e3,
              '%sx%sx%s' % (r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z']), r['Workgroup_Size_X'], short(r['Kernel_Name'])))
# union-of-intervals view of the last step: how much of the wall time has NO kernel running
if tr and '--timeline' in sys.argv:
    tr.sort(key=lambda r: int(r['Start_Timestamp']))
    n = len(tr) // steps
    last = tr[-n:]
    iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in last)
    t0, busy, cur_end, gaps = iv[0][0], 0, iv[0][0], []
    prev_name = ''
    for s, e, nm in iv:
        if s > cur_end:
            gaps.append((s - cur_end, prev_name, nm))
            cur_end = s
        if e > cur_end:
            busy += e - cur_end
            cur_end = e
            prev_name = nm
    wall = cur_end - t0
    print('\nlast step: %d launches, wall %.1f us, some kernel running %.1f us, idle %.1f us in %d gaps' % (
        len(iv), wall / 1e3, busy / 1e3, (wall - busy) / 1e3, len(gaps)))
    gaps.sort(reverse=True)
    for g, a, b in gaps[:25]:
        print('  gap %6.1f us   after %-40s before %s' % (g / 1e3, a[:40], b[:40]))
    hist = collections.Counter(min(int(g / 500), 10) for g, _, _ in gaps)
    print('  gap histogram (0.5 us bins):', sorted(hist.items()))
