#!/usr/bin/env python
"""Per-step stream timeline from a rocprofv3 --kernel-trace CSV directory: python tools/timeline.py <dir> [--all]"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'bconv_split_all' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
step = rows[a:b]
t0 = int(step[0]['Start_Timestamp'])
print('step span %.1f us, kernels %d' % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3, len(step)))
qs = collections.OrderedDict()
for r in step:
    qs.setdefault(r['Queue_Id'], []).append(r)
for q, l in qs.items():
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in l)
    print('queue', q, 'kernels', len(l), 'busy %.1f us' % (busy / 1e3), 'first %.1f last %.1f' % ((int(l[0]['Start_Timestamp']) - t0) / 1e3, (int(l[-1]['End_Timestamp']) - t0) / 1e3))
if '--all' in sys.argv:
    for r in step:
        s = (int(r['Start_Timestamp']) - t0) / 1e3; e = (int(r['End_Timestamp']) - t0) / 1e3
        print('%7.1f %7.1f %6.1f q%s %s' % (s, e, e - s, r['Queue_Id'], r['Kernel_Name'].replace('void mv3d::', '').replace('mv3d::', '')[:48]))
