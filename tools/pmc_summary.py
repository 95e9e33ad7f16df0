#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel: mean counter values and duration."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if 'mv3d' not in k:
        continue
    k = k.replace('void mv3d::', '').split('(')[0]
    a = agg.setdefault(k, collections.OrderedDict())
    a.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    a.setdefault('dur_us', []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    a['meta'] = 'grid=%s wg=%s lds=%s vgpr=%s agpr=%s sgpr=%s scratch=%s' % (r['Grid_Size'], r['Workgroup_Size'], r['LDS_Block_Size'], r['VGPR_Count'], r['Accum_VGPR_Count'], r['SGPR_Count'], r['Scratch_Size'])
for k, a in agg.items():
    print(k, '|', a.pop('meta'))
    for c, v in a.items():
        print('   %-28s n=%-3d mean=%.6g' % (c, len(v), sum(v) / len(v)))
