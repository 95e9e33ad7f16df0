#!/usr/bin/env python
"""Build profiles/traffic.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs).

    python tools/make_traffic.py <fetch dir or csv> <write dir or csv> > profiles/traffic.json

Per MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in KB; on gfx950 FETCH_SIZE under-reports wide coalesced
reads by 2x (128-byte requests tallied at 64) and is doubled here; WRITE_SIZE is used as is.  Output: one entry per C++ kernel
instantiation (mean bytes per launch, launches per step) and `_family`: HBM bytes per STEP of the conv / deconv family (the
kernels bench.py's `roofline` aggregates) and of everything else.  A step is delimited by the one filter conversion launch
(bconv_split_all_kernel) it starts with.
"""
import collections
import csv
import glob
import json
import sys

CONV_FAMILY = ('cconv_kernel', 'cwgrad_kernel', 'bconv', 'sconv', 's2conv', 'wgrad_b3', 'wgrad_tile', 'hconv', 'igemm', 'smallc_', 'thin_', 'filtgrad',
               'reduce_slabs', 'grad_finalize', 'transpose_filter')


def family(k):
    if 'fc_' in k:
        return 'fc'
    return 'conv' if any(t in k for t in CONV_FAMILY) else 'other'


def load(d, counter):
    f = d if d.endswith('.csv') else glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'mv3d' not in k or r['Counter_Name'] != counter:
            continue
        k = k.replace('void mv3d::', '').split('(')[0]
        acc.setdefault(k, []).append(float(r['Counter_Value']))
    return acc


fetch = load(sys.argv[1], 'FETCH_SIZE')
write = load(sys.argv[2], 'WRITE_SIZE')
steps = max(len(fetch.get('mv3d::bconv_split_all_kernel', [])), 1)
out = collections.OrderedDict()
fam = collections.OrderedDict()
for k, v in fetch.items():
    fb = sum(v) / len(v) * 1024 * 2
    w = write.get(k, [0.0])
    wb = sum(w) / len(w) * 1024
    per_step = len(v) / steps
    out[k] = {'hbm_bytes_per_launch': int(fb + wb), 'fetch_bytes_x2_corrected': int(fb), 'write_bytes': int(wb),
              'launches_per_step': round(per_step, 2), 'family': family(k)}
    f = fam.setdefault(family(k), {'hbm_bytes_per_step': 0, 'fetch_bytes_per_step': 0, 'write_bytes_per_step': 0, 'launches_per_step': 0.0})
    f['hbm_bytes_per_step'] += int((fb + wb) * per_step)
    f['fetch_bytes_per_step'] += int(fb * per_step)
    f['write_bytes_per_step'] += int(wb * per_step)
    f['launches_per_step'] = round(f['launches_per_step'] + per_step, 2)
for f in fam.values():
    f['source'] = ('rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KB units, FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section; '
                   'sum over the family\'s kernels of mean bytes per launch x launches per step, %d profiled steps' % steps)
out['_family'] = fam
json.dump(out, sys.stdout, indent=1)
print()
