#!/usr/bin/env python
"""Build profiles/traffic.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs).

    python tools/make_traffic.py profiles/r01_d_pmc_fetch_size.csv profiles/r01_d_pmc_write_size.csv > profiles/traffic.json
    (arguments: the counter_collection.csv files of the two passes, or the rocprofv3 output directories holding them)

Per MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in KB; on gfx950 FETCH_SIZE under-reports by 2x
(64-byte requests counted as 32) and is doubled here; WRITE_SIZE is used as is. Values are means over the launches of a
kernel instantiation within the profiled run, keyed by the bench label of that instantiation (only instantiations that
serve exactly one label are listed; shared ones are keyed by their C++ name).
"""
import collections
import csv
import glob
import json
import sys

LABEL = {
    # split-bf16 kernels (round-1 'd' profiles)
    'bconvu_kernel<25, 5, 2, 1, 4>': 'bconv<1ph,256px,N32>',
    'bconvu_kernel<25, 5, 1, 1, 2>': 'bconv<1ph,64px,N32>',
    'bconv_kernel<4, 1, 1, 4>': 'bconv<4ph,128px,N32>',
    'wgrad_b3_kernel<7, 1, 4>': 'wgrad_b3<5x5,K32>',
    'wgrad_b3_kernel<7, 1, 8>': 'wgrad_b3<5x5,K32>/stride2',
    'wgrad_b3_kernel<7, 2, 4>': 'wgrad_b3<5x5,K64>',
    'wgrad_b3_kernel<7, 2, 8>': 'wgrad_b3<5x5,K64>/stride2',
    'fc_stream_b3_kernel<false, 2>': 'fc_stream_b3<fwd>',
    'fc_stream_b3_kernel<true, 2>': 'fc_stream_b3<dgrad>',
    'mv3d::fc_wgrad_b3_kernel': 'fc_wgrad_b3',
    'mv3d::bconv_split_all_kernel': 'bconv_split_all',
    'thin_deconv_s2_tile_kernel<5, 2, false>': 'thin_deconv_s2<2>',
    'smallc_b3_kernel<1, 5>': 'smallc_img2feat<N32>',
    'wgrad_b3_kernel<5, 2, 4>': 'wgrad_b3<3x3,K64>',
    'wgrad_b3_kernel<5, 2, 8>': 'wgrad_b3<3x3,K64>/stride2',
    'bconvu_kernel<9, 3, 1, 1, 4>': 'bconv<small-img,128px,N32>/3x3',
    'bconv_kernel<1, 1, 1, 4>': 'bconv<small-img,128px,N32>/deconv',
    'bconvu_kernel<25, 5, 1, 1, 4>': 'bconv<1ph,128px,N32>',
    'bconvu_kernel<25, 5, 1, 2, 2>': 'bconv<1ph,64px,N64>',
    # exact-fp32 kernels (round-1 'b' / 'c' profiles, MV3D_DISABLE=4096)
    'hconvp_kernel<25, 1, 1, false, 9, 8, 25, 25, 25>': 'hconvp<5x5,256px,N32,nmajorB>',
    'hconvp_kernel<25, 1, 1, false, 13, 4, 25, 25, 25>': 'hconvp<5x5,128px,N32,nmajorB>',
    'hconvp_kernel<25, 1, 2, false, 13, 4, 25, 25, 25>': 'hconvp<5x5,128px,N64,nmajorB>',
    'hconv_kernel<4, 1, 1, false, 4>': 'hconv<4ph,128px,N32,nmajorB>',
    'hconv_kernel<1, 1, 2, false, 4>': 'hconv<1ph,128px,N64,nmajorB>',
    'wgrad_tile_kernel<7, 1>': 'wgrad_tile<5x5,K32>',
    'wgrad_tile_kernel<7, 2>': 'wgrad_tile<5x5,K64>',
    'wgrad_tile_kernel<5, 2>': 'wgrad_tile<3x3,K64>',
    'fc_stream_kernel<false, 2>': 'fc_stream<fwd>',
    'fc_stream_kernel<true, 2>': 'fc_stream<dgrad>',
    'mv3d::fc_wgrad_kernel': 'fc_wgrad',
    # both
    'mv3d::adam_kernel': 'adam',
    'mv3d::reduce_slabs_kernel': 'reduce_slabs',
    'reduce_slabs_kernel<4>': 'reduce_slabs',
    'reduce_slabs_kernel<1>': 'reduce_slabs/scalar',
    'mv3d::igemm_splitk_epilogue': 'igemm_splitk_epilogue',
    'igemm_splitk_epilogue<4>': 'igemm_splitk_epilogue',
    'igemm_splitk_epilogue<1>': 'igemm_splitk_epilogue/scalar',
    'resample_kernel<false>': 'resample_fwd',
    'resample_tile_kernel<2, 3>': 'resample_loss',
    'resample_tile_kernel<0, 3>': 'resample_fwd',
    'resample_tile_kernel<1, 3>': 'resample_bwd',
    'resample_kernel<true>': 'resample_bwd',
    'mv3d::pixel_loss_kernel': 'pixel_loss',
    'thin_deconv_s2_kernel<5, 2>': 'thin_deconv_s2<2>',
    'smallc_img2feat_kernel<1>': 'smallc_img2feat<N32>',
}


COUNTS = {}


def means(d, counter):
    f = d if d.endswith('.csv') else glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'mv3d' not in k or r['Counter_Name'] != counter:
            continue
        k = k.replace('void mv3d::', '').split('(')[0]
        acc.setdefault(k, []).append(float(r['Counter_Value']))
    steps = max(len(acc.get('mv3d::bconv_split_all_kernel', [])), 1)       # one filter conversion per step
    for k, v in acc.items():
        COUNTS[k] = len(v) / steps
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch = means(sys.argv[1], 'FETCH_SIZE')
write = means(sys.argv[2], 'WRITE_SIZE')
out = collections.OrderedDict()
for k in fetch:
    fb = int(fetch[k] * 1024 * 2)
    wb = int(write.get(k, 0.0) * 1024)
    out[LABEL.get(k, k)] = {
        'hbm_bytes_per_launch': fb + wb, 'fetch_bytes_x2_corrected': fb, 'write_bytes': wb, 'cxx_kernel': k,
        'launches_per_step': round(COUNTS.get(k, 0), 2),
        'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KB units, FETCH_SIZE doubled per '
                  "MI355X_MICROARCH.md HBM section; mean over the kernel's launches"}
json.dump(out, sys.stdout, indent=1)
print()
