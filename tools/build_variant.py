#!/usr/bin/env python
"""Experiment builds of libmv3d_hip.so: `python tools/build_variant.py NAME [--unit conv.hip] -DFLAG ...` compiles one unit of csrc/
(default cconv.hip) with the extra flags and links dynamic_multiview_3d_amd/variants/libmv3d_hip_NAME.so from it and the regular
objects (run with MV3D_LIB=<path>).
Results of such builds are wrong by construction when a flag removes work; they only time ablations."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamic_multiview_3d_amd import build as B
name, flags = sys.argv[1], sys.argv[2:]
unit = 'cconv.hip'
if flags and flags[0] == '--unit':
    unit, flags = flags[1], flags[2:]
B.build()
vdir = os.path.join(B.HERE, 'variants')
os.makedirs(vdir, exist_ok=True)
obj = os.path.join(vdir, '%s_%s.o' % (unit.replace('.hip', ''), name))
subprocess.run([B._hipcc()] + B.COMMON + flags + ['-c', os.path.join(B.CSRC, unit), '-o', obj], check=True)
objs = [os.path.join(B.HERE, 'build', s.replace('.hip', '.o')) for s in B.UNITS if s != unit] + [obj]
out = os.path.join(vdir, 'libmv3d_hip_%s.so' % name)
subprocess.run([B._hipcc(), '-shared', '-fPIC', '--offload-arch=gfx950', '-o', out] + objs, check=True)
print(out)
