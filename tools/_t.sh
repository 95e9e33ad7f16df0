cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_layers.py -x -q -m gpu -k "layers_at or rung or fused_step or kernel_instances" > gpurun_out/r3_t19.log 2>&1 || { tail -30 gpurun_out/r3_t19.log; exit 1; }
tail -3 gpurun_out/r3_t19.log
for v in 0 512 0 512; do
  MV3D_TC_S1_BELOW=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-timing --steps 60 --warmup 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('S1_BELOW=$v  %d img/s %.4f ms median %.4f min %.4f launches %s'%(d['value'],d['ms_per_step'],d['step_ms']['median'],d['step_ms']['min'],d['config']['launches_per_step']))" >> gpurun_out/r3_sweep20.txt || exit 1
done
cat gpurun_out/r3_sweep20.txt
timeout -k 10 300 python bench.py --no-cpu-baseline --dump-kernels > gpurun_out/r3_b17.json 2> gpurun_out/r3_b17.err
grep -n "s2conv\|cconv" gpurun_out/r3_b17.err
python -c "
import json
d=json.loads(open('gpurun_out/r3_b17.json').read().strip().splitlines()[-1])
print(d['value'],d['ms_per_step'],d['roofline']['alone']['achieved'],d['roofline']['alone']['ms_per_step'])"
