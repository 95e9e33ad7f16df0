run() { echo "== $*"; env "$@" python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['step_ms']['median'], d['step_ms']['min'])"; }
run A=1
run MV3D_SI_BLOCKS=256
run MV3D_SI_BLOCKS=128
run MV3D_CC_MINTILES=200
run MV3D_CC_MINTILES=1000
run MV3D_WG_CUS=128
run MV3D_SIDE_STREAMS=1
run MV3D_SIDE_STREAMS=3
run A=2
