python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv or encoder" 2>&1 | tail -3
python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "encoder or golden" 2>&1 | tail -3
run() { echo "== $*"; env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-corr-calibration | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['ms_per_step'],3), round(d['roofline_mfma']['encoder']['frac'],4), 'corr', round(r['frac'],3))"; }
run MVT_FOLD_DOWNSAMPLE=1
run MVT_FOLD_DOWNSAMPLE=0
run MVT_FOLD_DOWNSAMPLE=1
run MVT_FOLD_DOWNSAMPLE=0
