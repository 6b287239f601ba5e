run() { echo "== $*"; env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-corr-calibration | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['ms_per_step'],3), round(d['roofline_mfma']['encoder']['frac'],4), 'corr', round(r['frac'],3))"; }
run MVT_CONV_BIG=0
run MVT_CONV_BIG_SHARED=0
run MVT_CONV_BIG_SHARED=0 MVT_ENC_STREAMS=1
run MVT_CONV_BIG_SHARED=0 MVT_ENC_CHUNK=48
run MVT_CONV_BIG_SHARED=1
run MVT_CONV_BIG_SHARED=1 MVT_ENC_STREAMS=1
run MVT_CONV_BIG=0
