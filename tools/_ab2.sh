set -e
MVT_STEM_PERSIST=1 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -x -q -m gpu -k "stem or encoder" 2>&1 | tail -3
export PROF_CONV_ITERS=41
for v in 0 1 0 1; do
  echo "== MVT_STEM_PERSIST=$v"
  MVT_STEM_PERSIST=$v python tools/prof_conv.py 24 512 512 4 64 7 2 3 1
  MVT_STEM_PERSIST=$v python tools/prof_conv.py 36 720 1280 4 64 7 2 3 1
done
for v in 0 1 0 1; do
  MVT_STEM_PERSIST=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-corr-calibration > gpurun_out/r4_sp_$v.json
  python -c "
import json; d=json.load(open('gpurun_out/r4_sp_$v.json')); print('persist=$v', round(d['ms_per_step'],3), round(d['roofline_mfma']['encoder']['frac'],4))"
done
