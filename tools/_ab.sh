timeout -k 10 200 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "c64_persistent" 2>&1 | tail -3
for rep in 1 2; do
for on in 1 0; do
  export MVT_CONV_C64=$on; echo "== MVT_CONV_C64=$on"
  timeout -k 10 60 python tools/prof_conv.py 24 256 256 64 64 3 1 1 1
  timeout -k 10 60 python tools/prof_conv.py 24 256 256 64 64 3 1 1 1 0
done
done
unset MVT_CONV_C64
STAMP_RAW=1 STAMP_NORM=1 MVT_LIB=mvtracker_amd/lib/libmvtracker_hip_stamps.so timeout -k 5 100 python tools/stamp_conv.py 24 256 256 64 64 > gpurun_out/r4_stamp_c64b.txt 2>&1
