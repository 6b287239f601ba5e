python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv or encoder" 2>&1 | tail -2
for rep in 1 2; do
for lib in "" nopack; do
  if [ -n "$lib" ]; then export MVT_LIB=mvtracker_amd/lib/libmvtracker_hip_$lib.so; else unset MVT_LIB; fi
  echo "== ${lib:-new(4 waves 1x1)}"
  python tools/prof_conv.py 24 128 128 256 128 1 1 0 1
  python tools/prof_conv.py 24 256 256 64 96 1 2 0 1
  python tools/prof_conv.py 24 128 128 96 128 1 2 0 1
  python tools/prof_conv.py 24 64 64 128 128 1 2 0 1
done
done
