python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv" 2>&1 | tail -5
for big in 1 0; do
  export MVT_CONV_BIG=$big; echo "== MVT_CONV_BIG=$big"
  python tools/prof_conv.py 24 128 128 416 256 3 1 1 1 0
  python tools/prof_conv.py 48 180 320 416 256 3 1 1 1 0
done
