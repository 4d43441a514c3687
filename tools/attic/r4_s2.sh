mkdir -p gpurun_out/r4h
python -m pytest tests/test_hip_kernels.py -x -q -k "conv_geometry" 2>&1 | tail -4 | tee gpurun_out/r4h/tests.txt
for f in 128 0; do
  echo "=== FLAGS2=$f"
  RLDM_DBG_FLAGS2=$f python tools/bench_conv.py --B 16 2>&1 | grep -E "downsample|sum over"
done > gpurun_out/r4h/conv_s2.txt 2>&1
cat gpurun_out/r4h/conv_s2.txt
tools/ab_env.sh 3 "RLDM_DBG_FLAGS2=128" "-" 2>&1 | tee gpurun_out/r4h/ab_s2.txt
python -m pytest tests/test_hip_models.py -x -q 2>&1 | tail -3
