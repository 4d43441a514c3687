mkdir -p gpurun_out/r4g
python -m pytest tests/test_hip_kernels.py -x -q -k "64px-tiles or default" 2>&1 | tail -4 | tee gpurun_out/r4g/tests.txt
for f in 16 0; do
  echo "=== FLAGS2=$f"
  RLDM_DBG_FLAGS2=$f python tools/bench_conv.py --B 16 2>&1 | grep -E "L1.*conv|sum over"
done > gpurun_out/r4g/conv_l1.txt 2>&1
cat gpurun_out/r4g/conv_l1.txt
tools/ab_env.sh 3 "RLDM_DBG_FLAGS2=16" "-" 2>&1 | tee gpurun_out/r4g/ab_l1.txt
