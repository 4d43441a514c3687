mkdir -p gpurun_out/r4d
python -m pytest tests/test_hip_kernels.py -x -q -k "conv_geometry or conv_gn" 2>&1 | tail -3
for l in librangeldm_hip_nopf2.so librangeldm_hip.so; do
  echo "=== $l"
  RLDM_LIB=$PWD/rangeldm_amd/$l python tools/bench_conv.py --B 16 2>&1 | grep -E "L1.*conv|L2.upsample|sum over"
done > gpurun_out/r4d/conv_pf2.txt 2>&1
cat gpurun_out/r4d/conv_pf2.txt
tools/ab_libs.sh 3 rangeldm_amd/librangeldm_hip_nopf2.so default 2>&1 | tee gpurun_out/r4d/ab_pf2.txt
