mkdir -p gpurun_out/r4f
python -m pytest tests/test_hip_kernels.py -x -q -k "conv_geometry or conv_gn" 2>&1 | tail -4 | tee gpurun_out/r4f/tests.txt
for l in librangeldm_hip_nofg.so librangeldm_hip.so; do for f in 7 0; do
  echo "=== $l FLAGS2=$f"
  RLDM_LIB=$PWD/rangeldm_amd/$l RLDM_DBG_FLAGS2=$f python tools/bench_conv.py --B 16 --vae 2>&1 | grep -E "L0.*conv|L1.upsample|vae.*conv[12]|vae.*upsample|sum over"
done; done > gpurun_out/r4f/conv_fg.txt 2>&1
cat gpurun_out/r4f/conv_fg.txt
tools/ab_libs.sh 3 rangeldm_amd/librangeldm_hip_nofg.so default 2>&1 | tee gpurun_out/r4f/ab_fg.txt
