mkdir -p gpurun_out/r4c
python -m pytest tests/test_hip_kernels.py -x -q -k "specialised" 2>&1 | tail -5 | tee gpurun_out/r4c/tests_spec.txt
for f in 7 39 0; do
  echo "=== FLAGS2=$f"
  RLDM_DBG_FLAGS2=$f python tools/bench_conv.py --B 16 --vae 2>&1 | grep -E "L0.*conv|L1.upsample|vae.*conv[12]|vae.*upsample|sum over"
done > gpurun_out/r4c/conv_spec.txt 2>&1
cat gpurun_out/r4c/conv_spec.txt
for f in 7 39; do
  echo "=== FLAGS2=$f (ablate timeline)"
  RLDM_LIB=$PWD/rangeldm_amd/librangeldm_hip_ablate.so RLDM_DBG_FLAGS2=$f python tools/bench_conv.py --ts --custom 16,128,0,256,16,128,3,1,0,1,0,1 --custom 16,128,0,512,32,128,3,1,0,1,128,0 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r4c/ts_spec.txt 2>&1
grep -E "===|block 0|CU 0x0|custom" gpurun_out/r4c/ts_spec.txt
