mkdir -p gpurun_out/r4a
for f in 0 256 512 1024; do
  echo "=== FLAGS2=$f (ablate timeline)"
  RLDM_LIB=$PWD/rangeldm_amd/librangeldm_hip_ablate.so RLDM_DBG_FLAGS2=$f python tools/bench_conv.py --ts --custom 16,128,0,256,16,128,3,1,0,1,0,1 --custom 16,128,0,512,32,128,3,1,0,1,128,0 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r4a/ts3.txt 2>&1
cat gpurun_out/r4a/ts3.txt
for f in 0 256 512 1024; do
  echo "=== FLAGS2=$f"
  RLDM_DBG_FLAGS2=$f python tools/bench_conv.py --B 16 --vae 2>&1 | grep -E "L0.*conv|vae.*conv[12]|vae.*upsample|sum over"
done > gpurun_out/r4a/conv3.txt 2>&1
cat gpurun_out/r4a/conv3.txt
tools/ab_env.sh 2 "RLDM_DBG_FLAGS2=0" "RLDM_DBG_FLAGS2=256" "RLDM_DBG_FLAGS2=512" 2>&1 | tee gpurun_out/r4a/ab3.txt
