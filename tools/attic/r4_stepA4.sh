mkdir -p gpurun_out/r4a
for f in 0 512 $((8*65536)) $((16*65536)); do
  echo "=== FLAGS2=$f (ablate timeline, ILV=1)"
  RLDM_LIB=$PWD/rangeldm_amd/librangeldm_hip_ablate.so RLDM_DBG_FLAGS2=$f python tools/bench_conv.py --ts --custom 16,128,0,256,16,128,3,1,0,1,0,1 --custom 16,128,0,512,32,128,3,1,0,1,128,0 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r4a/ts4.txt 2>&1
cat gpurun_out/r4a/ts4.txt
for l in librangeldm_hip_noilv.so librangeldm_hip.so; do for f in 7 0 $((8*65536)); do
  echo "=== $l FLAGS2=$f"
  RLDM_LIB=$PWD/rangeldm_amd/$l RLDM_DBG_FLAGS2=$f python tools/bench_conv.py --B 16 --vae 2>&1 | grep -E "L[01].*conv|L.\.upsample|vae.*conv[12]|vae.*upsample|sum over"
done; done > gpurun_out/r4a/conv4.txt 2>&1
cat gpurun_out/r4a/conv4.txt
tools/ab_libs.sh 2 rangeldm_amd/librangeldm_hip_noilv.so default 2>&1 | tee gpurun_out/r4a/ab4.txt
RLDM_DBG_FLAGS2=7 tools/ab_libs.sh 2 rangeldm_amd/librangeldm_hip_noilv.so default 2>&1 | tee gpurun_out/r4a/ab4_old.txt
