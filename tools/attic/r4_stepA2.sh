mkdir -p gpurun_out/r4a
export RLDM_LIB=$PWD/rangeldm_amd/librangeldm_hip_ablate.so
for f in 7 0; do
  echo "=== FLAGS2=$f"
  RLDM_DBG_FLAGS2=$f python tools/bench_conv.py --ts --custom 16,128,0,256,16,128,3,1,0,1,0,1 --custom 16,128,0,512,32,128,3,1,0,1,128,0 --custom 16,64,0,1024,64,64,3,1,0,1,64,0 2>&1
done > gpurun_out/r4a/ts.txt 2>&1
cat gpurun_out/r4a/ts.txt
unset RLDM_LIB
RLDM_LIB=$PWD/rangeldm_amd/librangeldm_hip_prio.so python tools/bench_conv.py --B 16 --vae --only conv > gpurun_out/r4a/conv_prio.txt 2>&1
grep -E "L0|vae" gpurun_out/r4a/conv_prio.txt
tools/ab_libs.sh 2 default rangeldm_amd/librangeldm_hip_prio.so 2>&1 | tee gpurun_out/r4a/ab_prio.txt
