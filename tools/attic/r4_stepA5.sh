mkdir -p gpurun_out/r4a
for f2 in 512 0 7; do for d in 0 32768 65536 98304; do
  echo "=== FLAGS2=$f2 dbg=$d"
  RLDM_LIB=$PWD/rangeldm_amd/librangeldm_hip_ablate.so RLDM_DBG_FLAGS2=$f2 python tools/bench_conv.py --dbg $d --ts --custom 16,128,0,256,16,128,3,1,0,1,0,1 2>&1 | grep -E "block [02] stamps|custom"
done; done > gpurun_out/r4a/ts5.txt 2>&1
cat gpurun_out/r4a/ts5.txt
