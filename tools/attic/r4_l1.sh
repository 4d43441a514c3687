mkdir -p gpurun_out/r4d
RLDM_LIB=$PWD/rangeldm_amd/librangeldm_hip_ablate.so python tools/bench_conv.py --ts --custom 16,128,0,128,8,128,3,1,0,1,0,1 --custom 16,128,0,128,8,128,3,1,0,1,128,0 --custom 16,256,128,128,8,128,3,1,0,1,0,1 --custom 16,128,0,128,8,128,3,1,0,1,384,0 2>&1 | grep -E "block 0|CU 0x0|custom|workgroups" > gpurun_out/r4d/l1_ts.txt
cat gpurun_out/r4d/l1_ts.txt
