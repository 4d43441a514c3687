set -x
mkdir -p gpurun_out/r4a
python -m pytest tests/test_hip_kernels.py -x -q -k "conv_geometry or conv_gn" 2>&1 | tail -15 > gpurun_out/r4a/tests.txt
cat gpurun_out/r4a/tests.txt
RLDM_DBG_FLAGS2=7 python tools/bench_conv.py --B 16 --vae > gpurun_out/r4a/conv_old.txt 2>&1
RLDM_DBG_FLAGS2=0 python tools/bench_conv.py --B 16 --vae > gpurun_out/r4a/conv_new.txt 2>&1
paste -d'\n' gpurun_out/r4a/conv_old.txt gpurun_out/r4a/conv_new.txt | grep -v "^$" | head -120
tools/ab_env.sh 2 "RLDM_DBG_FLAGS2=7" "RLDM_DBG_FLAGS2=7 RLDM_DBG_FLAGS=268435456" "RLDM_DBG_FLAGS2=0" "RLDM_DBG_FLAGS2=1" "RLDM_DBG_FLAGS2=3" 2>&1 | tee gpurun_out/r4a/ab.txt
