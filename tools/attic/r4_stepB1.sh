mkdir -p gpurun_out/r4b
tools/ab_env.sh 2 "RLDM_DBG_FLAGS2=1" "RLDM_DBG_FLAGS2=8" "RLDM_DBG_FLAGS2=$((1<<16))" "RLDM_DBG_FLAGS2=$((5<<16))" "RLDM_DBG_FLAGS2=0" "RLDM_DBG_FLAGS2=$((17<<16))" "RLDM_DBG_FLAGS2=$((25<<16))" 2>&1 | tee gpurun_out/r4b/ab_skew.txt
python -m pytest tests/test_hip_models.py -x -q 2>&1 | tail -8 | tee gpurun_out/r4b/tests_models.txt
