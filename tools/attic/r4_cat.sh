python -m pytest tests/test_hip_kernels.py -x -q -k "conv_gn" 2>&1 | tail -3
python -m pytest tests/test_hip_models.py -x -q 2>&1 | tail -3
tools/ab_env.sh 3 "RLDM_SMALL_CONCAT_GN=0" "-" 2>&1
