python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('build+smoke in one process ok')" 2>&1 | tail -2
python -m pytest tests/test_hip_models.py -x -q 2>&1 | tail -3
tools/ab_libs.sh 3 rangeldm_amd/librangeldm_hip_noclpf.so default 2>&1
