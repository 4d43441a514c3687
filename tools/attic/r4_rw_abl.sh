for t in 1 2; do for a in 0 16 1 2 8 15 14; do echo "== teams $t abl $a"; RLDM_RW_TEAMS=$t RLDM_RW_ABL=$a python tools/bench_conv.py --B 16 --vae 2>&1 | grep -E "vae.up2.conv[12] +2"; done; done
