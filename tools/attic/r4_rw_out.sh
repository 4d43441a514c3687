for f in 67108864 83886080; do echo "== flags2 $f"; RLDM_DBG_FLAGS2=$f python tools/bench_conv.py --B 16 --vae 2>&1 | grep -E "vae.conv_out|vae.up2.conv[12] +2"; done
