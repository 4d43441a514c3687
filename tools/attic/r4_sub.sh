for B in 16 4 1; do for f in 0 134217728; do echo "== B $B flags2 $f"; RLDM_DBG_FLAGS2=$f RLDM_SUB_MIN=${SUBMIN:-96} python tools/bench_conv.py --B $B --vae 2>&1 | grep -E "upsample"; done; done
