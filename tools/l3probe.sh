#!/bin/bash
# level-3 conv 256->256 (+identity residual, temb) at batch 16: tile / GroupNorm-placement variants
CASE="16,256,0,32,2,256,3,1,0,1,256,1"
CASEP="16,256,0,32,2,256,3,1,0,0,256,1"
CASEU="16,512,0,32,2,256,3,1,0,0,512,1"
PW="16,256,0,32,2,256,1,1,0,0,256,0"
for cfg in "gn-fused,32px::$CASE" "gn-fused,64px,1view:RLDM_FAKE_VIEWS=1:$CASE" "preact,32px::$CASEP" "preact,64px,1view:RLDM_FAKE_VIEWS=1:$CASEP" "preact,64px,2views:RLDM_FAKE_VIEWS=2:$CASEP" "preact,64px,0view(flag524288):RLDM_DBGX=1:$CASEP" "up preact 32px::$CASEU" "up preact 64px 1view:RLDM_FAKE_VIEWS=1:$CASEU" "pw 32px::$PW" "pw 64px 1view:RLDM_FAKE_VIEWS=1:$PW" "pw 64px 2view:RLDM_FAKE_VIEWS=2:$PW"; do
  name=${cfg%%:*}; rest=${cfg#*:}; envs=${rest%%:*}; c=${rest#*:}
  dbg=0; [ "$envs" = "RLDM_DBGX=1" ] && { dbg=524288; envs=""; }
  echo "== $name"
  env $envs python tools/bench_conv.py --custom $c --iters 50 --dbg $dbg 2>&1 | grep custom
  env $envs RLDM_LIB=$PWD/rangeldm_amd/librangeldm_hip_ablate.so python tools/bench_conv.py --custom $c --iters 20 --dbg $dbg --ts 2>&1 | grep -E "block 0|workgroups"
done
