#!/bin/bash
# level-2 (64x4 images, batch 16) conv_small launches with in-kernel stamps (ABLATE build)
for cfg in "down conv1/2 gn-fused 256->256 +idres:16,256,0,64,4,256,3,1,0,1,256,1" "up conv1 preact 512->256:16,512,0,64,4,256,3,1,0,0,0,1" "up conv2 gn-fused 256->256 + shortcut 512:16,256,0,64,4,256,3,1,0,1,512,0" "up conv1 preact 384->256:16,384,0,64,4,256,3,1,0,0,0,1" "pw 256->256:16,256,0,64,4,256,1,1,0,0,256,0"; do
  name=${cfg%%:*}; c=${cfg#*:}
  echo "== $name"
  python tools/bench_conv.py --custom $c --iters 50 2>&1 | grep custom
  RLDM_LIB=$PWD/rangeldm_amd/librangeldm_hip_ablate.so python tools/bench_conv.py --custom $c --iters 20 --ts 2>&1 | grep -E "block 0|workgroups"
done
