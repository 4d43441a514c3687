#!/bin/bash
# Round 3: where a 128x8 (level 1) conv's time goes, in three settings -- evidence for DESIGN.md 3.7's open question (a 128x8 phase's K
# loop inside the persistent launch against the same conv alone).  ABLATE build; writes gpurun_out/r3_l1_probe.txt.
# Stamps of conv_stream_body: entry | stream ptr | lambdas [| cluster wait] | ring issued | GN fold | chunk 0 stored | K loop start |
#                             K loop end | epilogue LDS | epilogue end (| arrive)
export RLDM_LIB=$PWD/rangeldm_amd/librangeldm_hip_ablate.so
OUT=gpurun_out/r3_l1_probe.txt
mkdir -p gpurun_out
: > $OUT
echo "== E1: stand-alone launches inside the batch-16 UNet forward (conv ordinals 6, 7 = down.1.resnets.0 conv1 / conv2)" >> $OUT
for o in 6 7; do echo "-- ord $o" >> $OUT; RLDM_TS_ORD=$o python tools/trunk_timeline.py >> $OUT 2>&1; done
echo "== E2: the same pair as a 2-phase persistent launch (flag 1 << 30), first such launch of the plan" >> $OUT
RLDM_DBG_FLAGS=$((1<<30)) RLDM_TS_TRUNK=2 RLDM_TS_TRUNK_FIRST=1 python tools/trunk_timeline.py >> $OUT 2>&1
echo "== E3: the conv alone (tools/bench_conv.py), weights + input warm in L2 / swept out of the L2s (96 MB) / out of the Infinity Cache (768 MB)" >> $OUT
for mb in 0 96 768; do
  echo "-- thrash $mb MB" >> $OUT
  RLDM_BENCH_THRASH_MB=$mb python tools/bench_conv.py --B 16 --custom 16,128,0,128,8,128,3,1,0,1,0,1 --custom 16,128,0,128,8,128,3,1,0,1,128,0 --ts >> $OUT 2>&1
done
echo "== RangeDM 512-channel levels: the 1024-input-channel convs, split-K sweep of the generic kernel (B = 1 and 4)" >> $OUT
for B in 1 4; do
python tools/bench_conv.py --B $B --custom $B,512,512,64,4,512,3,1,0,1,0,1 --custom $B,512,0,64,4,512,3,1,0,1,1024,0 --custom $B,512,512,32,2,512,3,1,0,1,0,1 --custom $B,512,0,32,2,512,3,1,0,1,1024,0 --custom $B,512,256,128,8,256,3,1,0,1,0,1 --tiles 0x0,64x64x2,64x64x4,64x64x8,64x64x16,64x32x8,64x32x16 >> $OUT 2>&1
done
cat $OUT
