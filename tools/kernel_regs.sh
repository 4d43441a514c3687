#!/bin/bash
# VGPR / spill / scratch of every kernel in one .hip file: tools/kernel_regs.sh rangeldm_amd/csrc/conv_small.hip
f=$1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm -Wno-array-bounds $EXTRA --cuda-device-only -S "$f" -o /tmp/_regs.s -I$(dirname $f) 2>/dev/null
python3 - <<'PY'
import re
txt=open('/tmp/_regs.s').read()
for m in re.finditer(r'- \.agpr_count:.*?\.wavefront_size', txt, re.S):
    blk=m.group(0)
    g=lambda k: (re.search(r'\.'+k+r':\s*(\S+)', blk) or [None,'?'])[1]
    print(g('name')[:90], 'vgpr', g('vgpr_count'), 'spill', g('vgpr_spill_count'), 'scratch', g('private_segment_fixed_size'), 'sgpr', g('sgpr_count'))
PY
