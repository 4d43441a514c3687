# conv_regw's clock under its ablations: workgroup 0's lifetime in core cycles (s_memtime) and on the 100 MHz counter (s_memrealtime)
for t in 1 2; do for a in 0 14 1 15; do echo "== teams $t abl $a"; RLDM_RW_TEAMS=$t RLDM_RW_ABL=$a python tools/bench_conv.py --B 16 --vae --ts > /tmp/ts.log 2>&1; grep -E "vae.up2.conv[12] +2" /tmp/ts.log; grep "stamps.*\[0, " /tmp/ts.log | python3 -c "
import sys,re
for l in sys.stdin:
    v=[int(x) for x in re.findall(r'\d+', l.split('[',1)[1])]
    core, real = v[1], (v[3]-v[2])*0.01
    print(f'   workgroup 0: {core} core cycles in {real:.1f} us = {core/real:.0f} MHz')
"; done; done
