#!/usr/bin/env python3
"""Per-kernel table of one bench.py line (stdin): share, launches per batch, average us, TFLOP/s, algorithmic GB/s."""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(f"{d['value']:.2f} {d['unit']}  ({d['ms_per_step']:.3f} ms per batch)")
def find(o):
    if isinstance(o, dict):
        if "kernels" in o and isinstance(o["kernels"], dict):
            return o["kernels"]
        for v in o.values():
            r = find(v)
            if r:
                return r
    return None
for k, v in (find(d) or {}).items():
    print(f"{k:62s} {v}")
