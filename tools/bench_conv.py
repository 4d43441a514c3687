#!/usr/bin/env python3
"""Per-shape timing of the fused circular conv kernel (rldm_bench_conv) over every distinct conv of the RangeLDM UNet
and VAE decoder at batch B.  Prints launches-per-step, us, TFLOP/s and the share of one step.

usage: python tools/bench_conv.py [--B 16] [--tiles 128x128,64x64x4,...  (BMxBN[xksplit])] [--only substr]
"""
import argparse
import collections
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rangeldm_amd import _lib  # noqa: E402

Case = collections.namedtuple("Case", "name count B C0 C1 W H Cout k stride pad up gn silu res temb")


def unet_cases(B, W=256, H=16, boc=(128, 128, 256, 256), attn_down=(0, 1, 1, 1), attn_up=(1, 1, 1, 0), cin=5, cout=4):
    """(case, count) for one UNet forward; mirrors the plan builder's walk."""
    cnt = collections.Counter()

    def add(**kw):
        c = Case(**kw)
        cnt[c] += 1

    L = len(boc)

    def res(name, w, h, c0, c1, co):
        add(name=f"{name}.conv1", count=0, B=B, C0=c0, C1=c1, W=w, H=h, Cout=co, k=3, stride=1, pad=0, up=0, gn=1, silu=1,
            res=0, temb=1)
        # conv2 carries the shortcut as its residual K-phase: identity (res = co) or the 1x1 conv over the block input
        add(name=f"{name}.conv2", count=0, B=B, C0=co, C1=0, W=w, H=h, Cout=co, k=3, stride=1, pad=0, up=0, gn=1, silu=1,
            res=c0 + c1, temb=0)

    def attn(name, w, h, c):
        add(name=f"{name}.qkv", count=0, B=B, C0=c, C1=0, W=w, H=h, Cout=3 * c, k=1, stride=1, pad=0, up=0, gn=1, silu=0,
            res=0, temb=0)
        add(name=f"{name}.out", count=0, B=B, C0=c, C1=0, W=w, H=h, Cout=c, k=1, stride=1, pad=0, up=0, gn=0, silu=0,
            res=c, temb=0)

    w, h = W, H
    add(name="conv_in", count=0, B=B, C0=cin, C1=0, W=w, H=h, Cout=boc[0], k=3, stride=1, pad=0, up=0, gn=0, silu=0, res=0,
        temb=0)
    skips = [boc[0]]
    out = boc[0]
    for i in range(L):
        ci, out = out, boc[i]
        for j in range(2):
            res(f"L{i}.down", w, h, ci if j == 0 else out, 0, out)
            if attn_down[i]:
                attn(f"L{i}.attn", w, h, out)
            skips.append(out)
        if i != L - 1:
            add(name=f"L{i}.downsample", count=0, B=B, C0=out, C1=0, W=w, H=h, Cout=out, k=3, stride=2, pad=0, up=0, gn=0,
                silu=0, res=0, temb=0)
            w, h = w // 2, h // 2
            skips.append(out)
    res(f"L{L-1}.mid", w, h, out, 0, out)
    attn(f"L{L-1}.attn", w, h, out)
    res(f"L{L-1}.mid", w, h, out, 0, out)
    for i in range(L):
        lvl = L - 1 - i
        prev, out = out, boc[lvl]
        for j in range(3):
            sk = skips.pop()
            res(f"L{lvl}.up", w, h, prev if j == 0 else out, sk, out)
            if attn_up[i]:
                attn(f"L{lvl}.attn", w, h, out)
        if i != L - 1:
            add(name=f"L{lvl}.upsample", count=0, B=B, C0=out, C1=0, W=w, H=h, Cout=out, k=3, stride=1, pad=0, up=1, gn=0,
                silu=0, res=0, temb=0)
            w, h = w * 2, h * 2
    add(name="conv_out", count=0, B=B, C0=boc[0], C1=0, W=w, H=h, Cout=cout, k=3, stride=1, pad=0, up=0, gn=1, silu=1, res=0,
        temb=0)
    # merge cases that differ only by name
    merged = collections.OrderedDict()
    for c, n in cnt.items():
        key = c._replace(name="")
        if key in merged:
            merged[key] = (merged[key][0], merged[key][1] + n)
        else:
            merged[key] = (c.name, n)
    return [(k._replace(name=v[0]), v[1]) for k, v in merged.items()]


def vae_decoder_cases(B, W=256, H=16, ch=64, mult=(1, 2, 4), z=4, cout=2):
    cnt = collections.OrderedDict()

    def add(name, c0, w, h, co, k=3, up=0, gn=0, silu=0, res=0):
        c = Case(name="", count=0, B=B, C0=c0, C1=0, W=w, H=h, Cout=co, k=k, stride=1, pad=0, up=up, gn=gn, silu=silu, res=res,
                 temb=0)
        if c in cnt:
            cnt[c] = (cnt[c][0], cnt[c][1] + 1)
        else:
            cnt[c] = (name, 1)

    def res(name, w, h, ci, co):
        add(name + ".conv1", ci, w, h, co, gn=1, silu=1)
        add(name + ".conv2", co, w, h, co, gn=1, silu=1, res=ci)

    chs = [ch * m for m in mult]
    c = chs[-1]
    w, h = W, H
    add("vae.conv_in", z, w, h, c)
    res("vae.mid", w, h, c, c)
    res("vae.mid", w, h, c, c)
    for i in range(len(mult)):
        co = chs[len(mult) - 1 - i]
        for j in range(3):
            res(f"vae.up{i}", w, h, c, co)
            c = co
        if i != len(mult) - 1:
            add(f"vae.up{i}.upsample", c, w, h, c, up=1)
            w, h = 2 * w, 2 * h
    add("vae.conv_out", c, w, h, cout, gn=1, silu=1)
    return [(k._replace(name=v[0]), v[1]) for k, v in cnt.items()]


def bench_case(c, iters=20, warmup=3):
    import torch
    d = _lib.ConvDescC()
    d.B, d.Cin0, d.Cin1, d.Win, d.Hin = c.B, c.C0, c.C1, c.W, c.H
    d.Cout, d.ksize, d.stride, d.pad_mode, d.upsample = c.Cout, c.k, c.stride, c.pad, c.up
    d.gn, d.silu, d.eps = c.gn, c.silu, 1e-5
    us = C.c_float(0)
    name = C.create_string_buffer(128)
    _lib.check(_lib.lib().rldm_bench_conv(C.byref(d), c.res, c.temb, warmup, iters, C.byref(us), name, 128,
                                          _lib.stream_ptr(torch.device("cuda"))), "rldm_bench_conv")
    return us.value, name.value.decode()


def flops(c):
    up = 2 if c.up else 1
    wo, ho = c.W * up // c.stride, c.H * up // c.stride
    sc = c.res if c.res != c.Cout else 0        # identity residual: no algorithmic FLOPs
    return 2.0 * c.B * wo * ho * c.Cout * ((c.C0 + c.C1) * c.k * c.k + sc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=16)
    ap.add_argument("--tiles", default="")
    ap.add_argument("--only", default="")
    ap.add_argument("--vae", action="store_true")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dbg", type=int, default=0)
    ap.add_argument("--custom", action="append", default=[],
                    help="extra case B,C0,C1,W,H,Cout,k,stride,up,gn,res,temb (repeatable; replaces the network's list)")
    ap.add_argument("--ts", action="store_true", help="print in-kernel s_memtime stamps (deltas, cycles) of block 0")
    a = ap.parse_args()
    _lib.require_gpu()
    _lib.lib().rldm_debug_set_flags(a.dbg)
    if a.ts:
        _lib.lib().rldm_debug_timestamps(None)
    cases = unet_cases(a.B)
    if a.vae:
        cases += vae_decoder_cases(a.B)
    if a.custom:
        cases = []
        for spec in a.custom:
            v = list(map(int, spec.split(",")))
            cases.append((Case(name="custom", count=0, B=v[0], C0=v[1], C1=v[2], W=v[3], H=v[4], Cout=v[5], k=v[6], stride=v[7],
                               pad=0, up=v[8], gn=v[9], silu=v[9], res=v[10], temb=v[11]), 1))
    tiles = [tuple(map(int, t.split("x"))) for t in a.tiles.split(",") if t] or [(0, 0, 0)]
    tiles = [t if len(t) == 3 else (t[0], t[1], 0) for t in tiles]
    tot = collections.defaultdict(float)
    print(f"{'case':18s} {'n':>3s} {'C0+C1':>9s} {'WxH':>8s} {'N':>4s} k s u g   r | " +
          " | ".join(f"{('auto' if t == (0, 0, 0) else f'{t[0]}x{t[1]}k{t[2]}'):>29s}" for t in tiles))
    for c, n in cases:
        if a.only and a.only not in c.name:
            continue
        cols = []
        for t in tiles:
            _lib.lib().rldm_debug_force_tile(t[0], t[1], t[2])
            try:
                us, kn = bench_case(c, iters=a.iters)
                tf = flops(c) / us / 1e6
                cols.append(f"{us:8.1f}us {tf:6.0f}TF {(kn.split('<')[1].rstrip('>') if '<' in kn else kn).replace(',taps', 't').replace('CK', 'c'):>14s}")
                tot[t] += us * n
                if a.ts:
                    buf = (C.c_ulonglong * 256)()
                    _lib.lib().rldm_debug_timestamps(buf)
                    if "conv_stream" in kn or "256,128,c64t9" in kn or True:
                        nb = 2048
                        bt = (C.c_ulonglong * (2 * nb))()
                        if _lib.lib().rldm_debug_block_times(bt, nb) == 0:
                            M48 = (1 << 48) - 1
                            st = [bt[2 * i] for i in range(nb) if bt[2 * i]]
                            en = [bt[2 * i + 1] & M48 for i in range(nb) if bt[2 * i]]
                            cu = [bt[2 * i + 1] >> 48 for i in range(nb) if bt[2 * i]]
                            if st:
                                t0 = min(st)
                                life = sorted((e - s_) * 0.01 for s_, e in zip(st, en))
                                print(f"    {len(st)} workgroups: starts spread {(max(st) - t0) * 0.01:.2f} us, ends {(min(en) - t0) * 0.01:.2f} .. "
                                      f"{(max(en) - t0) * 0.01:.2f} us after the first start; lifetime min / median / max "
                                      f"{life[0]:.2f} / {life[len(life) // 2]:.2f} / {life[-1]:.2f} us", file=sys.stderr)
                            if st:
                                # workgroups of one CU, in start order: how do co-resident ones overlap?
                                by_cu = collections.defaultdict(list)
                                for s_, e_, c_ in zip(st, en, cu):
                                    by_cu[c_].append(((s_ - t0) * 0.01, (e_ - t0) * 0.01))
                                k0 = sorted(by_cu)[0]
                                print(f"    {len(by_cu)} CUs seen; CU {k0:#x}: " + " ".join(f"[{a:.1f},{b:.1f}]" for a, b in sorted(by_cu[k0])[:12]),
                                      file=sys.stderr)
                    for blk in range(4):
                        v = [buf[blk * 64 + i] for i in range(64)]
                        v = [x for x in v if x]
                        print(f"    block {blk} stamps (cycles since start):", [int(x - v[0]) for x in v], file=sys.stderr)
            except RuntimeError as e:
                cols.append(f"{'fail':>29s}")
                print("   ", str(e)[:150], file=sys.stderr)
        print(f"{c.name:18s} {n:3d} {c.C0:4d}+{c.C1:<4d} {c.W:4d}x{c.H:<3d} {c.Cout:4d} {c.k} {c.stride} {c.up} {c.gn} {c.res:3d} | " +
              " | ".join(cols))
    print("sum over one forward (us): " + ", ".join(f"{('auto' if t == (0, 0, 0) else f'{t[0]}x{t[1]}k{t[2]}')}: {v:.0f}" for t, v in tot.items()))


if __name__ == "__main__":
    main()
