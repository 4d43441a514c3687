import os, sys, torch
sys.path.insert(0, "/root/repo")
from rangeldm_amd import train_ops as T
for shape in [(8, 256, 16, 128), (8, 256, 16, 256), (8, 128, 8, 256)]:
    x = torch.randn(*shape, device="cuda"); dy = torch.randn(*shape, device="cuda")
    C = shape[3]; g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
    dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    y, st = T.gn_forward(x, g, b, 32, 1e-5, True)
    dx = T.gn_backward(x, dy, st, g, b, 32, True, dg, db)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    for _ in range(20): y, st = T.gn_forward(x, g, b, 32, 1e-5, True)
    e[1].record()
    for _ in range(20): T.gn_backward(x, dy, st, g, b, 32, True, dg, db, dx=dx)
    e[2].record(); torch.cuda.synchronize()
    print(shape, f"fwd {e[0].elapsed_time(e[1])/20*1e3:.1f} us, bwd {e[1].elapsed_time(e[2])/20*1e3:.1f} us")
