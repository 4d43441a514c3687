import os, sys, torch
sys.path.insert(0, '/root/repo')
os.chdir('/root/repo')
from rangeldm_amd import training as TR, distributed as D, train_ops as T
from rangeldm_amd.config import UNetConfig
from rangeldm_amd.params import unet_param_shapes
from rangeldm_amd.synth import synth_state_dict
SMALL = dict(sample_size=(32, 8), block_out_channels=(32, 32, 64, 64))
cfg = UNetConfig(**SMALL)
sd = synth_state_dict(unet_param_shapes(cfg), prefix="tr.")
x = torch.randn(2, 5, 32, 8, generator=torch.Generator().manual_seed(1)).cuda()
target = torch.randn(2, 4, 32, 8, generator=torch.Generator().manual_seed(2)).cuda()
t = torch.tensor([5, 900]).cuda()
def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm())
a = TR.UNetTrainer(cfg, sd, use_ema=False, bucket_mb=1)
pa = a.forward(x, t)
pred_ref = pa.clone()
a.backward(T.mse(pa, target)[1], reduce=False)
ga = a.grads.clone()
# run-to-run noise of the single-process path
for i in range(5):
    a.grads.zero_()
    a.backward(T.mse(a.forward(x, t), target)[1], reduce=False)
    print("single-process repeat", i, rel(a.grads, ga))
mode = sys.argv[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29633", RANK="0", WORLD_SIZE="1")
if mode == "none":
    pass
elif mode == "cabi":
    os.environ["RLDM_COLLECTIVE"] = "cabi"
    torch.distributed.init_process_group("gloo", rank=0, world_size=1)
else:
    torch.distributed.init_process_group("nccl", rank=0, world_size=1)
b = TR.UNetTrainer(cfg, sd, use_ema=False, bucket_mb=1)
for i in range(8 if mode != "none" else 0):
    b.grads.zero_()
    b.backward(T.mse(b.forward(x, t), target)[1], reduce=True)
    torch.cuda.synchronize()
    print(mode, "reduce path", i, rel(b.grads, ga))
# fresh trainers: the first reduce of a trainer (lazy communicator, cold kernels)
worst = 0.0
for i in range(300):
    if mode == "cabi" and i % 5 == 0:
        D._COMM = None            # a new communicator every few rounds, as the test suite does
    c = TR.UNetTrainer(cfg, sd, use_ema=False, bucket_mb=1)
    pc = c.forward(x, t)
    pred_c = pc.clone()
    c.backward(T.mse(pc, target)[1], reduce=(mode != "none"))
    torch.cuda.synchronize()
    r = rel(c.grads, ga)
    worst = max(worst, r)
    if r > 1e-6:
        print(mode, "fresh trainer", i, "rel", r, "forward output rel diff", rel(pred_c, pred_ref), "equal params", bool(torch.equal(c.params, a.params)),
              "wf equal", all(torch.equal(c.wf[k], a.wf[k]) for k in a.wf), "wt equal", all((c.wt[k] is None) or torch.equal(c.wt[k], a.wt[k]) for k in a.wt))
        bad = []
        for n in c.names:
            o, k = c.offsets[n], c.sizes[n]
            d = float((c.grads[o:o + k].double() - ga[o:o + k].double()).norm())
            if d > 0:
                bad.append((d, d / (float(ga[o:o + k].double().norm()) + 1e-30), n, o, k))
        bad.sort(reverse=True)
        print("   worst parameters (abs diff, rel, name, offset, size):", [(float("%.3e" % a_), round(r_, 5), n_, o_, k_) for a_, r_, n_, o_, k_ in bad[:5]], "|grad|", float(ga.double().norm()))
print(mode, "fresh trainers: worst rel", worst)
