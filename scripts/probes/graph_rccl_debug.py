"""bisect: plain vs graph vs graph+RCCL losses on lrw_full_b2, with knobs"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch, torch.distributed as dist
from golden_cases import build_case
from syncvsr_amd import ops
from syncvsr_amd.engine import TrainStep
from syncvsr_amd.model import Model

dev = torch.device("cuda:0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
cfg, sd, batch, training, gold = build_case("lrw_full_b2")
cfg.optim.scheduler.num_warmup_steps = 1
gb = [t.to(dev) for t in batch]

def run(**kw):
    model = Model(cfg); model.load_state_dict(sd); model.to(dev).train()
    ts = TrainStep(model, cfg, **kw)
    losses, grads = [], []
    for _ in range(3):
        losses.append(ts.step(*gb)["loss_total"].clone())
        torch.cuda.synchronize()
        grads.append(model.store().grad.clone())
    run.offsets = model.store().offsets
    run.launched = list(ts.dp.launched) if ts.dp is not None else []
    return [float(l) for l in losses], model.store().flat.clone(), grads

def diff(a, b, tag):
    for s, (x, y) in enumerate(zip(a[2], b[2])):
        if not torch.equal(x, y):
            bad = []
            for name, (lo, n, _) in run.offsets.items():
                if not torch.equal(x[lo:lo + n], y[lo:lo + n]):
                    bad.append((name, lo, n, float((x[lo:lo+n]-y[lo:lo+n]).abs().max()), float(y[lo:lo+n].abs().max())))
            print(tag, "step", s, "grads differ in", len(bad), "params:", bad[:12])
            return
    print(tag, "grads equal")

BUCKET = 8.0
for arg in sys.argv[1:]:
    k, v = arg.split("=")
    if k == "group": ops.WGRAD_GROUP = bool(int(v))
    elif k == "bucket": BUCKET = float(v)
    else: ops.tune(**{k: int(v)})
plain = run()
print("plain     ", plain[0])
g = run(use_graph=True)
print("graph     ", g[0], bool(torch.equal(g[1], plain[1])))
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
e = run(always_reduce=True, bucket_mb=BUCKET)
print("eager+rccl", e[0], bool(torch.equal(e[1], plain[1])), "launched", run.launched)
g2 = run(always_reduce=True, bucket_mb=BUCKET, use_graph=True)
print("graph+rccl", g2[0], bool(torch.equal(g2[1], plain[1])))
diff(g2, plain, "graph+rccl")
if os.environ.get("SVSR_DBG_NO_AR") == "1":
    dist.all_reduce = lambda *a, **k: None
    g3 = run(always_reduce=True, bucket_mb=BUCKET, use_graph=True)
    print("graph+rccl(no all_reduce call)", g3[0], bool(torch.equal(g3[1], plain[1])))
dist.destroy_process_group()
