import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from golden_cases import build_case
from syncvsr_amd.engine import TrainStep
from syncvsr_amd.model import Model
dev = torch.device("cuda:0")
cfg, sd, batch, training, gold = build_case("lrw_full_b2")
cfg.optim.scheduler.num_warmup_steps = 1
cfg.model.bert.hidden_dropout_prob = 0.1
cfg.model.bert.attention_probs_dropout_prob = 0.1
gb = [t.to(dev) for t in batch]
def run(native, sync=False):
    model = Model(cfg, seed=3); model.load_state_dict(sd); model.to(dev).train()
    ts = TrainStep(model, cfg, native=native)
    outs = []
    for _ in range(4):
        o = ts.step(*gb)
        if sync: torch.cuda.synchronize()
        outs.append({k: v.clone() for k, v in o.items()})
    torch.cuda.synchronize()
    st = model.store()
    return outs, st.flat.clone(), st.grad.clone()
for label, kw in (("eager", dict(native=False)), ("eager2", dict(native=False)), ("native", dict(native=True)), ("native+sync", dict(native=True, sync=True))):
    res = run(**kw)
    if label == "eager":
        base = res
    diffs = []
    for i, (a, b) in enumerate(zip(base[0], res[0])):
        for k in a:
            if not torch.equal(a[k], b[k]): diffs.append((i, k, a[k].item(), b[k].item()))
    print(label, "side", os.environ.get("SVSR_SIDE_TRUNK", "1"), "diffs:", diffs[:6], "params equal", torch.equal(base[1], res[1]), "grad diff elems", int((base[2] != res[2]).sum()))
import numpy as np
model = Model(cfg, seed=3); model.load_state_dict(sd); model.to(dev).train()
o = model(*gb)
lc, la, lt = o["loss_category"].item(), o["loss_audio"].item(), o["loss_total"].item()
lam = model.lambda_audio
f = np.float32
print("lambda", lam, "lc", repr(lc), "la", repr(la), "torch total", repr(lt))
print("two roundings", repr(float(f(lc) + f(f(la) * f(lam)))), "fma", repr(float(f(np.float64(lc) + np.float64(la) * np.float64(f(lam))))), "double lam", repr(float(f(np.float64(lc) + np.float64(f(np.float64(la) * lam))))))
from syncvsr_amd import ops
print("lincomb2", repr(ops.lincomb2(o["loss_category"].detach(), o["loss_audio"].detach(), lam).item()))
for native in (False, True):
    model = Model(cfg, seed=3); model.load_state_dict(sd); model.to(dev).train()
    ts = TrainStep(model, cfg, native=native)
    for i in range(3):
        o = ts.step(*gb)
        lc, la, lt = o["loss_category"].item(), o["loss_audio"].item(), o["loss_total"].item()
        print("native" if native else "eager ", i, "lc", repr(lc), "la", repr(la), "total", repr(lt), "| two roundings", repr(float(f(lc) + f(f(la) * f(10.0)))),
              "fma", repr(float(f(np.float64(lc) + np.float64(la) * 10.0))))
