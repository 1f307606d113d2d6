"""Run-to-run reproducibility of the LRW gradient (in-line launches) at several batch sizes: the tiny-batch parity case is
chaotic (BatchNorm over a few hundred samples amplifies bf16 rounding flips that follow from fp32 atomic-add order)."""
import sys, torch
sys.path.insert(0, '.')
from syncvsr_amd.config import default_lrw_config
from syncvsr_amd.init import synthetic_batch
from syncvsr_amd.model import Model
dev = torch.device('cuda:0')
cfg = default_lrw_config()
model = Model(cfg, seed=0).to(dev).train()
def cos(a, b): return float(torch.dot(a, b) / (a.norm() * b.norm()))
for B in (2, 8, 32):
    gb = [t.to(dev) for t in synthetic_batch(cfg, B, seed=5)]
    gs = []
    for side in (False, False, False, True, True):
        model._side.enabled = side
        model(*gb)["loss_total"].backward(); torch.cuda.synchronize()
        gs.append(model.store().grad.clone())
    print(f"B={B}: inline/inline {cos(gs[1], gs[2]):.6f}  side/inline {cos(gs[3], gs[2]):.6f}  side/side {cos(gs[3], gs[4]):.6f}")

# per-tensor run-to-run cosine at B=32 (in-line), sorted by contribution to the squared gradient norm
gb = [t.to(dev) for t in synthetic_batch(cfg, 32, seed=5)]
model._side.enabled = False
gs = []
for _ in range(2):
    model(*gb)["loss_total"].backward(); torch.cuda.synchronize()
    gs.append(model.store().grad.clone())
st = model.store()
rows = []
tot = float(gs[0].norm() ** 2)
for n, (o, numel, shape) in st.offsets.items():
    a, b = gs[0][o:o + numel], gs[1][o:o + numel]
    rows.append((float(a.norm() ** 2) / tot, cos(a, b), n))
rows.sort(reverse=True)
for r in rows[:14]: print("share %.4f cos %.6f %s" % r)
print("worst cos among share>1e-4:", sorted((c, n) for s_, c, n in rows if s_ > 1e-4)[:8])
print("encoder / head tensors:")
for s_, c, n in rows:
    if n.startswith(("encoder.encoder.layer.5", "encoder.encoder.layer.0.attention.self.query.weight", "audio_projection", "cls_token", "encoder.embeddings")):
        print("  cos %.6f %s" % (c, n))
