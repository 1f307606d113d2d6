import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from golden_cases import build_lrs_case
from oracle import lrs_oracle as O
torch.set_num_threads(8)
args, odim, sd, batch, training, gold = build_lrs_case("lrs_full_b2")
x, lengths, tokens, label = batch
def run(autocast):
    osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd.items()}
    if autocast:
        with torch.autocast("cpu", dtype=torch.bfloat16):
            out = O.forward(osd, args, x, lengths, tokens, label, training=True)
    else:
        out = O.forward(osd, args, x, lengths, tokens, label, training=True)
    out["loss"].backward()
    return out, osd
o32, s32 = run(False)
o16, s16 = run(True)
print("loss fp32", float(o32["loss"]), "autocast", float(o16["loss"]), "rel", abs(float(o16["loss"])-float(o32["loss"]))/float(o32["loss"]))
for k in ("loss_ctc","loss_att","loss_audio"):
    print(k, float(o32[k]), float(o16[k]))
cos=[]
for n in s32:
    if s32[n].is_floating_point() and s32[n].grad is not None:
        a, b = s16[n].grad.float().flatten(), s32[n].grad.flatten()
        if b.norm() > 1e-6:
            cos.append((float(torch.dot(a,b)/(a.norm()*b.norm()+1e-30)), n))
cos.sort()
print("min", cos[0], "median", cos[len(cos)//2][0], "n", len(cos))
print(cos[:5])
