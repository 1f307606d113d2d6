"""Child process of tests/test_gpu_syncdbg.py: one forward + backward of the word-level model at the benchmark batch and of the sentence-level
model on a short bucket, with whatever library SVSR_LIB_VARIANT selects; writes a bit-exact fingerprint of every output, every saved statistic
and every parameter gradient to the JSON file named on the command line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402


def fp(t: torch.Tensor) -> str:
    """order-sensitive 64-bit fingerprint of the tensor's BITS (two tensors with different bits collide with probability ~2^-60)"""
    x = t.detach().contiguous().view(-1)
    if x.element_size() == 2:
        x = x.view(torch.int16).to(torch.int64) & 0xFFFF
    elif x.element_size() == 4:
        x = x.view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    else:
        x = x.to(torch.int64)
    i = torch.arange(x.numel(), device=x.device, dtype=torch.int64)
    a = (x * ((i * 2654435761) % 1000003 + 1)).sum()
    b = (x * x + i % 8191).sum()
    return f"{int(a.item()) & 0xFFFFFFFFFFFFFFFF:016x}{int(b.item()) & 0xFFFFFFFFFFFFFFFF:016x}"


def main(out_path: str) -> None:
    from syncvsr_amd import _lib, ops
    from syncvsr_amd.config import default_lrw_config
    from syncvsr_amd.init import synthetic_batch
    from syncvsr_amd.lrs_init import LRS_ODIM, default_lrs_args, lrs_synthetic_batch
    from syncvsr_amd.lrs_model import E2E
    from syncvsr_amd.model import Model

    if os.environ.get("SVSR_LIB_VARIANT"):
        ops.tune("p8_stagger", 0)         # the two wave groups of the persistent kernel in lock step as well (its "split" barrier)
    dev = torch.device("cuda:0")
    res = {"library": os.path.basename(_lib.LIB_PATH)}
    # ---- word-level model, benchmark batch (persistent convolutions, fused encoder, fused audio head, unit-list weight gradients) ----
    cfg = default_lrw_config()
    cfg.train.batch_size = 32
    model = Model(cfg, seed=0).to(dev).train()
    model.reseed_dropout(99)
    batch = [t.to(dev) for t in synthetic_batch(cfg, 32, seed=4321)]
    out = model(*batch)
    out["loss_total"].backward()
    torch.cuda.synchronize()
    for k, v in out.items():
        res[f"lrw.out.{k}"] = fp(v)
    for k, p in model.named_parameters():
        res[f"lrw.grad.{k}"] = fp(p.grad)
    for k, b in model.named_buffers():
        res[f"lrw.buf.{k}"] = fp(b)
    del model, out
    torch.cuda.empty_cache()
    # ---- sentence-level model (4-wave dense layers with K split, streamed-key attention, Conformer convolution module, decoder, CTC) ----
    args = default_lrs_args(dropout_rate=0.1, transformer_attn_dropout_rate=0.1)
    m = E2E(LRS_ODIM, args, seed=0).to(dev).train()
    m.reseed_dropout(7)
    lb = [t.to(dev) for t in lrs_synthetic_batch(args, 4, 96, seed=11)]
    o = m(*lb)
    o[0].backward()
    torch.cuda.synchronize()
    for i, v in enumerate(o):
        res[f"lrs.out.{i}"] = fp(v)
    for k, p in m.named_parameters():
        if p.grad is not None:
            res[f"lrs.grad.{k}"] = fp(p.grad)
    json.dump(res, open(out_path, "w"))
    print(f"{len(res)} fingerprints from {res['library']}")


if __name__ == "__main__":
    main(sys.argv[1])
