"""Is the foreign resident kernel of svsr_debug_occupy_start really resident BESIDE work on the default stream? (tests/test_gpu_cotenant.py)"""
import sys, time, torch
sys.path.insert(0, '.')
from syncvsr_amd import _lib, ops
lib = _lib.load()
dev = torch.device("cuda:0")
side = torch.cuda.Stream()
x = torch.randn(4096, 4096, device=dev)
y = torch.empty_like(x)
for _ in range(3):
    torch.mm(x, x, out=y)
    y.add_(1.0)
torch.cuda.synchronize()
for wgs, lds in ((32, 96 * 1024), (256, 96 * 1024), (32, 1024)):
    ev = torch.cuda.Event()
    rc = lib.svsr_debug_occupy_start(wgs, lds, side.cuda_stream)
    ev.record(side)
    time.sleep(0.1)
    d0 = ev.query()
    t0 = time.perf_counter()
    y.add_(1.0)
    torch.cuda.synchronize(dev) if False else torch.cuda.current_stream().synchronize()
    t1 = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    torch.mm(x, x, out=y)
    torch.cuda.current_stream().synchronize()
    t2 = (time.perf_counter() - t0) * 1e3
    print(f"{wgs} x {lds >> 10} KiB: rc {rc}, done after 0.1 s {d0}; add_ beside {t1:.2f} ms, mm beside {t2:.2f} ms, occupy still running {not ev.query()}")
    lib.svsr_debug_occupy_stop()
    side.synchronize()
