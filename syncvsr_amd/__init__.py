"""SyncVSR training hot path for MI355X (gfx950): hand-written HIP kernels behind a C ABI + the host mirror of the reference modules."""
import os as _os

# HIP maps streams onto a small pool of hardware queues (4 by default).  A data-parallel step uses four streams at once — compute,
# the weight-gradient side stream (model._SideStream), the gradient all-reduce stream (engine.GradReducer) and RCCL's own — and with
# 4 queues two of them share one: the side stream then serialises behind the collective stream and the step is SLOWER than with no
# side stream at all (measured, 1 rank + RCCL: 7.77 ms; without the side stream 7.43; with 8 queues 6.96; no process group 6.67).
# Must be in the environment before the HIP runtime initialises, i.e. before the first torch.cuda call of the process.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
