#!/bin/bash
# kernel trace of the DEFAULT step (side stream on) -> timeline analysis of one steady-state step
mkdir -p gpurun_out/trace
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --profile-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/trace_run.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('gpurun_out/trace/t_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# find adamw launches to delimit steps
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('k_adamw')]
a, b = idx[-3], idx[-2]
step = rows[a + 1: b + 1]
t0 = int(step[0]['Start_Timestamp']); t1 = int(step[-1]['End_Timestamp'])
print("step wall us", (t1 - t0) / 1e3, "kernels", len(step))
ev = []
for r in step:
    ev.append((int(r['Start_Timestamp']), 1)); ev.append((int(r['End_Timestamp']), -1))
ev.sort()
cur = 0; last = t0; busy1 = busy2 = idle = 0
for t, d in ev:
    dt = t - last
    if cur == 0: idle += dt
    elif cur == 1: busy1 += dt
    else: busy2 += dt
    cur += d; last = t
print("idle us", idle / 1e3, "one kernel", busy1 / 1e3, ">=2 kernels", busy2 / 1e3)
# per-stream totals and the biggest gaps
by = collections.defaultdict(float)
for r in step: by[r['Stream_Id']] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print("per stream busy us", dict(by))
# phases: find first backward kernel (k_ce_bwd)
names = [r['Kernel_Name'][:40] for r in step]
def t_of(prefix, which=0):
    for r in (step if which == 0 else reversed(step)):
        if r['Kernel_Name'].startswith(prefix): return (int(r['Start_Timestamp']) - t0) / 1e3
print("t(ce_fwd)", t_of('k_ce_fwd'), "t(ce_bwd)", t_of('k_ce_bwd'), "t(stem wgrad)", t_of('void k_stem_conv_wgrad'), "t(sumsq)", t_of('k_grad_sumsq'))
PY
