#!/bin/bash
# per-kernel times of the LRS step (rocprofv3 --kernel-trace --stats): attention kernels
cd /tmp; export TMPDIR=/tmp PYTHONUNBUFFERED=1
rm -rf /tmp/pp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o l -- python $GRAFT_REPO_ROOT/bench.py --workload lrs --no-cpu-baseline --profile-steps 0 --steps 6 --warmup 2 --enqueue eager > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/pp/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot / 1e6)
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:60]:
    print(f"{r['Name'][:80]:80s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us tot {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r5d; cp $(find /tmp/pp -name "*kernel_stats.csv") $GRAFT_REPO_ROOT/gpurun_out/r5d/lrs_kernel_stats.csv
