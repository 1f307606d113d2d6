#!/bin/bash
# PMC passes (separate runs, kernel-trace only) over one eager training step: HBM bytes and MFMA busy per kernel.
mkdir -p gpurun_out/pmc
export PYTHONUNBUFFERED=1
# per-kernel numbers are taken with every launch in line (no side-stream overlap), like bench.py's roofline leg
export SVSR_SIDE_TRUNK=0 SVSR_SIDE_ENCODER=0
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" ; do
  tag=$(echo $ctr | tr ' ' '_')
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc -o $tag -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --profile-steps 1 > $GRAFT_REPO_ROOT/gpurun_out/pmc/$tag.log 2>&1
  echo "$tag exit $?"
done
cd $GRAFT_REPO_ROOT
ls -la gpurun_out/pmc | head -20
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc/*counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row['Kernel_Name'][:60]
            agg[k][row['Counter_Name']] += float(row['Counter_Value']); cnt[(k, row['Counter_Name'])] += 1
    print('==', f)
    for k in sorted(agg, key=lambda k: -max(agg[k].values()))[:14]:
        print('  ', k, {c: (round(v / cnt[(k, c)], 1), cnt[(k, c)]) for c, v in agg[k].items()})
PY
find gpurun_out/pmc -name "*kernel_trace.csv" -delete
