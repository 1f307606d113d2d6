#!/bin/bash
# SQ stall breakdown + LDS conflict counters for the conv forward kernels (op_bench fwd), one PMC pass.
mkdir -p gpurun_out/pmc_sq
export PYTHONUNBUFFERED=1 OPB_GRAPH=0
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sq -o sq -- python $GRAFT_REPO_ROOT/scripts/op_bench.py fwd wgrad > $GRAFT_REPO_ROOT/gpurun_out/pmc_sq/run.log 2>&1
echo "exit $?"
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections, re
for f in sorted(glob.glob('gpurun_out/pmc_sq/*counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = re.sub(r'\(.*', '', row['Kernel_Name'])[:48]
            agg[k][row['Counter_Name']] += float(row['Counter_Value']); cnt[(k, row['Counter_Name'])] += 1
    for k in sorted(agg):
        v = agg[k]
        wc = v.get('SQ_WAVE_CYCLES', 0) or 1
        print(f"{k:50s} wait_any {v.get('SQ_WAIT_ANY',0)/wc:5.2f} wait_inst {v.get('SQ_WAIT_INST_ANY',0)/wc:5.2f} active {v.get('SQ_ACTIVE_INST_ANY',0)/wc:5.2f} "
              f"wait_lds {v.get('SQ_WAIT_INST_LDS',0)/wc:5.2f} lds_conflict/active {v.get('SQ_LDS_BANK_CONFLICT',0)/(v.get('SQ_LDS_IDX_ACTIVE',0) or 1):5.2f} n={cnt[(k,'SQ_WAVE_CYCLES')]}")
PY
find gpurun_out/pmc_sq -name "*kernel_trace.csv" -delete
