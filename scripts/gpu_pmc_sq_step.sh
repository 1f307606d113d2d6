#!/bin/bash
# SQ stall breakdown + LDS conflict counters per kernel over eager, in-line training steps ($1 = lrw | lrs).
mkdir -p gpurun_out/pmc_sq
export PYTHONUNBUFFERED=1 SVSR_SIDE_TRUNK=0 SVSR_SIDE_ENCODER=0
W=${1:-lrw}
cd /tmp && export TMPDIR=/tmp
timeout 800 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sq -o step_$W -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 1 > $GRAFT_REPO_ROOT/gpurun_out/pmc_sq/run_$W.log 2>&1
echo "exit $?"
cd $GRAFT_REPO_ROOT
python - $W <<'PY'
import csv, glob, collections, re, sys
f = glob.glob(f'gpurun_out/pmc_sq/step_{sys.argv[1]}_counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
with open(f) as fh:
    for row in csv.DictReader(fh):
        k = re.sub(r'\(.*', '', row['Kernel_Name'])[:46]
        agg[k][row['Counter_Name']] += float(row['Counter_Value']); cnt[(k, row['Counter_Name'])] += 1
for k in sorted(agg, key=lambda k: -agg[k].get('SQ_BUSY_CYCLES', 0))[:40]:
    v = agg[k]; wc = v.get('SQ_WAVE_CYCLES', 0) or 1
    print(f"{k:48s} busy {v.get('SQ_BUSY_CYCLES',0)/1e6:8.1f}M wait_any {v.get('SQ_WAIT_ANY',0)/wc:5.2f} wait_inst {v.get('SQ_WAIT_INST_ANY',0)/wc:5.2f} active {v.get('SQ_ACTIVE_INST_ANY',0)/wc:5.2f} "
          f"wait_lds {v.get('SQ_WAIT_INST_LDS',0)/wc:5.2f} lds_conf {v.get('SQ_LDS_BANK_CONFLICT',0)/(v.get('SQ_LDS_IDX_ACTIVE',0) or 1):5.2f} n={cnt[(k,'SQ_WAVE_CYCLES')]}")
PY
find gpurun_out/pmc_sq -name "*kernel_trace.csv" -delete
