#!/bin/bash
# Everything profiles/roundN_* is made from (N = $ROUND, default 6), in one gpurun call (every pass is its own rocprofv3 run; PMC passes carry --kernel-trace only).
# All launches in line (no side-stream overlap), like bench.py's roofline leg.  Output: gpurun_out/round$ROUND/;
# scripts/collect_profiles.py copies the summaries into profiles/ and stamps the commit.
ROUND=${ROUND:-6}
OUT=$GRAFT_REPO_ROOT/gpurun_out/round$ROUND
rm -rf $OUT; mkdir -p $OUT
export ROUND PYTHONUNBUFFERED=1 SVSR_SIDE_TRUNK=0 SVSR_SIDE_ENCODER=0
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-lrs-leg --sustained-steps 0 --enqueue eager --profile-steps 1"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o lrw -- $B --steps 5 --warmup 2 > $OUT/lrw_run.log 2>&1; echo "lrw stats $?"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o lrs -- $B --workload lrs --steps 3 --warmup 2 > $OUT/lrs_run.log 2>&1; echo "lrs stats $?"
for ctr in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  tag=pmc_$(echo $ctr | tr ' ' '_')
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT -o $tag -- $B --steps 2 --warmup 1 > $OUT/$tag.log 2>&1; echo "$tag $?"
done
for ctr in FETCH_SIZE WRITE_SIZE; do          # the sentence-level step's traffic (its dominant kernel differs)
  tag=lrspmc_$ctr
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT -o $tag -- $B --workload lrs --steps 1 --warmup 1 > $OUT/$tag.log 2>&1; echo "$tag $?"
done
timeout 800 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT -o sq_lrw -- $B --steps 2 --warmup 1 > $OUT/sq_lrw.log 2>&1; echo "sq $?"
cd $GRAFT_REPO_ROOT
# the side-stream (default) step for the headline number, same box
unset SVSR_SIDE_TRUNK SVSR_SIDE_ENCODER
python bench.py --steps 50 --warmup 5 > $OUT/bench_lrw.json 2> $OUT/bench_lrw.err; tail -c 300 $OUT/bench_lrw.json
# reduce the PMC csvs to per-kernel averages here (the raw files are large)
python - <<'PY'
import csv, glob, collections, json, os, re
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "round" + os.environ.get("ROUND", "6"))
rec = collections.defaultdict(dict)
for f in sorted(glob.glob(os.path.join(out, "pmc_*counter_collection.csv"))):
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = (re.sub(r"\(.*", "", row["Kernel_Name"]).strip(), row["Counter_Name"])
        agg[k] += float(row["Counter_Value"]); cnt[k] += 1
    for (k, c), v in agg.items():
        rec[k][f"{c}_avg_per_dispatch"] = v / cnt[(k, c)]
        rec[k][f"dispatches_{c}"] = cnt[(k, c)]
json.dump(rec, open(os.path.join(out, "pmc_per_kernel.json"), "w"), indent=1, sort_keys=True)
rec = collections.defaultdict(dict)
for f in sorted(glob.glob(os.path.join(out, "lrspmc_*counter_collection.csv"))):
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = (re.sub(r"\(.*", "", row["Kernel_Name"]).strip(), row["Counter_Name"])
        agg[k] += float(row["Counter_Value"]); cnt[k] += 1
    for (k, c), v in agg.items():
        rec[k][f"{c}_avg_per_dispatch"] = v / cnt[(k, c)]
        rec[k][f"dispatches_{c}"] = cnt[(k, c)]
json.dump(rec, open(os.path.join(out, "pmc_lrs_per_kernel.json"), "w"), indent=1, sort_keys=True)
lines = []
for f in glob.glob(os.path.join(out, "sq_lrw_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", row["Kernel_Name"])[:46]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k, row["Counter_Name"]] += 1
    for k in sorted(agg, key=lambda k: -agg[k].get("SQ_BUSY_CYCLES", 0))[:40]:
        v = agg[k]; wc = v.get("SQ_WAVE_CYCLES", 0) or 1
        lines.append(f"{k:48s} busy {v.get('SQ_BUSY_CYCLES',0)/1e6:8.1f}M wait_any {v.get('SQ_WAIT_ANY',0)/wc:5.2f} wait_inst {v.get('SQ_WAIT_INST_ANY',0)/wc:5.2f} "
                     f"active {v.get('SQ_ACTIVE_INST_ANY',0)/wc:5.2f} wait_lds {v.get('SQ_WAIT_INST_LDS',0)/wc:5.2f} "
                     f"lds_conf {v.get('SQ_LDS_BANK_CONFLICT',0)/(v.get('SQ_LDS_IDX_ACTIVE',0) or 1):5.2f} dispatches={cnt[k,'SQ_WAVE_CYCLES']}")
open(os.path.join(out, "sq_stall_breakdown.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:12]))
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
ls -la $OUT | head -30
