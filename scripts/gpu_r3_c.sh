#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3c
rm -rf $OUT; mkdir -p $OUT
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|TA_[A-Z0-9_]*\|TCC_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > $OUT/counters.txt
cat > /tmp/p8_one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from syncvsr_amd import ops
dev = torch.device("cuda:0"); BF16 = torch.bfloat16
x3 = (torch.randn(928, 6, 6, 256, device=dev) * 0.5).to(BF16); w3 = (torch.randn(256, 3, 3, 256, device=dev) * 0.05).to(BF16)
x2 = (torch.randn(928, 11, 11, 128, device=dev) * 0.5).to(BF16); w2 = (torch.randn(128, 3, 3, 128, device=dev) * 0.05).to(BF16)
for mode in (0, 1):
    ops.tune("p8", mode)
    for _ in range(5):
        ops.conv2d_fwd(x3, w3, 3, 1, 1, want_stats=True)
        ops.conv2d_fwd(x2, w2, 3, 1, 1, want_stats=True)
torch.cuda.synchronize()
PY
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM"
P3="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT -o pmc$i -- python /tmp/p8_one.py > $OUT/pmc$i.log 2>&1; echo "pmc$i $?"
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, os, re, collections
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r3c")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
        if "igemm" not in k: continue
        k += " grid=" + row.get("Grid_Size", row.get("Grid_Size_X", "?"))
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
lines = []
for k, v in agg.items():
    lines.append(k)
    for c, vals in sorted(v.items()):
        lines.append(f"   {c:32s} {sum(vals)/len(vals):16.0f}  (n={len(vals)})")
open(os.path.join(out, "pmc_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
find $OUT -name "*.csv" -delete; find $OUT -name "*.db" -delete
