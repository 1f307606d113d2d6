#!/usr/bin/env python
"""Copies the summaries scripts/gpu_profiles.sh left under gpurun_out/roundN/ into profiles/roundN_* (N = $ROUND, default 6), stamped with the
commit they were measured at (run right after the gpurun call, with a clean work tree):   python scripts/collect_profiles.py"""
import json
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RN = "round" + os.environ.get("ROUND", "6")
SRC, DST = os.path.join(ROOT, "gpurun_out", RN), os.path.join(ROOT, "profiles")
head = subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
dirty = bool(subprocess.run(["git", "status", "--porcelain", "--", "syncvsr_amd", "bench.py", "include"], cwd=ROOT, capture_output=True, text=True).stdout.strip())
stamp = f"{head}{' + uncommitted changes' if dirty else ''}"
note = (f"# measured at commit {stamp} on one MI355X (gfx950) by scripts/gpu_profiles.sh: rocprofv3, every launch in line (SVSR_SIDE_*=0)\n")
for src, dst in (("lrw_kernel_stats.csv", RN + "_lrw_kernel_stats.csv"), ("lrs_kernel_stats.csv", RN + "_lrs_kernel_stats.csv")):
    with open(os.path.join(SRC, src)) as f, open(os.path.join(DST, dst), "w") as g:
        g.write(note + f.read())
with open(os.path.join(SRC, "sq_stall_breakdown.txt")) as f, open(os.path.join(DST, RN + "_sq_stall_breakdown.txt"), "w") as g:
    g.write(note + "# rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES;\n"
            "# fractions of SQ_WAVE_CYCLES: wait_any = parked on s_waitcnt/barrier, wait_inst = issue stall, active = issuing; lds_conf = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE\n" + f.read())
META = {"commit": stamp, "tool": "rocprofv3 --pmc (separate passes: FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES) --kernel-trace",
        "units": "FETCH_SIZE / WRITE_SIZE in KiB per dispatch; on gfx950 FETCH_SIZE counts 64 B per 128-B request (MI355X_MICROARCH.md §HBM): double it"}
if os.path.exists(os.path.join(SRC, "pmc_lrs_per_kernel.json")):
    lrs = json.load(open(os.path.join(SRC, "pmc_lrs_per_kernel.json")))
    lrs["__meta__"] = dict(META, workload="LRS step (B = 16, one 160-frame bucket)")
    json.dump(lrs, open(os.path.join(DST, RN + "_pmc_lrs_per_kernel.json"), "w"), indent=1, sort_keys=True)
rec = json.load(open(os.path.join(SRC, "pmc_per_kernel.json")))
rec["__meta__"] = {"commit": stamp, "tool": "rocprofv3 --pmc (separate passes: FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES) --kernel-trace",
                   "units": "FETCH_SIZE / WRITE_SIZE in KiB per dispatch; on gfx950 FETCH_SIZE counts 64 B per 128-B request (MI355X_MICROARCH.md §HBM): double it"}
json.dump(rec, open(os.path.join(DST, RN + "_pmc_per_kernel.json"), "w"), indent=1, sort_keys=True)
line = open(os.path.join(SRC, "bench_lrw.json")).read().strip().splitlines()[-1]
bench_line = json.loads(line)
# the bench run on the GPU box read the PMC file committed BEFORE this profiling call: point roofline.traffic at the passes of this one
import sys
sys.path.insert(0, ROOT)
import bench as _bench
def _refresh(roof, lrs):
    """traffic and frac_rocprof of a roofline object from the passes of THIS call (the run on the GPU box read the previous round's files)"""
    roof["traffic"] = _bench.pmc_traffic(roof["kernel"], lrs=lrs)
    rp = _bench.rocprof_avg_us(roof["kernel"], lrs=lrs)
    if rp is not None:
        flops_per_launch = roof["achieved"] * 1e12 * roof["avg_launch_us"] * 1e-6
        roof["frac_rocprof"] = round(flops_per_launch / (rp[0] * 1e-6) / _bench.MFMA_PEAK_BF16, 5)
        roof["rocprof_avg_launch_us"] = round(rp[0], 2)
        roof["rocprof_source"] = rp[1]


if "roofline" in bench_line:
    _refresh(bench_line["roofline"], False)
if "roofline" in bench_line.get("lrs", {}):
    _refresh(bench_line["lrs"]["roofline"], True)
open(os.path.join(DST, RN + "_bench_line.json"), "w").write(json.dumps(bench_line) + "\n")
open(os.path.join(DST, RN + "_COMMIT"), "w").write(stamp + "\n")
print(f"profiles/{RN}_* written for", stamp)
