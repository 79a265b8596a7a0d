#!/bin/bash
# SQ counter passes over the attention kernels alone (run on the GPU box): bash tools/probe/pmc_attn.sh <tag> [fwd|bwd]
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
OUT="$ROOT/gpurun_out"; tag="$1"; what="${2:-fwd}"
cd /tmp && export TMPDIR=/tmp
export WHAT=$what N=4
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$OUT/pmca_${tag}_1" -o p -- python3 "$ROOT/tools/probe/attn_only.py" > "$OUT/pmca_${tag}_1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM --output-format csv -d "$OUT/pmca_${tag}_2" -o p -- python3 "$ROOT/tools/probe/attn_only.py" > "$OUT/pmca_${tag}_2.log" 2>&1
python3 - <<PY
import csv, glob, collections
for p in (1, 2):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for path in glob.glob("$OUT/pmca_${tag}_%d/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(path)):
            if "attn" not in r["Kernel_Name"]: continue
            a = acc[(r["Kernel_Name"][:60], r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for (k, c), (v, n) in sorted(acc.items()):
        print(f"{k:60s} {c:28s} {v / max(n, 1):14.4g} per launch ({n} launches)")
PY
