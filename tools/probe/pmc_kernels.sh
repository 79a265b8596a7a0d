#!/bin/bash
# one rocprofv3 --pmc pass (8 SQ counters) over a probe script; prints the mean counter values per launch and kernel
#   bash tools/probe/pmc_kernels.sh <tag> <script.py> [counters...]
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
tag="$1"; script="$2"; shift 2
C="${*:-SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT}"
OUT="$ROOT/gpurun_out/pmc_$tag"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT" -o p -- python3 "$ROOT/$script" > "$OUT.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); ids = defaultdict(set)
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); ids[k].add(r["Dispatch_Id"])
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0)):
    n = len(ids[k]); v = acc[k]
    print(k, "launches", n)
    print("   " + "  ".join(f"{c}={v[c]/n:.4g}" for c in sorted(v)))
PY
