"""Summarise the SQ counter pass of tools/profile.sh sq <tag> into gpurun_out/pmc_<tag>_SQ_by_kernel.tsv: per kernel (and grid
size, so that the encoder's and the decoder's launches of one kernel stay apart) the mean counter values per launch and
    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (32 * SQ_BUSY_CYCLES)
(the SQ counters are summed over the 32 shader engines' SQs: SQ_BUSY_CYCLES / 32 is the launch's length in cycles, and
SQ_VALU_MFMA_BUSY_CYCLES counts per SIMD, 1024 of them: the ratio is the share of the matrix pipes' cycles that had an MFMA in
flight), lds_active likewise = SQ_ACTIVE_INST_LDS / (32 * SQ_BUSY_CYCLES) per SIMD-equivalent."""
import csv
import glob
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
COUNTERS = ["SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT"]


def main():
    tag = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(float))
    ids = defaultdict(set)
    for path in glob.glob(os.path.join(OUT, f"pmc_{tag}_SQ", "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                key = (r["Kernel_Name"][:90], r.get("Grid_Size", ""))
                acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
                ids[key].add(r["Dispatch_Id"])
    path = os.path.join(OUT, f"pmc_{tag}_SQ_by_kernel.tsv")
    with open(path, "w") as f:
        f.write("kernel\tgrid\tlaunches\t" + "\t".join(c + "_per_launch" for c in COUNTERS) + "\tmfma_busy\tlds_active\n")
        for key in sorted(acc, key=lambda k: -acc[k].get("SQ_BUSY_CYCLES", 0.0)):
            n = max(len(ids[key]), 1)
            v = acc[key]
            busy = v.get("SQ_BUSY_CYCLES", 0.0)
            mf = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (32.0 * busy) if busy else 0.0
            ld = v.get("SQ_ACTIVE_INST_LDS", 0.0) / (32.0 * busy) if busy else 0.0
            f.write(f"{key[0]}\t{key[1]}\t{n}\t" + "\t".join(f"{v.get(c, 0.0) / n:.4g}" for c in COUNTERS) + f"\t{mf:.3f}\t{ld:.3f}\n")
    print(path)


if __name__ == "__main__":
    main()
