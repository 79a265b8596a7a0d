"""Summarise the FETCH_SIZE / WRITE_SIZE rocprofv3 passes of tools/profile.sh pmc <tag> into
gpurun_out/pmc_<tag>_traffic.json (copy it to profiles/r4_pmc_traffic.json): HBM bytes per launch of every kernel,
FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B: MI355X_MICROARCH.md §HBM), WRITE_SIZE as is; both counters
are in KB.  The record carries the hash of the dominant kernel's comment-stripped source (tools/kernel_hash.py) so that bench.py can
refuse a stale file."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")


sys.path.insert(0, ROOT)
from tools.kernel_hash import DOMINANT_SOURCES, source_hash  # noqa: E402


def kernel_source_hash():
    return source_hash(DOMINANT_SOURCES)


def per_kernel(tag, counter):
    rows = {}
    for path in glob.glob(os.path.join(OUT, f"pmc_{tag}_{counter}", "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if r.get("Counter_Name") != counter:
                    continue
                a = rows.setdefault(r["Kernel_Name"], [0.0, set()])
                a[0] += float(r["Counter_Value"])
                a[1].add(r["Dispatch_Id"])
    return {k: (v[0], len(v[1])) for k, v in rows.items()}


def main():
    tag = sys.argv[1]
    fetch, write = per_kernel(tag, "FETCH_SIZE"), per_kernel(tag, "WRITE_SIZE")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py --steps 2 --warmup 1; KB; "
                     "FETCH_SIZE doubled per MI355X_MICROARCH.md, WRITE_SIZE as is", "kernel_source_sha256_16": kernel_source_hash(),
           "kernels": {}}
    for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, (0, 1))[0])):
        f, nf = fetch.get(k, (0.0, 0))
        w, nw = write.get(k, (0.0, 0))
        n = max(nf, nw, 1)
        out["kernels"][k[:100]] = {"launches": n, "fetch_kb_raw_per_launch": f / max(nf, 1), "write_kb_per_launch": w / max(nw, 1),
                                   "hbm_bytes_per_launch": (2.0 * f / max(nf, 1) + w / max(nw, 1)) * 1024.0}
    dom = [v for k, v in out["kernels"].items() if "ns_gemm_p8s_kernel" in k or "ns_gemm_p8_kernel" in k]   # persistent form + its one-tile fallback
    if dom:
        n = sum(v["launches"] for v in dom)
        out["kernel"] = "ns_gemm_p8_kernel+ns_gemm_p8s_kernel"
        out["launches"] = n
        out["hbm_bytes_per_launch"] = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in dom) / n
    path = os.path.join(OUT, f"pmc_{tag}_traffic.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(path, out.get("hbm_bytes_per_launch"))


if __name__ == "__main__":
    main()
