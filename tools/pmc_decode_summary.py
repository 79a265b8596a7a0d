"""Summarise tools/profile.sh decode_pmc <tag> into gpurun_out/pmcd_<tag>_traffic.json (copy it to
profiles/r4_decode_pmc_traffic.json): HBM bytes per DECODE STEP, greedy and beam-5.

tools/bench_decode.py MARK=1 runs, after a warm-up, greedy at 64 and at 32 new tokens and beam-5 at 64 and 32, with a
marker launch (torch's cumsum) behind each generation; the dispatch list of each counter pass is cut at the markers.  The
encoder pass and the prompt are the same in the long and in the short generation, so (bytes(64) - bytes(32)) / 32 is the
traffic of one step at the mean cache length -- the same differencing bench.py's eval.roofline uses for the time.
FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B: MI355X_MICROARCH.md, HBM), WRITE_SIZE as is; both in KB."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
sys.path.insert(0, ROOT)
from tools.kernel_hash import DECODE_SOURCES, source_hash  # noqa: E402

NAMES = ["greedy_64", "greedy_32", "beam5_rep5_ngram2_64", "beam5_rep5_ngram2_32"]


def segments(tag, counter):
    rows = []
    for path in glob.glob(os.path.join(OUT, f"pmcd_{tag}_{counter}", "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if r.get("Counter_Name") == counter:
                    rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    segs, cur, by_kernel = [], 0.0, {}
    allk = []
    for _, name, v in rows:
        if "cumsum" in name.lower() or "scan" in name.lower():
            segs.append(cur)
            allk.append(by_kernel)
            cur, by_kernel = 0.0, {}
        else:
            cur += v
            a = by_kernel.setdefault(name[:80], [0.0, 0])
            a[0] += v
            a[1] += 1
    return segs, allk


def main():
    tag = sys.argv[1]
    f, fk = segments(tag, "FETCH_SIZE")
    w, wk = segments(tag, "WRITE_SIZE")
    # segment 0 = everything before the first marker (engine set-up + warm-up); then the four generations
    assert len(f) == 5 and len(w) == 5, (len(f), len(w))
    tot = {n: (2.0 * f[i + 1] + w[i + 1]) * 1024.0 for i, n in enumerate(NEW_NAMES)}
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of tools/bench_decode.py MARK=1 GRAPH=0 "
                     "(whisper-base 273-ch, B=128); KB; FETCH_SIZE doubled per MI355X_MICROARCH.md, WRITE_SIZE as is",
           "kernel_source_sha256_16": source_hash(DECODE_SOURCES), "hbm_bytes_per_generation": tot, "hbm_bytes_per_step": {},
           "per_kernel_hbm_bytes_per_step": {}}
    for mode in ("greedy", "beam5_rep5_ngram2"):
        out["hbm_bytes_per_step"][mode] = (tot[mode + "_64"] - tot[mode + "_32"]) / 32.0
        i64, i32 = NEW_NAMES.index(mode + "_64") + 1, NEW_NAMES.index(mode + "_32") + 1
        pk = {}
        for k in set(fk[i64]) | set(wk[i64]):
            b64 = 2.0 * fk[i64].get(k, [0, 0])[0] + wk[i64].get(k, [0, 0])[0]
            b32 = 2.0 * fk[i32].get(k, [0, 0])[0] + wk[i32].get(k, [0, 0])[0]
            n64, n32 = fk[i64].get(k, [0, 0])[1], fk[i32].get(k, [0, 0])[1]
            pk[k] = {"bytes_per_step": (b64 - b32) * 1024.0 / 32.0, "launches_per_step": (n64 - n32) / 32.0}
        out["per_kernel_hbm_bytes_per_step"][mode] = dict(sorted(pk.items(), key=lambda kv: -kv[1]["bytes_per_step"]))
    path = os.path.join(OUT, f"pmcd_{tag}_traffic.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(path, out["hbm_bytes_per_step"])


NEW_NAMES = NAMES
if __name__ == "__main__":
    main()
