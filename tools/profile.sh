#!/bin/bash
# rocprofv3 evidence for profiles/ (run ON the GPU box through gpurun):
#   bash tools/profile.sh stats <tag> [bench args]   kernel-trace + stats of bench.py            -> gpurun_out/prof_<tag>/
#   bash tools/profile.sh pmc   <tag>                 FETCH_SIZE / WRITE_SIZE in SEPARATE passes  -> gpurun_out/pmc_<tag>_*/
#   bash tools/profile.sh sq    <tag>                 four SQ counters (MFMA busy, LDS) in one pass       -> gpurun_out/pmc_<tag>_SQ/
#   bash tools/profile.sh decode <tag>                kernel stats of tools/bench_decode.py
#   bash tools/profile.sh decode_pmc <tag>            FETCH_SIZE / WRITE_SIZE per decode step -> gpurun_out/pmcd_<tag>_traffic.json
# Counters are collected in runs of their own (never together with --stats / trace domains other than kernel-trace).
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
mode="$1"; tag="$2"; shift 2
cd /tmp && export TMPDIR=/tmp
# every pass runs the step EAGERLY (NS_TRAIN_GRAPH=0: the same kernels with the same arguments, one dispatch record per launch --
# rocprofv3's kernel trace on this ROCm does not list the kernels of a replayed hipGraph); bench.py's timed loop replays the graph
case "$mode" in
  stats)
    NS_TRAIN_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$tag" -o k -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline "$@" > "$OUT/bench_$tag.json" 2> "$OUT/prof_$tag.log"
    ;;
  pmc)
    for c in FETCH_SIZE WRITE_SIZE; do
      NS_TRAIN_GRAPH=0 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_${tag}_$c" -o p -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-eval > "$OUT/pmc_${tag}_$c.json" 2> "$OUT/pmc_${tag}_$c.log"
    done
    python3 "$ROOT/tools/pmc_summary.py" "$tag"
    ;;
  sq)
    NS_TRAIN_GRAPH=0 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/pmc_${tag}_SQ" -o p -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-eval > "$OUT/pmc_${tag}_SQ.json" 2> "$OUT/pmc_${tag}_SQ.log"
    python3 "$ROOT/tools/pmc_sq_summary.py" "$tag"
    ;;
  decode_pmc)
    # FETCH_SIZE / WRITE_SIZE of ONE decode step: eager launches (GRAPH=0: a replayed hipGraph's kernels are not listed), greedy and beam-5 at
    # 64 and 32 new tokens in one process, a marker launch between the four generations (tools/bench_decode.py MARK=1)
    for c in FETCH_SIZE WRITE_SIZE; do
      GRAPH=0 MARK=1 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmcd_${tag}_$c" -o p -- python3 "$ROOT/tools/bench_decode.py" > "$OUT/pmcd_${tag}_$c.log" 2>&1
    done
    python3 "$ROOT/tools/pmc_decode_summary.py" "$tag"
    ;;
  decode)
    # (GRAPH=0: launch lists all the way -- a session's second call would capture hipGraphs, whose kernels the trace does not list)
    GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$tag" -o k -- python3 "$ROOT/tools/bench_decode.py" "$@" > "$OUT/decode_$tag.log" 2> "$OUT/prof_$tag.log"
    ;;
esac
