#!/bin/bash
# One gpurun call = GPU tests + default bench + (optionally) profiles; logs under gpurun_out/<tag>_*.
# usage: bash tools/gpu_round.sh <tag> [tests|bench|stats|pmc|sq|decode|decode_pmc|loop20 ...]
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out"; mkdir -p "$OUT"
tag="$1"; shift
cd "$ROOT"
for what in "$@"; do
  case "$what" in
    tests)   timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/${tag}_tests.log" 2>&1; echo "tests rc=$?" >> "$OUT/${tag}_tests.log"; tail -5 "$OUT/${tag}_tests.log";;
    bench)   timeout 600 python bench.py > "$OUT/${tag}_bench.json" 2> "$OUT/${tag}_bench.err"; echo "bench rc=$?"; head -c 600 "$OUT/${tag}_bench.json";;
    benchq)  timeout 600 python bench.py --no-cpu-baseline --no-large-v2 > "$OUT/${tag}_bench.json" 2> "$OUT/${tag}_bench.err"; echo "bench rc=$?"; head -c 600 "$OUT/${tag}_bench.json";;
    trainq)  timeout 600 python bench.py --no-cpu-baseline --no-eval --steps 20 --warmup 5 > "$OUT/${tag}_train.json" 2> "$OUT/${tag}_train.err"; echo "train rc=$?"; head -c 400 "$OUT/${tag}_train.json";;
    stats|pmc|sq|decode|decode_pmc) bash tools/profile.sh "$what" "$tag";;
    loop20)  ok=0; for i in $(seq 1 20); do timeout 300 python -m pytest tests/test_cli_gpu.py -q -x -k test_evaluation_graph_replay_with_the_feed_thread_running > "$OUT/${tag}_loop_$i.log" 2>&1 && ok=$((ok+1)); done; echo "loop20: $ok / 20 green" | tee "$OUT/${tag}_loop20.txt";;
  esac
done
