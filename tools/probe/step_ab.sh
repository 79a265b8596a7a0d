#!/bin/bash
# Same-box A/B of the whole training step: the tree's library against tools/probe/build/libns_prev.so (NS_LIB_PATH), two interleaved rounds.
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"; cd "$ROOT"
for r in 1 2; do
  for v in prev new; do
    if [ $v = prev ]; then export NS_LIB_PATH="$ROOT/tools/probe/build/libns_prev.so"; else unset NS_LIB_PATH; fi
    python bench.py --no-cpu-baseline --no-eval --no-roofline --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v round $r', d['ms_per_step'], 'ms/step', d['value'], 'samples/s')"
  done
done
