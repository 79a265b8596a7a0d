# same-box bench A/B of library builds: bash tools/probe/bench_ab.sh <name> ...   (tools/probe/build/lib_<name>.so; "" = the in-tree library)
for i in 1 2; do for v in "" $@; do
  if [ -z "$v" ]; then echo -n "default "; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-eval --no-large-v2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])";
  else echo -n "$v "; NS_LIB_PATH=tools/probe/build/lib_$v.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-eval --no-large-v2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; fi
done; done
