# rocprofv3 kernel stats of soak_train.py variants (outlier hunt outside the bench configuration): bash tools/probe/prof_variants.sh "208 adalora" "273 lora" ...
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  set -- $v; ch=$1; kind=$2
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_var_${ch}_$kind -o k -- python3 $GRAFT_REPO_ROOT/tools/soak_train.py 40 $ch $kind > $GRAFT_REPO_ROOT/gpurun_out/var_${ch}_$kind.log 2>&1
  tail -1 $GRAFT_REPO_ROOT/gpurun_out/var_${ch}_$kind.log
done
