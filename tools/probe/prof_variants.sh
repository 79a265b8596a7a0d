cd /tmp; export TMPDIR=/tmp
for kind in adalora full; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r3r_$kind -o k -- python3 $GRAFT_REPO_ROOT/tools/soak_train.py 40 208 $kind > $GRAFT_REPO_ROOT/gpurun_out/r3r_$kind.log 2>&1
  tail -2 $GRAFT_REPO_ROOT/gpurun_out/r3r_$kind.log
done
