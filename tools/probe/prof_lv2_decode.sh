cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_lv2dec -o k -- python3 $GRAFT_REPO_ROOT/tools/probe/large_v2_decode.py > $GRAFT_REPO_ROOT/gpurun_out/lv2dec.log 2>&1
grep "large-v2" $GRAFT_REPO_ROOT/gpurun_out/lv2dec.log
head -16 $GRAFT_REPO_ROOT/gpurun_out/prof_lv2dec/k_kernel_stats.csv | cut -c1-140
