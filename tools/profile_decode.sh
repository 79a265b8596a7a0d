cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
BEAMS=5 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dec5 -o k -- python3 $R/tools/bench_decode.py > $R/gpurun_out/dec5.log 2>&1
BEAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dec1 -o k -- python3 $R/tools/bench_decode.py > $R/gpurun_out/dec1.log 2>&1
