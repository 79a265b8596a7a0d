cd $GRAFT_REPO_ROOT
python tools/bench_kernels.py "dec " > gpurun_out/sm_ab.log 2>&1
NS_EXTRA_HIPCC_FLAGS="-DNS_SM_MAXTILES=8192 -DNS_SM_MAXM=4096" python -c "
import os
from neuspeech1_amd import build as b
os.utime(os.path.join(b.CSRC,'ns_gemm_smallm.hip'))
b.build()"
echo "--- smallm for everything" >> gpurun_out/sm_ab.log
python tools/bench_kernels.py "dec " >> gpurun_out/sm_ab.log 2>&1
