#!/bin/bash
# PMC counters of the 8-phase GEMM kernels (separate passes, 4 counters each):  gpurun -- 'bash tools/pmc_gemm8.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_gemm8; mkdir -p $O
cat > /tmp/g8.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
from ops_binding import P, lib, ok, stream
L = lib()
for (M, N, K, flags) in [(16384, 1280, 5120, 932 << 8), (16384, 3840, 1280, 932 << 8), (16384, 10240, 1280, (825 << 8) | 1), (16384, 1280, 5120, 320 << 8)]:
    A = torch.randn(M, K, device="cuda").half(); W = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
    bias = torch.randn(N, device="cuda"); No = N // 2 if flags & 1 else N
    o16 = torch.empty(M, No, device="cuda", dtype=torch.half)
    for _ in range(3): ok(L.gdf_op_gemm(P(A), K, P(W), P(bias), None, None, No, P(o16), No, None, No, M, N, K, flags, stream()), L)
    torch.cuda.synchronize()
PY
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $set -d $O/$n -o r --output-format csv -- python3 /tmp/g8.py > /dev/null 2>&1
done
python3 $R/tools/pmc_summary.py $O > $O/summary.txt; rm -rf $O/SQ_*; cat $O/summary.txt
