#!/bin/bash
# The Gaussian on the matrix cores (RGBL_GAUSS_MFMA=1) against k_gauss7 - parity, step A/B on both workloads, instruction counts
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -n 4 -k "matrix_cores or cfg2 or stages" ) > gpurun_out/tests_gauss.log 2>&1; tail -3 gpurun_out/tests_gauss.log
bash tools/gpu_4k_ab.sh "RGBL_GAUSS_MFMA=0 --workload kitti --steps 30" "RGBL_GAUSS_MFMA=1 --workload kitti --steps 30" "RGBL_GAUSS_MFMA=0 --workload 4k" "RGBL_GAUSS_MFMA=1 --workload 4k" "RGBL_GAUSS_MFMA=0 --workload kitti --steps 30" "RGBL_GAUSS_MFMA=1 --workload kitti --steps 30"
for m in 0 1; do bash tools/gpu_pmc.sh g$m "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" RGBL_GAUSS_MFMA=$m | grep -E "gauss"; done
