#!/bin/bash
# PMC passes over the attention micro-benchmark (one variant, fusion shape).  Each --pmc pass is its own run.
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
V=${1:-72}; VIEWS=${2:-100}
rocprofv3 -L > gpurun_out/pmc/counters_list.txt 2>&1
CMD="python tools/kernel_bench.py --what attnonly --variants $V --views $VIEWS"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES -d gpurun_out/pmc/p1 --output-format csv -- $CMD > gpurun_out/pmc/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA -d gpurun_out/pmc/p2 --output-format csv -- $CMD > gpurun_out/pmc/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VALU_TRANS -d gpurun_out/pmc/p3 --output-format csv -- $CMD > gpurun_out/pmc/p3.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc/p4 --output-format csv -- $CMD > gpurun_out/pmc/p4.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d gpurun_out/pmc/p5 --output-format csv -- $CMD > gpurun_out/pmc/p5.log 2>&1
find gpurun_out/pmc -name "*.csv" | head -20
for p in p1 p2 p3 p4 p5; do tail -3 gpurun_out/pmc/$p.log; done
