#!/bin/bash
mkdir -p gpurun_out/d
export TMPDIR=/tmp
O=gpurun_out/d
( timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_align.py tests/test_image.py tests/test_kernels_gpu.py tests/test_focal.py tests/test_pnp.py -m gpu -q -x -p no:cacheprovider -k "portrait or align or pin or readme or image or dpt_final or load_images or cpu_tensors or golden or graph or chunk or sharded or rccl" 2>&1 | tail -30 > $O/pytest.log )
tail -5 $O/pytest.log
F3R_LAB_LIB=$PWD/tools/lab/libf3r_hip_lab.so timeout 600 python tools/kernel_bench.py --what attnonly --variants 72,75,56,70 --views 100,320 > $O/attn_variants.jsonl 2> $O/attn_variants.err
cat $O/attn_variants.jsonl; tail -2 $O/attn_variants.err
timeout 300 python tools/kernel_bench.py --what attnproduct --views 320 > $O/attn_product.jsonl 2>> $O/attn_variants.err; cat $O/attn_product.jsonl
