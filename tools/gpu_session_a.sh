#!/usr/bin/env bash
# round-2 GPU session A: new LoRA/GEMM kernels (parity + micro-benchmark), corrected e2e parity, attention variants
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out; mkdir -p $O
echo "== primitives (lora / linear / pack)"; timeout 600 python -m pytest tests/test_gpu_primitives.py -m gpu -q -s -k "lora or linear or pack" > $O/r02_prim_lora.log 2>&1; tail -3 $O/r02_prim_lora.log
echo "== gemm micro-benchmark"; timeout 300 python tools/bench_kernels.py --only gemm --iters 20 --legacy 1 2>&1 | grep -v JSON > $O/r02_kb_gemm.txt; cat $O/r02_kb_gemm.txt
echo "== e2e parity"; timeout 1500 python -m pytest tests/test_gpu_end_to_end.py -m gpu -q -s --timeout 900 -k "hot_path or fp16_pipeline or written_out or hipgraph or smoke or training_steps" > $O/r02_e2e_b.log 2>&1; grep -E "parity|smoke:|passed|failed" $O/r02_e2e_b.log | cut -c1-700
echo "== attention variants"; timeout 900 bash tools/try_variants.sh run > $O/r02_variants_attn.txt 2>&1; cat $O/r02_variants_attn.txt
