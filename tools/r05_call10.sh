#!/usr/bin/env bash
# call 10: attribute the small kernels of one eager training step to their launching ops (torch profiler)
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$ROOT"; O="$ROOT/gpurun_out"; mkdir -p "$O"
timeout 420 python tools/trace_small_kernels_gpu.py > "$O/r05c10_small_kernels.txt" 2> "$O/r05c10_small_kernels.err"
echo "rc=$?"; tail -3 "$O/r05c10_small_kernels.err"; head -60 "$O/r05c10_small_kernels.txt" | cut -c1-230
