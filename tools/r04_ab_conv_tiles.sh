#!/usr/bin/env bash
set -u
TAG="${1:-r04g}"
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
O="$ROOT/gpurun_out"; mkdir -p "$O"
timeout 300 python tools/ab_switches.py --half train --kernels conv3x3,gemm_nt "" "MOS_CONV_SPLIT_MAX_TILES=640" "MOS_CONV_SPLIT_MAX_TILES=1024" > "$O/${TAG}_ab_train.txt" 2>&1
cut -c1-150 "$O/${TAG}_ab_train.txt"
timeout 500 python tools/ab_switches.py --half regional --steps 4 --timeout 300 --kernels conv3x3,gemm_nt "" "MOS_CONV_SPLIT_MAX_TILES=640" "MOS_CONV_SPLIT_MAX_TILES=1024" > "$O/${TAG}_ab_regional.txt" 2>&1
cut -c1-150 "$O/${TAG}_ab_regional.txt"
