#!/usr/bin/env bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
reg() { timeout 600 python bench.py --mode regional --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'ms/sample; lib', d['library_kernel_ms_per_sample'])"; }
step() { timeout 600 python bench.py --no-cpu-baseline --no-regional --steps 16 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'images/s', d['ms_per_step'], 'ms')"; }
echo "== regional: all on"; reg
echo "== regional: conv1x1 off"; MOS_CONV1X1=0 reg
echo "== regional: conv3x3+1x1 off"; MOS_CONV3X3=0 reg
echo "== regional: all on (again)"; reg
echo "== train: all on"; step
echo "== train: conv1x1 off"; MOS_CONV1X1=0 step
