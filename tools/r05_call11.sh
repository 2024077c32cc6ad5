#!/usr/bin/env bash
# Round 5, GPU call 11: the halo convolution without per-step address VALU (scalar-offset DMA, pattern + immediate fragment reads,
# VGPRs 207 -> 90), and its 32-channel-chunk variants (3-4 workgroups per CU). Same box: kernel bench of every conv shape, the conv
# parity cases per library, then the whole step / sample per library.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$ROOT"; O="$ROOT/gpurun_out"; mkdir -p "$O"
TAG=r05c11
V="$ROOT/_variants"
declare -A LIBS=( [head]="$V/libmos_hip_head.so" [new]="" [ck32_bn64]="$V/libmos_hip_ck32_bn64.so" [ck32_bn128]="$V/libmos_hip_ck32_bn128.so" )
ORDER="head new ck32_bn64 ck32_bn128"
for name in $ORDER; do
  echo "== kernel bench [$name]"
  MOS_HIP_LIB="${LIBS[$name]}" timeout 200 python tools/bench_kernels.py --only conv --ref 0 --iters 30 2>&1 | grep -E "^B[0-9]|conv3x3 B" | sed "s/^/[$name] /"
done > "$O/${TAG}_kernel_bench_conv_variants.txt" 2>&1
python - "$O/${TAG}_kernel_bench_conv_variants.txt" <<'PY'
import sys,re,collections
rows=collections.OrderedDict()
for l in open(sys.argv[1]):
    m=re.match(r'\[(\w+)\] (B\d+ \S+ \S+)\s+([\d.]+)\s+nan\s+([\d.]+)',l)
    if m: rows.setdefault(m.group(2),{})[m.group(1)]=(float(m.group(3)),float(m.group(4)))
names=['head','new','ck32_bn64','ck32_bn128']
print('%-26s'%'shape (fwd/bwd us)'+''.join('%18s'%n for n in names))
for s,d in rows.items():
    print('%-26s'%s+''.join('%18s'%('%.1f/%.1f'%d[n] if n in d else '-') for n in names))
PY
for name in new ck32_bn64 ck32_bn128; do
  echo "== conv parity [$name]"
  MOS_HIP_LIB="${LIBS[$name]}" timeout 300 python -m pytest tests/test_gpu_primitives.py -m gpu -q -k "conv3x3" 2>&1 | tail -2
done
for rep in 1 2; do
for name in $ORDER; do
  MOS_HIP_LIB="${LIBS[$name]}" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=[k for k in d['dominant_kernels_by_name'] if k['kernel'].startswith('conv3x3')]
r=d.get('regional',{})
print('[$name] train', d['value'], 'img/s', d['ms_per_step'], 'ms; regional', r.get('value_ms_latent'), '/', r.get('value_ms_image'), 'ms;', [(k['kernel'], k['ms'], k.get('frac_of_mfma_peak')) for k in c])"
done; done 2>&1 | tee "$O/${TAG}_ab_same_box_conv_variants.txt"
