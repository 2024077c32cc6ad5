#!/usr/bin/env bash
# Round 5, GPU call 12: VAE-sized convolutions (Cout = 128-multiples, <= 256 / >= 512 input channels) on 32-channel-chunk halo tiles.
# Variant libraries a..d (csrc/build.sh, MOS_CONV_FLAGS): a = 8x16x128 for Cin <= 256; b = 16x16x128 for Cin <= 256;
# c / d = the same plus the 16x16x128 tile of the >= 512-channel stages on 32-channel chunks (two workgroups per CU instead of one).
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$ROOT"; O="$ROOT/gpurun_out"; mkdir -p "$O"
TAG=r05c12
V="$ROOT/_variants"
declare -A LIBS=( [new]="" [a]="$V/libmos_hip_vae_a.so" [b]="$V/libmos_hip_vae_b.so" [c]="$V/libmos_hip_vae_c.so" [d]="$V/libmos_hip_vae_d.so" )
ORDER="new a b c d"
for name in $ORDER; do
  MOS_HIP_LIB="${LIBS[$name]}" timeout 200 python tools/bench_kernels.py --only convvae --ref 0 --iters 30 2>&1 | grep -E "^B[0-9]|conv3x3 B" | sed "s/^/[$name] /"
done > "$O/${TAG}_kernel_bench_conv_vae_tiles.txt" 2>&1
python - "$O/${TAG}_kernel_bench_conv_vae_tiles.txt" <<'PY'
import sys,re,collections
rows=collections.OrderedDict()
seen=collections.Counter()
for l in open(sys.argv[1]):
    m=re.match(r'\[(\w+)\] (B\d+ \S+ \S+)\s+([\d.]+)\s+nan\s+([\d.]+)',l)
    if m:
        k=(m.group(1),m.group(2)); seen[k]+=1
        rows.setdefault(m.group(2)+(' #%d'%seen[k] if seen[k]>1 else ''),{})[m.group(1)]=(float(m.group(3)),float(m.group(4)))
names=['new','a','b','c','d']
print('%-28s'%'shape (fwd/bwd us)'+''.join('%16s'%n for n in names))
for s,d in rows.items():
    print('%-28s'%s+''.join('%16s'%('%.1f/%.1f'%d[n] if n in d else '-') for n in names))
PY
for name in b d; do
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
done; done 2>&1 | tee "$O/${TAG}_ab_same_box_conv_vae_tiles.txt"
