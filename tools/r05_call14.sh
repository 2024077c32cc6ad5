#!/usr/bin/env bash
# Round 5, GPU call 14 (no source change): PMC of the 16x16x128 / 32-channel-chunk halo tile on one VAE shape (B4 128->128 512x512):
# does the 64-byte-row swizzle (row >> 1) & 3 read conflict-free on the hardware, how busy is the matrix pipe at 2 workgroups per CU.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$ROOT"; O="$ROOT/gpurun_out"; mkdir -p "$O"
PMC_BENCH_ARGS="--ref 0" timeout 300 bash tools/pmc_collect.sh convvae1 > "$O/r05c14_pmc_run.log" 2>&1
cp "$O/pmc_convvae1.txt" "$O/r05c14_pmc_conv_halo_16x16x128_ck32.txt" 2>/dev/null
grep -A24 "conv3x3_halo_kernel" "$O/r05c14_pmc_conv_halo_16x16x128_ck32.txt" | head -60
