"""Same-box A/B of host-side switches: one bench.py subprocess per setting, one table.

Boxes of the pool differ by up to 12 % on byte-identical kernels, so a switch is only ever judged against
its neighbour rows of ONE call. Every row also prints the dK/dV (train) or level-0 forward (regional) time of the run as the
clock reference of the box.

    python tools/ab_switches.py --half train    "" "MOS_GN_FINALIZE=0" "MOS_FUSE_ADD_LN=0 MOS_FUSE_GN_RES=0"
    python tools/ab_switches.py --half regional "" "MOS_CONV3X3_MIN_PIXELS=3072" "MOS_CONV3X3_MIN_PIXELS=1024"

A setting is a space-separated list of NAME=VALUE pairs ("" = defaults). About 40 s per train row, 60 s per regional row.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(setting, half, steps, warmup, timeout):
    env = dict(os.environ)
    for kv in setting.split():
        k, v = kv.split('=', 1)
        env[k] = v
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--no-cpu-baseline', '--steps', str(steps), '--warmup', str(warmup)]
    cmd += ['--no-regional'] if half == 'train' else ['--mode', 'regional']
    t0 = time.time()
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    line = next((l for l in reversed(p.stdout.strip().splitlines()) if l.startswith('{')), None)
    if p.returncode != 0 or line is None:
        return dict(error=(p.stderr.strip().splitlines() or ['no output'])[-1][:200], wall=time.time() - t0)
    d = json.loads(line)
    full = os.path.join(ROOT, 'gpurun_out', 'bench_full.json')      # round 6: stdout carries the compact line, tables are in the file
    if 'full_record' in d and os.path.exists(full) and os.path.getmtime(full) >= t0:
        d = json.load(open(full))
    by_name = {k['kernel']: k['ms'] for k in (d.get('dominant_kernels_by_name') or [])}
    return dict(value=d['value'], unit=d['unit'], ms=d['ms_per_step'], clock_us=(d.get('roofline') or {}).get('avg_us'),
                lib_ms=d.get('library_kernel_ms_per_step', d.get('library_kernel_ms_per_sample')), by_name=by_name,
                wall=time.time() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('settings', nargs='+')
    ap.add_argument('--half', default='train', choices=['train', 'regional'])
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=None)
    ap.add_argument('--timeout', type=int, default=240)
    ap.add_argument('--kernels', default='conv3x3,gemm_nt,groupnorm_apply,groupnorm_stats,add_layernorm_fwd,attn_fwd,region_attn')
    args = ap.parse_args()
    steps = args.steps or (10 if args.half == 'train' else 2)
    warmup = args.warmup if args.warmup is not None else (3 if args.half == 'train' else 1)
    names = args.kernels.split(',')
    print(f'# {args.half} half, {steps} timed steps after {warmup} warm-up; per-kernel columns: ms per step / sample by kernel name')
    print(f'{"setting":44s} {"value":>10s} {"ms":>9s} {"clock us":>9s} {"lib ms":>8s} ' + ' '.join(f'{n[:14]:>14s}' for n in names))
    for s in args.settings:
        r = run(s, args.half, steps, warmup, args.timeout)
        label = s if s else '(defaults)'
        if 'error' in r:
            print(f'{label:44s} FAILED after {r["wall"]:.0f}s: {r["error"]}')
            continue
        cols = ' '.join(f'{r["by_name"].get(n, float("nan")):14.3f}' for n in names)
        print(f'{label:44s} {r["value"]:10.3f} {r["ms"]:9.3f} {r["clock_us"] or float("nan"):9.1f} {r["lib_ms"] or float("nan"):8.2f} {cols}',
              flush=True)


if __name__ == '__main__':
    main()
