"""Device-side idle time of the graph-replayed step: from a `rocprofv3 --kernel-trace` CSV (start / end timestamps of every
kernel), the busy time, the idle gaps between consecutive kernels and where they sit, over the replayed steps only.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-regional
    python tools/trace_gaps.py gpurun_out/trace
"""
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    paths = glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True)
    assert paths, f'no kernel trace under {root}'
    rows = []
    with open(max(paths, key=os.path.getsize)) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    # a replayed step = the span between two launches of the fused AdamW / first kernel of the step: use the dK/dV level-0 kernel
    # (5 per step) as the step marker and keep the last 4 complete steps
    marks = [i for i, r in enumerate(rows) if 'attn_bwd_dkdv' in r[2] and '40' in r[2]]
    per_step = 5
    steps = [marks[i] for i in range(0, len(marks), per_step)]
    if len(steps) < 6:
        print('too few steps in the trace', len(steps))
        return
    lo, hi = steps[-5], steps[-1]              # 4 steps, marker to marker
    sel = rows[lo:hi]
    span = sel[-1][1] - sel[0][0]
    busy = 0
    gaps = []
    end = sel[0][0]
    for s, e, name in sel:
        if s > end:
            gaps.append((s - end, name))
            busy += e - s
            end = e
        else:                                   # overlap with the previous kernel (concurrent streams)
            busy += max(0, e - end)
            end = max(end, e)
    n = len(sel)
    print(f'kernels in 4 replayed steps: {n} ({n / 4:.0f} per step); span {span / 4e6:.3f} ms per step; busy {busy / 4e6:.3f} ms; '
          f'idle {(span - busy) / 4e6:.3f} ms per step ({100.0 * (span - busy) / span:.1f} %)')
    gs = sorted(g for g, _ in gaps)
    if gs:
        print(f'gaps: {len(gs) / 4:.0f} per step, median {gs[len(gs) // 2] / 1e3:.2f} us, p90 {gs[int(len(gs) * 0.9)] / 1e3:.2f} us, '
              f'max {gs[-1] / 1e3:.1f} us, sum of gaps > 10 us: {sum(g for g in gs if g > 10000) / 4e6:.3f} ms per step')
    agg = {}
    for g, name in gaps:
        k = name.split('(')[0][:70]
        a = agg.setdefault(k, [0, 0])
        a[0] += g
        a[1] += 1
    print('idle time in front of (top 12 by total):')
    for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:12]:
        print(f'  {t / 4e6:7.3f} ms/step  {c / 4:6.1f} x/step  {t / c / 1e3:6.2f} us each  {k}')
    dur = {}
    for s, e, name in sel:
        k = name.split('(')[0][:70]
        a = dur.setdefault(k, [0, 0])
        a[0] += e - s
        a[1] += 1
    print('kernel time (top 25 by total):')
    for k, (t, c) in sorted(dur.items(), key=lambda kv: -kv[1][0])[:25]:
        print(f'  {t / 4e6:7.3f} ms/step  {c / 4:6.1f} x/step  {t / c / 1e3:7.2f} us each  {k}')


if __name__ == '__main__':
    main()
