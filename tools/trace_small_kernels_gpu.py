"""Which host ops launch the SMALL device kernels of one eager ED-LoRA training step (configs[1])?
rocprofv3 (profiles/r05_rocprofv3_kernel_stats_bench_train.csv) counts per step ~137 __amd_rocclr_copyBuffer, ~98 FillFunctor<float>
and ~67 float16_copy launches (3 % of the device time together); the ATen-level tracer (tools/trace_copies_gpu.py) sees ~80 copy-like
ops in forward + backward. This one asks the profiler: every device kernel of ONE whole `engine.step` (input staging, forward, backward,
optimiser tail) is attributed to the ATen op that launched it and to the innermost repo frame of that op.
    python tools/trace_small_kernels_gpu.py [--max-us 10]
GPU only; nothing here is product code."""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mos_path  # noqa: E402,F401
import torch  # noqa: E402


def repo_frames(stack, n=3):
    out = []
    for fr in stack or []:
        if ('mix-of-show_amd' in fr or 'bench.py' in fr) and 'trace_small' not in fr:
            p = fr.split('/')[-1]
            out.append(p.strip())
        if len(out) >= n:
            break
    if out:
        return ' < '.join(out)
    for fr in stack or []:
        if 'torch/' in fr and ('optim' in fr or 'amp' in fr or 'autograd' in fr):
            return 'torch: ' + fr.split('site-packages/')[-1].strip()
    return '?'


def short(name):
    for key in ('copyBuffer', 'fillBuffer', 'FillFunctor<float>', 'FillFunctor<c10::Half>', 'FillFunctor', 'float16_copy', 'direct_copy', 'CatArrayBatchedCopy',
                'multi_tensor_apply', 'reduce_kernel', 'index'):
        if key in name:
            return key
    name = name.replace('void ', '').replace('(anonymous namespace)::', '').replace('at::native::', '')
    return name[:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--max-us', type=float, default=10.0, help='kernels with a mean duration below this are listed')
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--size', type=int, default=512)
    args = ap.parse_args()
    from torch.profiler import ProfilerActivity, profile
    from bench import TRAIN_OPT, build_trainer, synthetic_batch
    from mixofshow.pipelines.train_loop import TrainEngine
    dev = torch.device('cuda', 0)
    tr = build_trainer('sd15', dev)
    tr.unet.to(memory_format=torch.channels_last)
    tr.vae.to(memory_format=torch.channels_last)
    engine = TrainEngine(tr, dict(TRAIN_OPT, optim_g=dict(TRAIN_OPT['optim_g'])), total_iter=100, mixed_precision='fp16')
    b = synthetic_batch(args.batch, args.size, dev, 0)
    for _ in range(2):
        engine.step(b)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
                 experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
        engine.step(b)
        torch.cuda.synchronize()
    rows = collections.defaultdict(lambda: [0, 0.0])
    per_kernel = collections.defaultdict(lambda: [0, 0.0])
    total_us = 0.0
    n_kernels = 0
    for ev in prof.events():
        ks = getattr(ev, 'kernels', None) or []
        if not ks:
            continue
        # only the innermost op owning the kernel: skip an op whose child also owns kernels (the child is listed itself)
        if any(getattr(c, 'kernels', None) for c in (ev.cpu_children or [])):
            own = set(id(k) for k in ks)
            for c in ev.cpu_children:
                for k in getattr(c, 'kernels', None) or []:
                    own.discard(id(k))
            ks = [k for k in ks if id(k) in own]
        for k in ks:
            n_kernels += 1
            total_us += k.duration
            where = repo_frames(ev.stack)
            par, node = ev.cpu_parent, None
            while par is not None:
                if 'Backward' in par.name or 'evaluate_function' in par.name:
                    node = par.name.replace('autograd::engine::evaluate_function: ', '')
                par = par.cpu_parent
            if node:
                where = f'[{node[:48]}] {where}'
            key = (short(k.name), ev.name[:40], where)
            rows[key][0] += 1
            rows[key][1] += k.duration
            per_kernel[short(k.name)][0] += 1
            per_kernel[short(k.name)][1] += k.duration
    print(f'# one eager engine.step (sd15, batch {args.batch}, {args.size}px): {n_kernels} device kernels / memcpys attributed, {total_us / 1e3:.2f} ms of device time')
    print('# per kernel, mean < %.0f us:' % args.max_us)
    for name, (n, us) in sorted(per_kernel.items(), key=lambda kv: -kv[1][0]):
        if us / n < args.max_us:
            print(f'  {n:5d} x {us / n:6.1f} us = {us / 1e3:6.3f} ms  {name}')
    print('# (kernel | launching op | innermost repo frames), small kernels only, by count:')
    for (kname, op, fr), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][0]):
        if us / n < args.max_us:
            print(f'{n:5d} {us / 1e3:7.3f} ms  {kname:28s} {op:40s} {fr}')


if __name__ == '__main__':
    main()
