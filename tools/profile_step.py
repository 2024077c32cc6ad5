"""Steady-state breakdown of one training step (or one regional UNet call) with torch.profiler: which kernels the
step spends its GPU time in, how many launches it makes, and GPU-busy time vs wall time (launch-bound or not).

  python tools/profile_step.py --mode train   > gpurun_out/step_breakdown_train.txt
  python tools/profile_step.py --mode regional
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mos_path  # noqa: E402,F401
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', default='train')
    ap.add_argument('--precision', default='fp16')
    ap.add_argument('--rows', type=int, default=70)
    ap.add_argument('--ops', type=int, default=0, help='also attribute the ATen elementwise / copy / fill kernels to the Python '
                    'source lines that launched them (top N rows)')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    if args.mode == 'train':
        from mixofshow.pipelines.train_loop import TrainEngine
        tr = bench.build_trainer('sd15', dev)
        tr.unet.train(); tr.text_encoder.train()
        eng = TrainEngine(tr, dict(bench.TRAIN_OPT, optim_g=dict(bench.TRAIN_OPT['optim_g'])), 1e9, args.precision)
        b = bench.synthetic_batch(4, 512, dev, 0)
        fn = lambda: eng.step(b)  # noqa: E731
    else:
        pipe = bench.build_regional_pipe('sd15', dev)
        prompt, neg = bench.regional_prompt(512, 768)
        lat = torch.randn((1, 4, 64, 96), generator=torch.manual_seed(14))
        fn = lambda: pipe(prompt=prompt, negative_prompt=[neg], height=512, width=768, num_inference_steps=5,  # noqa: E731
                          guidance_scale=7.5, latents=lat.clone(), output_type='latent')
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=bool(args.ops)) as prof:
        fn()
        torch.cuda.synchronize()
    # device-side kernel records only (operator rows double count their kernels)
    from torch.autograd import DeviceType
    agg = {}
    for e in prof.events():
        if e.device_type == DeviceType.CUDA:
            t = getattr(e, 'device_time_total', None)
            if t is None:
                t = e.cuda_time_total
            a = agg.setdefault(e.name, [0.0, 0])
            a[0] += t
            a[1] += 1
    rows = sorted(((t, c, k) for k, (t, c) in agg.items()), reverse=True)
    busy = sum(r[0] for r in rows)
    launches = sum(r[1] for r in rows)

    def cat(k):
        kl = k.lower()
        if 'igemm' in kl or 'conv' in kl and 'ck' in kl or 'miopen' in kl or 'naive_conv' in kl:
            return 'MIOpen convolution'
        if 'batched_transpose' in kl or 'subtensorop' in kl:
            return 'MIOpen layout / tensor ops'
        if kl.startswith('cijk_') or 'custom_cijk' in kl:
            return 'hipBLASLt / rocBLAS GEMM'
        if '_global__n_1' in kl or 'anonymous namespace' in kl and ('lora' in kl or 'gn_' in kl):
            return 'libmos_hip (this library)'
        if 'attn_fwd' == kl or 'bwd_kernel' in kl:
            return 'aotriton attention'
        if 'copy' in kl or 'cat' in kl:
            return 'ATen copies / cat'
        if 'elementwise' in kl or 'reduce' in kl or 'fill' in kl.lower():
            return 'ATen elementwise / reduce / fill'
        if 'multi_tensor' in kl or 'adam' in kl:
            return 'optimizer'
        return 'other'

    cats = {}
    for t, c, k in rows:
        a = cats.setdefault(cat(k), [0.0, 0])
        a[0] += t
        a[1] += c
    print(f'mode={args.mode} wall per call = {wall * 1e3:.2f} ms (eager launch) ; GPU busy (sum of kernel time) = {busy / 1e3:.2f} ms ; '
          f'kernel launches = {launches}')
    for name, (t, c) in sorted(cats.items(), key=lambda kv: -kv[1][0]):
        print(f'  {t / 1e3:9.3f} ms {100 * t / busy:5.1f}%  x{c:<5d} {name}')
    print()
    for t, c, k in rows[:args.rows]:
        print(f'{t / 1e3:9.3f} ms {100 * t / busy:5.1f}%  x{c:<5d} {k[:150]}')
    if args.ops:
        # ATen ops (not this library's kernels): device time by operator and by the innermost frames inside this repository
        want = ('aten::add', 'aten::add_', 'aten::copy_', 'aten::mul', 'aten::mul_', 'aten::fill_', 'aten::zero_', 'aten::cat',
                'aten::sub', 'aten::div', 'aten::where', 'aten::sum', 'aten::mean', 'aten::neg', 'aten::sigmoid', 'aten::silu',
                'aten::clone', 'aten::_to_copy', 'aten::contiguous', 'aten::zeros', 'aten::zeros_like', 'aten::empty_like')
        agg2 = {}
        for e in prof.key_averages(group_by_stack_n=12):
            if e.key not in want:
                continue
            t = getattr(e, 'self_device_time_total', None)
            if t is None:
                t = e.self_cuda_time_total
            if t <= 0:
                continue
            frames = [f for f in (e.stack or []) if ('mixofshow' in f or 'bench.py' in f or 'train_loop' in f) and 'profile_step' not in f]
            where = ' <- '.join(f.split('/')[-1].strip() for f in frames[:3]) or '(torch internals / autograd engine)'
            a = agg2.setdefault((e.key, where), [0.0, 0])
            a[0] += t
            a[1] += e.count
        print()
        print('ATen operator device time by launching source line (self device time):')
        for (op, where), (t, c) in sorted(agg2.items(), key=lambda kv: -kv[1][0])[:args.ops]:
            print(f'{t / 1e3:9.3f} ms  x{c:<5d} {op:18s} {where[:170]}')


if __name__ == '__main__':
    main()
