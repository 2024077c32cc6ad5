"""Library launches of ONE eager ED-LoRA training step (configs[1]) -- or, --mode regional, of ONE eager 512x768 regional sample
(configs[4]: adapter, 50 UNet steps, VAE decode) -- by kernel family: calls, total and average time, from the library's own
HIP-event profiler (mixofshow.hip.profiler). GPU only.   python tools/count_library_launches.py [--steps 2] [--top 40]"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mos_path  # noqa: E402,F401
import torch  # noqa: E402

import bench  # noqa: E402
from mixofshow.hip import profiler  # noqa: E402
from mixofshow.pipelines.train_loop import TrainEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--preset', default='sd15')
    ap.add_argument('--mode', default='train', choices=['train', 'regional'])
    ap.add_argument('--top', type=int, default=0, help='also list the N largest (kernel, shape) records')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    if args.mode == 'regional':
        return regional(args, dev)
    trainer = bench.build_trainer(args.preset, dev)
    trainer.unet.train(), trainer.text_encoder.train()
    engine = TrainEngine(trainer, dict(bench.TRAIN_OPT, optim_g=dict(bench.TRAIN_OPT['optim_g'])), total_iter=1e9,
                         mixed_precision='fp16', channels_last=True)
    batches = [bench.synthetic_batch(4, 512, dev, i) for i in range(2)]
    for i in range(2):
        engine.step(batches[i % 2])
    torch.cuda.synchronize()
    recs = []
    with profiler.profile(recs):
        for i in range(args.steps):
            engine.step(batches[i % 2])
        torch.cuda.synchronize()
    report(recs, args.steps, args.top, 'step')


def regional(args, dev):
    H, W = 512, 768
    pipe = bench.build_regional_pipe(args.preset, dev)
    pipe.unet.to(memory_format=torch.channels_last)
    pipe.vae.to(memory_format=torch.channels_last)
    prompt, neg = bench.regional_prompt(H, W)
    latents = torch.randn((1, 4, H // 8, W // 8), generator=torch.manual_seed(14))
    pipe.keypose_adapter, pose, _ = bench.synthetic_keypose_adapter(pipe, H, W, dev, torch.float16)
    b = bench.region_px(H, W)[0]

    def sample():
        return pipe(prompt=prompt, negative_prompt=[neg], height=H, width=W, num_inference_steps=50, guidance_scale=7.5,
                    latents=latents.clone(), output_type='pil', hipgraph=False, keypose_adapter_input=pose,
                    keypose_adaptor_weight=1.0, region_keypose_adaptor_weight=f'[{b[0]}, {b[1]}, {b[2]}, {b[3]}]-0.6').images
    sample()
    torch.cuda.synchronize()
    recs = []
    with profiler.profile(recs):
        sample()
        torch.cuda.synchronize()
    report(recs, 1, args.top, 'sample')


def report(recs, steps, top, unit):
    class A:
        pass
    args = A()
    args.steps, args.top = steps, top
    fam = collections.OrderedDict()
    for r in recs:
        k = r['name'].split(' ')[0]
        f = fam.setdefault(k, [0, 0.0])
        f[0] += r['calls']
        f[1] += r['total_ms']
    print(f'{"kernel family":34s} {"calls/" + unit:>12s} {"ms/" + unit:>10s} {"avg us":>8s}')
    for k, (c, ms) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print(f'{k:34s} {c / args.steps:10.1f} {ms / args.steps:9.3f} {ms * 1e3 / max(c, 1):8.1f}')
    if args.top:
        print()
        for r in sorted(recs, key=lambda r: -r['total_ms'])[:args.top]:
            gb = r['bytes'] / max(r['avg_us'], 1e-9) / 1e3
            tf = r['flops'] / max(r['avg_us'], 1e-9) / 1e6
            print(f"{r['name'][:78]:78s} {r['calls'] / args.steps:6.1f} {r['avg_us']:8.1f} us {r['total_ms'] / args.steps:7.3f} ms {tf:7.1f} TF/s {gb:7.0f} GB/s")
        print()
    print(f'{"all":34s} {sum(c for c, _ in fam.values()) / args.steps:10.1f} {sum(m for _, m in fam.values()) / args.steps:9.3f}')


if __name__ == '__main__':
    main()
