"""Library launches of ONE eager ED-LoRA training step (configs[1]) by kernel family: calls, total and average time, from the
library's own HIP-event profiler (mixofshow.hip.profiler). GPU only.   python tools/count_library_launches.py [--steps 2]"""
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
    ap.add_argument('--top', type=int, default=0, help='also list the N largest (kernel, shape) records')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
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
    fam = collections.OrderedDict()
    for r in recs:
        k = r['name'].split(' ')[0]
        f = fam.setdefault(k, [0, 0.0])
        f[0] += r['calls']
        f[1] += r['total_ms']
    print(f'{"kernel family":34s} {"calls/step":>10s} {"ms/step":>9s} {"avg us":>8s}')
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
