"""Which ATen copies does one EAGER ED-LoRA training step (configs[1]: sd15, batch 4, 512x512, fp16 autocast) issue on the device?
rocprofv3 counts ~50 strided copies (direct_copy_kernel, ~20 us each), ~135 plain device-to-device copies and ~65 dtype casts per
step; this logs every copy-like ATen op of one forward + backward with shape, strides-class, bytes and call site (Python frame for
ops issued from Python, autograd node + its forward site for ops issued by the engine). GPU only; nothing here is product code.
    python tools/trace_copies_gpu.py [--min-bytes 65536]
"""
import argparse
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mos_path  # noqa: E402,F401
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

COPY_OPS = {'clone', 'copy_', '_to_copy', 'contiguous', 'cat', 'stack', 'index_select', 'constant_pad_nd', 'fill_', 'zero_', 'add', 'add_'}


def site():
    out = []
    for f in traceback.extract_stack():
        fn = f.filename
        if ('mix-of-show_amd' in fn or fn.endswith(('bench.py', 'train_loop.py'))) and 'trace_copies' not in fn:
            out.append(f'{os.path.basename(fn)}:{f.lineno}({f.name})')
    return ' < '.join(reversed(out[-3:])) if out else '?'


def fwd_site(node):
    tb = node.metadata.get('traceback_') if node is not None else None
    if not tb:
        return None
    out = []
    for line in tb:
        for ln in line.splitlines():
            ln = ln.strip()
            if ln.startswith('File') and 'mix-of-show_amd' in ln:
                p = ln.split(',')
                out.append(f'{os.path.basename(p[0].split(chr(34))[1])}:{p[1].strip().split()[-1]}({p[2].strip().split()[-1]})')
    return ' < '.join(reversed(out[-3:])) if out else None


class Tracer(TorchDispatchMode):

    def __init__(self, min_bytes):
        super().__init__()
        self.phase, self.min_bytes = 'fwd', min_bytes
        self.rows = collections.Counter()
        self.bytes = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        out = func(*args, **kwargs)
        name = func._schema.name.split('::')[-1]
        if name not in COPY_OPS:
            return out
        flat = []
        for a in list(args) + list(kwargs.values()):
            if torch.is_tensor(a):
                flat.append(a)
            elif isinstance(a, (list, tuple)):
                flat += [t for t in a if torch.is_tensor(t)]
        o = out if torch.is_tensor(out) else (flat[0] if flat else None)
        if o is None or not o.is_cuda:
            return out
        nbytes = o.numel() * o.element_size()
        if nbytes < self.min_bytes:
            return out
        if name in ('add', 'add_') and all(t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)) for t in flat):
            return out                      # (plain adds are not copies; strided / mixed-layout ones behave like them)
        src = flat[-1] if name == 'copy_' and len(flat) > 1 else (flat[0] if flat else o)
        dense = src.is_contiguous() or (src.dim() == 4 and src.is_contiguous(memory_format=torch.channels_last))
        kind = 'dense' if dense else 'STRIDED'
        desc = f'{str(src.dtype)[6:]}->{str(o.dtype)[6:]} {tuple(o.shape)} {kind} strides{tuple(src.stride())}'
        s = site()
        if self.phase == 'bwd' and (s == '?' or 'backward' not in s):
            node = torch._C._current_autograd_node()
            fs = fwd_site(node)
            s = f'[{type(node).__name__ if node is not None else "engine"}] of {fs}' if fs else f'[{node.name() if node is not None else "engine"}] {s}'
        key = (self.phase, name, desc, s)
        self.rows[key] += 1
        self.bytes[key] += nbytes
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--min-bytes', type=int, default=65536)
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--size', type=int, default=512)
    args = ap.parse_args()
    from bench import TRAIN_OPT, build_trainer, synthetic_batch
    from mixofshow.hip import functional as F_hip
    from mixofshow.pipelines.train_loop import TrainEngine
    dev = torch.device('cuda', 0)
    tr = build_trainer('sd15', dev)
    tr.unet.to(memory_format=torch.channels_last)
    tr.vae.to(memory_format=torch.channels_last)
    engine = TrainEngine(tr, dict(TRAIN_OPT, optim_g=dict(TRAIN_OPT['optim_g'])), total_iter=100, mixed_precision='fp16')
    b = synthetic_batch(args.batch, args.size, dev, 0)
    engine.step(b)                                     # warm-up (caches, workspaces)
    torch.cuda.synchronize()
    t = Tracer(args.min_bytes)
    torch.autograd.set_detect_anomaly(True, check_nan=False)      # forward call sites of the autograd nodes
    images = b['images'].contiguous(memory_format=torch.channels_last)
    with t:
        engine.bucket.zero()
        with F_hip.direct_grad_accumulation(defer_finals=True, store=engine._finals_eager if hasattr(engine, '_finals_eager') else None):
            with torch.autocast('cuda', dtype=torch.float16):
                loss = tr(images, b['prompts'], b['masks'], b['img_masks'])
            t.phase = 'bwd'
            engine.scaler.scale(loss).backward()
    torch.cuda.synchronize()
    tot = collections.Counter()
    for (ph, name, desc, s), n in t.rows.items():
        tot[(ph, name, 'STRIDED' in desc)] += n
    print('# copy-like ATen ops >= %d bytes in one eager forward + backward (sd15, batch %d, %dpx): ' % (args.min_bytes, args.batch, args.size)
          + ', '.join(f'{ph}:{name}{" STRIDED" if st else ""}={n}' for (ph, name, st), n in sorted(tot.items(), key=lambda kv: -kv[1])))
    for key, n in sorted(t.rows.items(), key=lambda kv: -t.bytes[kv[0]]):
        ph, name, desc, s = key
        print(f'{n:4d} {t.bytes[key] / 1e6:9.1f} MB {ph} {name:10s} {desc:90s} {s}')


if __name__ == '__main__':
    main()
