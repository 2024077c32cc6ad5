"""Where do the small ATen kernels of a training step come from?  (CPU-only tool; nothing here is product code.)

rocprofv3 shows ~1.4 k ATen launches per configs[1] step (adds, strided copies, casts, fills: 6.6 of 41.5 ms) but not who
issues them, and torch's profiler returns no Python stacks on this image. This tool replays ONE forward+backward of the
ED-LoRA trainer on the CPU with the HOST code taking exactly the branches it takes on the GPU box:

  * the kernel-backed primitives (mixofshow.hip.ops) are replaced by the oracle's torch emulation and counted as ONE
    launch each (`mos::<name>`), their inner torch ops are not counted;
  * `Tensor.is_cuda` and the autocast queries are patched to answer like the GPU box (half autocast, device tensors), so
    functional.py / the models take their HIP branches; real arithmetic runs under CPU autocast (bf16);
  * a TorchDispatchMode logs every remaining ATen op that launches a kernel (views / metadata ops excluded) with dtype,
    shape, density and the Python call site; for ops issued by autograd's C++ nodes the site is the FORWARD call site of
    the node (anomaly-mode metadata).

CPU autocast casts fewer ops to fp32 than CUDA autocast (sum / softmax / mse_loss ...): the counts of those few ops are
approximate; everything that goes through functional.py is exact.

    python tools/trace_aten.py [--size 128] [--batch 2] [--preset sd15] [--top 60]
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

AMP = torch.bfloat16

VIEW_OPS = {
    'view', '_unsafe_view', 'reshape', 'permute', 'transpose', 't', 'expand', 'slice', 'select', 'unsqueeze', 'squeeze',
    'detach', 'alias', 'as_strided', 'split', 'split_with_sizes', 'unbind', 'chunk', 'narrow', 'view_as', 'unfold',
    'lift_fresh', 'empty', 'empty_like', 'empty_strided', 'new_empty', 'new_empty_strided', 'sym_size', 'sym_stride',
    'is_same_size', '_local_scalar_dense', 'device', 'resize_', 'set_', 'is_pinned', 'is_contiguous', 'stride', 'size',
    'dim', 'numel', 'storage_offset', 'sym_numel', 'sym_storage_offset', 'result_type', 'can_cast', '_has_compatible_shallow_copy_type',
    'is_nonzero', 'item', '_reshape_alias', 'movedim', 'flatten', 'unflatten', 'diagonal', 'real', 'imag', 'conj', 'is_complex',
    'is_floating_point', '_version', 'record_stream', 'prim_layout', 'layout', 'dtype', '_nested_tensor_size', 'is_coalesced',
}


def _site(skip_prefixes):
    """Innermost frame inside the package (or trainer) that is not functional.py's plumbing, plus the functional frame."""
    frames = traceback.extract_stack()
    pkg = []
    for f in frames:
        fn = f.filename
        if 'mix-of-show_amd' in fn or fn.endswith('trace_aten.py') and f.name not in ('__torch_dispatch__', '_site', 'wrapper'):
            pkg.append(f'{os.path.basename(fn)}:{f.lineno}({f.name})')
    return ' < '.join(reversed(pkg[-3:])) if pkg else '?'


def _fwd_site_of_node(node):
    tb = node.metadata.get('traceback_') if node is not None else None
    if not tb:
        return None
    pkg = []
    for line in tb:
        for ln in line.splitlines():
            ln = ln.strip()
            if ln.startswith('File') and 'mix-of-show_amd' in ln:
                parts = ln.split(',')
                fn = os.path.basename(parts[0].split('"')[1])
                pkg.append(f'{fn}:{parts[1].strip().split()[-1]}({parts[2].strip().split()[-1]})')
    return ' < '.join(reversed(pkg[-3:])) if pkg else None


class Tracer(TorchDispatchMode):

    def __init__(self):
        super().__init__()
        self.depth = 0          # > 0 while inside an emulated mos primitive
        self.phase = 'fwd'
        self.rows = collections.Counter()
        self.example = {}

    def note(self, name, site, desc=''):
        key = (self.phase, name, desc, site)
        self.rows[key] += 1

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        out = func(*args, **kwargs)
        if self.depth:
            return out
        name = func._schema.name.split('::')[-1]
        if name in VIEW_OPS or name.startswith('_assert') or name.startswith('sym_'):
            return out
        # ops that are views when nothing has to change
        ts = [a for a in list(args) + list(kwargs.values()) if torch.is_tensor(a)]
        if name in ('_to_copy', 'to', 'contiguous', 'clone', 'copy_') and not ts:
            return out
        desc = ''
        if ts:
            a = ts[0]
            o = out if torch.is_tensor(out) else a
            dense = all(t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)) for t in ts)
            same = all(t.stride() == ts[0].stride() for t in ts if t.shape == ts[0].shape)
            desc = (f'{str(a.dtype)[6:]}->{str(o.dtype)[6:]} {tuple(a.shape)}' + ('' if dense else ' STRIDED')
                    + ('' if same else ' MIXED-LAYOUT'))
        node = torch._C._current_autograd_node() if self.phase == 'bwd' else None
        site = _site(None)
        if self.phase == 'bwd' and (site == '?' or 'backward' not in site):
            fs = _fwd_site_of_node(node)
            site = f'[{type(node).__name__ if node is not None else "engine"}] of {fs}' if fs else f'[{node.name() if node is not None else "engine"}]'
        self.note('aten::' + name, site, desc)
        return out


def install_emulation(tracer):
    from oracle import emu_ops
    import mixofshow.hip.ops as ops

    def wrap(name, fn):

        def wrapper(*a, **k):
            if tracer.depth == 0:
                tracer.note('mos::' + name, _site(None))
            tracer.depth += 1
            try:
                with torch.autocast('cpu', enabled=False):       # a kernel is opaque to autocast
                    out = fn(*a, **k)
                # the kernels return 4-D results in the memory format of their input; torch's CPU ops do not always
                x4 = next((t for t in a if torch.is_tensor(t) and t.dim() == 4), None)
                if x4 is not None and not x4.is_contiguous() and x4.is_contiguous(memory_format=torch.channels_last):
                    fix = lambda t: (t.contiguous(memory_format=torch.channels_last)
                                     if torch.is_tensor(t) and t.dim() == 4 else t)
                    out = tuple(fix(t) for t in out) if isinstance(out, tuple) else fix(out)
                return out
            finally:
                tracer.depth -= 1

        return wrapper

    for name in emu_ops.EMULATED:
        setattr(ops, name, wrap(name, getattr(emu_ops, name)))


def patch_device_queries():
    torch.Tensor.is_cuda = property(lambda self: True)
    real_enabled, real_dtype = torch.is_autocast_enabled, torch.get_autocast_dtype
    torch.is_autocast_enabled = lambda device_type=None: real_enabled('cpu')
    torch.get_autocast_dtype = lambda device_type=None: real_dtype('cpu')


def report(tracer, header, top, out):
    rows = sorted(tracer.rows.items(), key=lambda kv: -kv[1])
    lines = [header]
    by_op, by_mos = collections.Counter(), collections.Counter()
    for (phase, name, desc, site), n in rows:
        (by_mos if name.startswith('mos::') else by_op)[(phase, name.split('::')[1])] += n
    lines.append(f'# launches: library {sum(by_mos.values())}, ATen {sum(by_op.values())}')
    lines.append('# ATen launches by op: ' + ', '.join(f'{p}:{o}={n}' for (p, o), n in by_op.most_common(40)))
    lines.append('# library launches by primitive: ' + ', '.join(f'{p}:{o}={n}' for (p, o), n in by_mos.most_common()))
    shown = 0
    for (phase, name, desc, site), n in rows:
        if name.startswith('mos::'):
            continue
        lines.append(f'{n:5d} {phase} {name[6:]:22s} {desc:60s} {site}')
        shown += 1
        if shown >= top:
            break
    text = '\n'.join(lines)
    print(text)
    if out:
        with open(out, 'w') as f:
            f.write(text + '\n')


def trace_regional(args, tracer, bench):
    """One UNet call (CFG pair) of the regional sampling pipeline in fp16, third step of a 3-step run (caches warm)."""
    from mixofshow.hip import functional as F_hip
    H, W = args.size, args.size * 3 // 2
    pipe = bench.build_regional_pipe(args.preset, 'cpu')
    pipe.unet.to(memory_format=torch.channels_last)
    px = [[int(b[0] * H / 512), int(b[1] * W / 768), int(b[2] * H / 512), int(b[3] * W / 768)] for b in bench.REGION_PX]
    ctx = 'three people near the castle, 4K, high quality, high resolution, best quality'
    neg = 'longbody, lowres, bad anatomy'
    regs = ['a <potter1> <potter2>, in Hogwarts uniform', 'a <hermione1> <hermione2>, girl', 'a <thanos1> <thanos2>, purple armor']
    prompt = [(ctx, [(p, neg, [b[0] / H, b[1] / W, b[2] / H, b[3] / W]) for p, b in zip(regs, px)])]
    real_px, bench.REGION_PX = bench.REGION_PX, px
    adapter_states = bench.synthetic_adapter_states(pipe, H, W, 'cpu', torch.float16)
    bench.REGION_PX = real_px
    latents = torch.randn((1, 4, H // 8, W // 8), generator=torch.manual_seed(14))
    patch_device_queries()
    F_hip._conv_min_pixels = 0
    calls = {'n': 0}
    real_forward = pipe.unet.forward

    def forward(*a, **k):
        calls['n'] += 1
        if calls['n'] == 3:
            with tracer:
                return real_forward(*a, **k)
        return real_forward(*a, **k)

    pipe.unet.forward = forward
    pipe(prompt=prompt, negative_prompt=[neg], height=H, width=W, num_inference_steps=3, guidance_scale=7.5,
         latents=latents, output_type='latent', hipgraph=False, adapter_states=adapter_states)
    report(tracer, f'# one regional UNet call (CFG pair, {H}x{W}, 3 regions, fp16, adapter states), preset {args.preset}', args.top,
           args.out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--preset', default='sd15')
    ap.add_argument('--size', type=int, default=128)
    ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--top', type=int, default=80)
    ap.add_argument('--out', default=None)
    ap.add_argument('--mode', default='train', choices=['train', 'regional'])
    args = ap.parse_args()
    torch.manual_seed(0)
    import bench
    from mixofshow.hip import functional as F_hip
    from mixofshow.parallel import dp
    from mixofshow.pipelines.train_loop import TrainEngine

    tracer = Tracer()
    install_emulation(tracer)
    if args.mode == 'regional':
        return trace_regional(args, tracer, bench)
    trainer = bench.build_trainer(args.preset, 'cpu')
    trainer.unet.train()
    trainer.text_encoder.train()
    # what TrainEngine.__init__ does on the GPU box (its `dev.type == 'cuda'` branches)
    TrainEngine._store_frozen_weights_in_half(trainer, AMP)
    trainer.unet.to(memory_format=torch.channels_last)
    trainer.vae.to(memory_format=torch.channels_last)
    bucket = dp.FlatGradBucket(trainer.trainable_parameters())
    batch = bench.synthetic_batch(args.batch, args.size, 'cpu', 0)
    images = batch['images'].contiguous(memory_format=torch.channels_last)
    patch_device_queries()
    F_hip._conv_min_pixels = 0 if args.size < 512 else F_hip._conv_min_pixels
    # warm-up step: operand caches (half weights, fused QKV weights, conv operands) are built once, not per step
    bucket.zero()
    with F_hip.direct_grad_accumulation(defer_finals=False):
        with torch.autocast('cpu', dtype=AMP):
            loss = trainer(images, batch['prompts'], batch['masks'], batch['img_masks'])
        loss.backward()
    tracer.rows.clear()
    torch.autograd.set_detect_anomaly(True, check_nan=False)
    with tracer:
        bucket.zero()
        with F_hip.direct_grad_accumulation(defer_finals=False):
            with torch.autocast('cpu', dtype=AMP):
                loss = trainer(images, batch['prompts'], batch['masks'], batch['img_masks'])
            tracer.phase = 'bwd'
            loss.backward()
    rows = sorted(tracer.rows.items(), key=lambda kv: -kv[1])
    lines = []
    tot = collections.Counter()
    for (phase, name, desc, site), n in rows:
        tot[(phase, name.startswith('mos::'))] += n
    lines.append(f'# one forward+backward, preset {args.preset}, batch {args.batch}, {args.size}px; launches: '
                 f'fwd mos {tot[("fwd", True)]} aten {tot[("fwd", False)]}; bwd mos {tot[("bwd", True)]} aten {tot[("bwd", False)]}')
    by_op = collections.Counter()
    for (phase, name, desc, site), n in rows:
        if not name.startswith('mos::'):
            by_op[(phase, name)] += n
    lines.append('# ATen launches by op: ' + ', '.join(f'{p}:{o[6:]}={n}' for (p, o), n in by_op.most_common(40)))
    by_mos = collections.Counter()
    for (phase, name, desc, site), n in rows:
        if name.startswith('mos::'):
            by_mos[(phase, name[5:])] += n
    lines.append('# library launches by primitive: ' + ', '.join(f'{p}:{o}={n}' for (p, o), n in by_mos.most_common()))
    shown = 0
    for (phase, name, desc, site), n in rows:
        if name.startswith('mos::'):
            continue
        lines.append(f'{n:5d} {phase} {name[6:]:22s} {desc:60s} {site}')
        shown += 1
        if shown >= args.top:
            break
    text = '\n'.join(lines)
    print(text)
    if args.out:
        with open(args.out, 'w') as f:
            f.write(text + '\n')


if __name__ == '__main__':
    main()
