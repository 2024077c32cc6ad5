"""bench.py — headline benchmark of the Mix-of-Show hot path on MI355X (contract: see the task prompt / DESIGN.md).

Default (`--mode train`): BASELINE.json metric part 1, ED-LoRA train images/sec @512x512 on SD-1.5, workload =
BASELINE.json configs[1] ("Single-concept ED-LoRA tune SD-1.5 512x512 rank=4 batch=4 on 1xMI355X"); with --gpus N>1
(launched by torch.distributed.run, one rank per GPU over RCCL) the same per-GPU batch => weak scaling (configs[2]).
A "step" is one full optimisation step: VAE encode, CLIP (16 layer-wise prompts / sample), UNet forward + backward
through the fused HIP attention path, all-reduce of the LoRA+concept-row gradient bucket, AdamW, embedding-norm rule.
Weights: seeded random init of the exact SD-1.5 architecture (no checkpoints offline); data: synthetic, resident in
HBM before the timed region.

`--mode regional`: metric part 2, 50-step 3-region sample latency at 512x768 (configs[4]); a "step" is one
complete 50-step sample (CFG pair per UNet call).

Prints ONE JSON line (rank 0). `roofline` describes the library kernel with the largest share of GPU time, timed
with HIP events on its launch stream (mos_profile_*), in a short profiled pass right after the timed region so that
`value` is not perturbed by event recording; `cpu_baseline` times the oracle path (oracle/trainer_ref.py, plain
torch fp32 = the "CPU diffusers reference path" stand-in) on the host cores for a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import mos_path  # noqa: F401
import torch

PEAK_MFMA_TFLOPS = 2500.0   # dense fp16/bf16 MFMA, MI355X_MICROARCH.md
PEAK_HBM_GBPS = 8000.0

FINETUNE_CFG = dict(
    text_embedding=dict(enable_tuning=True, lr=1e-3),
    text_encoder=dict(enable_tuning=True, lora_cfg=dict(rank=4, alpha=1.0, where='CLIPAttention'), lr=1e-5),
    unet=dict(enable_tuning=True, lora_cfg=dict(rank=4, alpha=1.0, where='Attention'), lr=1e-4))
TRAIN_OPT = dict(optim_g=dict(type='AdamW', lr=0.0, weight_decay=0.01, betas=[0.9, 0.999]), emb_norm_threshold=0.55)


_T0 = time.time()


def _log(msg):
    print(f'[bench +{time.time() - _T0:7.1f}s] {msg}', file=sys.stderr, flush=True)


def _sync_barrier(world):
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()


def _max_over_ranks(x, world, device):
    t = torch.tensor([x], dtype=torch.float64, device=device)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return float(t.item())


def build_trainer(preset, device, seed=0):
    from mixofshow.pipelines.trainer_edlora import EDLoRATrainer
    torch.manual_seed(seed)
    tr = EDLoRATrainer(f'synthetic://{preset}?seed=0', '<potter1>+<potter2>', '<rand-0.013>+man', True,
                       finetune_cfg={k: dict(v) if k == 'text_embedding' else dict(v, lora_cfg=dict(v['lora_cfg']))
                                     for k, v in FINETUNE_CFG.items()},
                       noise_offset=0.01, attn_reg_weight=0.01, reg_full_identity=False, use_mask_loss=True)
    return tr.to(device)


def synthetic_batch(B, size, device, seed):
    g = torch.Generator().manual_seed(seed)
    m = size // 8
    masks = torch.zeros(B, 1, m, m)
    masks[:, :, m // 4:3 * m // 4, m // 4:3 * m // 4] = 1
    return dict(images=(torch.rand(B, 3, size, size, generator=g) * 2 - 1).to(device), masks=masks.to(device),
                img_masks=torch.ones(B, 1, m, m, device=device),
                prompts=['a <potter1> <potter2> in the park, 4K, high quality'] * B)


def _pmc_traffic(kernel_name):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/r01_pmc_traffic.json:
    FETCH_SIZE / WRITE_SIZE collected in separate rocprofv3 --pmc runs, gfx950-corrected), or None."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r01_pmc_traffic.json')
    try:
        with open(path) as f:
            ent = json.load(f).get(kernel_name)
        return float(ent['total']) if ent else None
    except (OSError, ValueError, KeyError, TypeError):
        return None


def roofline_from_profile(records):
    lib_ms = sum(r['total_ms'] for r in records)
    top = records[0]
    ai = top['flops'] / max(1.0, top['bytes'])
    bound = 'mfma' if ai > 300 else 'hbm'
    sec = top['avg_us'] * 1e-6
    if bound == 'mfma':
        achieved, peak, unit = top['flops'] / sec / 1e12, PEAK_MFMA_TFLOPS, 'TFLOP/s'
    else:
        achieved, peak, unit = top['bytes'] / sec / 1e9, PEAK_HBM_GBPS, 'GB/s'
    return dict(kernel=top['name'], bound=bound, achieved=round(achieved, 3), peak=peak, unit=unit,
                frac=round(achieved / peak, 5), traffic=_pmc_traffic(top['name']), avg_us=round(top['avg_us'], 2),
                launches=top['calls'],
                share_of_library_gpu_time=round(top['total_ms'] / max(1e-9, lib_ms), 4),
                algorithmic_flops_per_launch=top['flops'], algorithmic_bytes_per_launch=top['bytes'])


def cpu_baseline_train(trainer, size, budget_s=60.0):
    """Oracle path (plain torch fp32, full (B*H,N,77) maps, 3 launches per LoRA site) on the host cores, B=1."""
    import signal
    from oracle import trainer_ref
    # torch's intra-op pool degrades badly when hundreds of threads fight over the many small ops of a UNet
    # (GroupNorm, SiLU, 77-token GEMMs); 32 threads is what we use and report as `cores`.
    threads = int(os.environ.get('MOS_CPU_BASELINE_THREADS', min(32, os.cpu_count() or 1)))
    torch.set_num_threads(threads)
    times = []

    class _Budget(Exception):
        pass

    def _alarm(signum, frame):
        raise _Budget()

    old = signal.signal(signal.SIGALRM, _alarm)
    signal.alarm(int(os.environ.get('MOS_CPU_BASELINE_TIMEOUT', 240)))   # hard wall-clock bound for the whole leg
    try:
        _log(f'cpu_baseline: building the oracle twin on the host ({threads} threads)')
        twin = trainer_ref.make_reference_twin(trainer, device='cpu', dtype=torch.float32)
        b = synthetic_batch(1, size, 'cpu', 123)
        params = trainer_ref.twin_parameters(twin)
        t_all = time.time()
        for i in range(3):
            for p in params:
                p.grad = None
            t0 = time.time()
            loss = trainer_ref.reference_forward(twin, b['images'], b['prompts'], b['masks'], b['img_masks'])
            loss.backward()
            times.append(time.time() - t0)
            _log(f'cpu_baseline: step {i} took {times[-1]:.2f}s')
            if time.time() - t_all + times[-1] > budget_s:   # bounded sample: stop before the next step would overrun
                break
    except _Budget:
        _log('cpu_baseline: wall-clock bound hit')
    finally:
        signal.alarm(0)
        signal.signal(signal.SIGALRM, old)
    if not times:
        return dict(value=None, unit='images/s', cores=threads, kind='port',
                    sample='oracle step did not finish inside the wall-clock bound')
    best = min(times[1:]) if len(times) > 1 else times[0]
    return dict(value=round(1.0 / best, 5), unit='images/s', cores=torch.get_num_threads(), kind='port',
                sample=f'{len(times)} forward+backward steps at batch 1, {size}x{size}, fp32 torch oracle '
                       f'(oracle/trainer_ref.py), best of the non-first; optimiser step excluded (negligible)')


def run_train(args, rank, world, device):
    from mixofshow.hip import profiler
    from mixofshow.pipelines.train_loop import TrainEngine
    B, size = args.batch, args.size
    _log(f'building EDLoRATrainer (synthetic://{args.preset}) on {device}')
    trainer = build_trainer(args.preset, device)
    _log('trainer ready')
    trainer.unet.train()
    trainer.text_encoder.train()
    engine = TrainEngine(trainer, dict(TRAIN_OPT, optim_g=dict(TRAIN_OPT['optim_g'])), total_iter=1e9,
                         mixed_precision=args.precision, channels_last=args.channels_last)
    batches = [synthetic_batch(B, size, device, 1000 * rank + i) for i in range(2)]
    graphed = False
    if args.graph:
        try:
            engine.enable_graph(batches[0])
            graphed = True
            _log('forward+backward captured in a hipGraph')
        except Exception as e:  # capture is an optimisation of launch overhead only; eager runs the same kernels
            engine._graph = None
            torch.cuda.synchronize()
            import traceback
            _log(f'hipGraph capture failed ({type(e).__name__}); running eager\n' + traceback.format_exc())
    for i in range(args.warmup):
        engine.step(batches[i % 2])
        torch.cuda.synchronize()
        _log(f'warmup step {i} done')
    _sync_barrier(world)
    t0 = time.perf_counter()
    for i in range(args.steps):
        engine.step(batches[i % 2])
    _sync_barrier(world)
    dt = _max_over_ranks(time.perf_counter() - t0, world, device)
    _log(f'timed region: {args.steps} steps in {dt:.3f}s')
    # profiled pass (same workload, same process): per-kernel HIP-event timings of the library kernels
    recs = []
    saved_graph, engine._graph = getattr(engine, '_graph', None), None   # events are recorded at launch: eager pass
    with profiler.profile(recs):
        for i in range(2):
            engine.step(batches[i % 2])
        torch.cuda.synchronize()
    engine._graph = saved_graph
    result = dict(
        metric='edlora_train_images_per_sec_512_sd15', value=round(B * world * args.steps / dt, 4), unit='images/s',
        n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 3),
        higher_is_better=True, scaling='weak', vs_baseline=None,
        dtype={'fp16': 'fp16', 'bf16': 'bf16'}.get(args.precision, 'fp16'), data='synthetic',
        config=dict(workload='BASELINE.json configs[1]: single-concept ED-LoRA tune, SD-1.5 UNet/CLIP/VAE '
                             f'(random init), {size}x{size}, LoRA rank 4 on Attention+CLIPAttention, attn_reg on, '
                             f'batch {B}/GPU', global_batch=B * world, per_gpu_batch=B, image_size=size,
                    parallelism=f'dp{world}', grad_bucket_bytes=engine.bucket.nbytes, preset=args.preset,
                    hipgraph=graphed),
        roofline=roofline_from_profile(recs) if recs else None,
        kernels=[dict(name=r['name'], calls_per_step=r['calls'] / 2, avg_us=round(r['avg_us'], 2),
                      ms_per_step=round(r['total_ms'] / 2, 3),
                      tflops=round(r['flops'] / (r['avg_us'] * 1e-6) / 1e12, 2)) for r in recs[:12]],
        library_kernel_ms_per_step=round(sum(r['total_ms'] for r in recs) / 2, 3))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result['cpu_baseline'] = cpu_baseline_train(trainer, size)
    return result


REGION_PX = [[2, 2, 512, 184], [7, 184, 512, 345], [1, 488, 512, 747]]  # regionally_sample.sh boxes x (1/2, 768/2048)


def build_regional_pipe(preset, device):
    from mixofshow.pipelines.pipeline_regionally_t2iadapter import RegionallyT2IAdapterPipeline
    pipe = RegionallyT2IAdapterPipeline.from_pretrained(f'synthetic://{preset}?seed=0', torch_dtype=torch.float16)
    names = ['<potter1>', '<potter2>', '<hermione1>', '<hermione2>', '<thanos1>', '<thanos2>']
    cfg = {}
    for i, n in enumerate(names):
        toks = [f'<new{16 * i + l}>' for l in range(16)]
        pipe.tokenizer.add_tokens(toks)
        cfg[n] = {'concept_token_ids': [pipe.tokenizer.convert_tokens_to_ids(t) for t in toks],
                  'concept_token_names': toks}
    pipe.text_encoder.resize_token_embeddings(len(pipe.tokenizer))
    pipe.set_new_concept_cfg(cfg)
    return pipe.to(device)


def regional_prompt(height, width):
    ctx = 'three people near the castle, 4K, high quality, high resolution, best quality'
    neg = 'longbody, lowres, bad anatomy, bad hands, missing fingers, extra digit, fewer digits, cropped, worst quality'
    regs = ['a <potter1> <potter2>, in Hogwarts uniform, holding hands, near the castle',
            'a <hermione1> <hermione2>, girl, in Hogwarts uniform, near the castle',
            'a <thanos1> <thanos2>, purple armor, near the castle']
    regions = [(p, neg, [b[0] / height, b[1] / width, b[2] / height, b[3] / width]) for p, b in zip(regs, REGION_PX)]
    return [(ctx, regions)], neg


def run_regional(args, rank, world, device):
    from mixofshow.hip import profiler
    H, W = 512, 768
    pipe = build_regional_pipe(args.preset, device)
    prompt, neg = regional_prompt(H, W)
    latents = torch.randn((1, 4, H // 8, W // 8), generator=torch.manual_seed(14))

    def sample():
        return pipe(prompt=prompt, negative_prompt=[neg], height=H, width=W, num_inference_steps=50,
                    guidance_scale=7.5, latents=latents.clone(), output_type='latent',
                    hipgraph=(args.graph >= 2)).images

    for _ in range(args.warmup):
        sample()
    _sync_barrier(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = sample()
    _sync_barrier(world)
    dt = _max_over_ranks(time.perf_counter() - t0, world, device)
    graphed = bool(getattr(pipe, 'last_call_graphed', False))
    recs = []
    args.graph = 0                                   # HIP events are recorded at launch: profiled pass runs eagerly
    with profiler.profile(recs):
        sample()
        torch.cuda.synchronize()
    return dict(metric='regional_sample_latency_ms_50step_512x768_3regions', value=round(dt / args.steps * 1e3, 2),
                unit='ms', n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 2),
                higher_is_better=False, scaling='weak', vs_baseline=None, dtype='fp16', data='synthetic',
                config=dict(workload='BASELINE.json configs[4]: 3-region (potter/hermione/thanos) 512x768, 50 '
                                     'DPM-Solver++(2M) steps, CFG 7.5, batch 1, SD-1.5 random init, no adapter',
                            replicas=world, preset=args.preset, finite=bool(torch.isfinite(out).all()),
                            hipgraph=graphed),
                roofline=roofline_from_profile(recs) if recs else None,
                kernels=[dict(name=r['name'], calls=r['calls'], avg_us=round(r['avg_us'], 2),
                              total_ms=round(r['total_ms'], 3)) for r in recs[:12]],
                library_kernel_ms_per_sample=round(sum(r['total_ms'] for r in recs), 3))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--mode', default='train', choices=['train', 'regional'])
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--preset', default='sd15')
    ap.add_argument('--precision', default='fp16', choices=['fp16', 'bf16'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--channels-last', type=int, default=0)
    ap.add_argument('--graph', type=int, default=1, help='train: capture fwd+bwd of the step in a hipGraph (default 1); regional: 2 = replay the UNet call '
                         'from a hipGraph (the loop is GPU-bound, no gain measured)')
    args = ap.parse_args()
    from mixofshow.parallel import dp
    rank, world, local = dp.init_distributed()
    assert world == args.gpus or world == 1, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    if not torch.cuda.is_available():
        print('bench.py needs a HIP device (MI355X); there is no CPU fallback', file=sys.stderr)
        sys.exit(2)
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)
    res = run_train(args, rank, world, device) if args.mode == 'train' else run_regional(args, rank, world, device)
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
