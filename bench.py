"""bench.py — headline benchmark of the Mix-of-Show hot path on MI355X (contract: see the task prompt / DESIGN.md).

Default (`--mode train`): BOTH halves of BASELINE.json's metric in ONE JSON line.
  * top level = part 1, ED-LoRA train images/sec @512x512 on SD-1.5, workload = BASELINE.json configs[1] ("Single-concept
    ED-LoRA tune SD-1.5 512x512 rank=4 batch=4 on 1xMI355X"); with --gpus N>1 (one rank per GPU over RCCL, launched by
    torch.distributed.run — or by this script itself when WORLD_SIZE is unset) the same per-GPU batch => weak scaling
    (configs[2]). A "step" is one full optimisation step: VAE encode, CLIP (16 layer-wise prompts / sample), UNet
    forward + backward through the fused HIP attention path, all-reduce of the LoRA+concept-row gradient bucket, AdamW,
    embedding-norm rule. The step runs the way the product runs it by default (train_edlora.py): forward+backward
    replayed from a hipGraph.
  * `regional` = part 2, 50-step 3-region sample latency at 512x768 (configs[4]), single GPU (replicas only), with its
    own roofline and cpu_baseline. Skipped (null) for N>1 and with --no-regional.
`--mode regional` prints part 2 alone; `--mode fusion` times configs[3] (gradient fusion of 14 synthetic ED-LoRAs).

Weights: seeded random init of the exact SD-1.5 architecture, calibrated to O(1) activations (no checkpoints offline);
data: synthetic, resident in HBM before the timed region.

`roofline` describes the library kernel with the largest share of GPU time, timed with HIP events on its launch
stream (mos_profile_*), in a short profiled pass right after the timed region so that `value` is not perturbed by
event recording; `attention_path` is the aggregate: SURVEY 8(d)'s algorithmic attention-path FLOPs per step over the
summed library-kernel time. `cpu_baseline` times the oracle path (oracle/*.py, plain torch fp32 = the "CPU diffusers
reference path" stand-in) on the host cores for a bounded sample.
"""
import argparse
import hashlib
import json
import os
import statistics
import subprocess
import sys
import time

import mos_path  # noqa: F401
import torch

PEAK_MFMA_TFLOPS = 2500.0   # dense fp16/bf16 MFMA, MI355X_MICROARCH.md
PEAK_HBM_GBPS = 8000.0
# algorithmic attention-path FLOPs (QK^T, PV, q/k/v/out projections, LoRA) per trained 512x512 image
# (fwd + bwd) and per CFG-pair UNet call at 512x768 with 3 regions
ATTN_PATH_GFLOP_PER_TRAINED_IMAGE = 604.0
ATTN_PATH_GFLOP_PER_REGIONAL_CALL = 829.5

FINETUNE_CFG = dict(
    text_embedding=dict(enable_tuning=True, lr=1e-3),
    text_encoder=dict(enable_tuning=True, lora_cfg=dict(rank=4, alpha=1.0, where='CLIPAttention'), lr=1e-5),
    unet=dict(enable_tuning=True, lora_cfg=dict(rank=4, alpha=1.0, where='Attention'), lr=1e-4))
TRAIN_OPT = dict(optim_g=dict(type='AdamW', lr=0.0, weight_decay=0.01, betas=[0.9, 0.999]), emb_norm_threshold=0.55)

ROOT = os.path.dirname(os.path.abspath(__file__))
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


# ---- the step fed by the data pipeline (JPEG decode -> PIL transform chain -> collate -> pinned H2D) -----------
JPEG_TRANSFORMS = [dict(type='HumanResizeCropFinalV3', size=512, crop_p=0.5), dict(type='ToTensor'),
                   dict(type='Normalize', mean=[0.5], std=[0.5]), dict(type='ShuffleCaption', keep_token_num=1),
                   dict(type='EnhanceText', enhance_type='human')]     # the shipped recipe's instance_transform chain


def make_jpeg_concept(root, n_images=8, side=768, seed=0):
    """A concept folder in the reference's layout (images/, mask/, caption/ + concept list json) with `n_images` synthetic
    photographs: smooth random fields (so that JPEG decode cost is that of a real photo, not of white noise), a centred
    elliptical subject mask, one caption per image. Returns the path of the concept list."""
    import numpy as np
    from PIL import Image
    rng = np.random.RandomState(seed)
    img_dir, mask_dir, cap_dir = (os.path.join(root, d) for d in ('image', 'mask', 'caption'))
    for d in (img_dir, mask_dir, cap_dir):
        os.makedirs(d, exist_ok=True)
    yy, xx = np.mgrid[0:side, 0:side].astype(np.float32) / side
    for i in range(n_images):
        low = rng.rand(12, 12, 3).astype(np.float32)
        img = np.asarray(Image.fromarray((low * 255).astype(np.uint8)).resize((side, side), Image.BICUBIC), dtype=np.float32)
        img = np.clip(img + rng.randn(side, side, 3) * 6.0, 0, 255).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(img_dir, f'{i:03d}.jpg'), quality=92)
        cx, cy, rx, ry = 0.5 + 0.1 * rng.randn(), 0.5 + 0.05 * rng.randn(), 0.22, 0.38
        mask = ((((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2) <= 1.0).astype(np.uint8) * 255
        Image.fromarray(mask, mode='L').save(os.path.join(mask_dir, f'{i:03d}.png'))
        with open(os.path.join(cap_dir, f'{i:03d}.txt'), 'w') as f:
            f.write('<TOK>, standing in the park, looking at the camera, 4K, high quality\n')
    clist = os.path.join(root, 'concept.json')
    with open(clist, 'w') as f:
        json.dump([dict(instance_prompt='<TOK>', instance_data_dir=img_dir, caption_dir=cap_dir, mask_dir=mask_dir)], f)
    return clist


def jpeg_loader(root, batch, size, workers, seed=0, n_images=8):
    """DataLoader over LoraDataset (mixofshow.data: the reference's dataset + transform chain) on a synthetic JPEG concept."""
    from mixofshow.data.lora_dataset import LoraDataset
    tf = [dict(t, size=size) if t['type'] == 'HumanResizeCropFinalV3' else dict(t) for t in JPEG_TRANSFORMS]
    ds = LoraDataset(dict(name='LoraDataset', concept_list=make_jpeg_concept(root, n_images, seed=seed), use_caption=True,
                          use_mask=True, replace_mapping={'<TOK>': '<potter1> <potter2>'}, dataset_enlarge_ratio=100000,
                          instance_transform=tf))
    kw = dict(num_workers=workers, persistent_workers=True, prefetch_factor=4) if workers > 0 else {}
    return torch.utils.data.DataLoader(ds, batch_size=batch, shuffle=True, drop_last=True,
                                       pin_memory=torch.cuda.is_available(), **kw)


# ---- which box, at which clocks (a line must say what it ran on) ---------------------------------------
def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _drm_device_dir(index=0):
    """sysfs directory of the index-th amdgpu render device (cards are not always numbered from 0)."""
    import glob
    cards = sorted(d for d in glob.glob('/sys/class/drm/card[0-9]*/device') if (_read(os.path.join(d, 'vendor')) or '') == '0x1002')
    return cards[index] if index < len(cards) else None


def _hwmon(dev, name):
    import glob
    for p in glob.glob(os.path.join(dev, 'hwmon', 'hwmon*', name)) if dev else []:
        v = _read(p)
        if v is not None:
            try:
                return int(v)
            except ValueError:
                return None
    return None


def box_info(local=0):
    """Fingerprint of the box (hostname + GPU unique id, hashed), the device as torch names it, and the power cap: read from sysfs
    once, outside every timed region. Every field is best-effort (null where the box does not expose it)."""
    import socket
    dev = _drm_device_dir(local)
    uid = _read(os.path.join(dev, 'unique_id')) if dev else None
    fp = hashlib.sha256(f'{socket.gethostname()}|{uid}'.encode()).hexdigest()[:10]
    out = dict(id=fp, gpu_uid=(uid or '')[-8:] or None)
    try:
        pr = torch.cuda.get_device_properties(local)
        out.update(name=pr.name, cus=pr.multi_processor_count, max_sclk_mhz=round(pr.clock_rate / 1e3),
                   hbm_gib=round(pr.total_memory / 2 ** 30))
    except Exception:
        pass
    cap = _hwmon(dev, 'power1_cap')
    out['power_cap_w'] = round(cap / 1e6) if cap else None
    out['rocm'] = (_read('/opt/rocm/.info/version') or '').split('-')[0] or None
    return out


class ClockSampler:
    """Shader clock / memory clock / board power during a timed region, read from the amdgpu hwmon files every `period` s by
    a daemon thread (two small sysfs reads; the graph-replayed step does not depend on the host thread). min / mean / max."""

    def __init__(self, local=0, period=0.05):
        import threading
        self.dev, self.period, self.rows, self._stop = _drm_device_dir(local), period, [], threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            self.rows.append((_hwmon(self.dev, 'freq1_input'), _hwmon(self.dev, 'freq2_input'),
                              _hwmon(self.dev, 'power1_average') or _hwmon(self.dev, 'power1_input')))
            self._stop.wait(self.period)

    def __enter__(self):
        if self.dev:
            self._t.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self.dev:
            self._t.join(timeout=1.0)

    def summary(self):
        def col(i, scale):
            v = [r[i] / scale for r in self.rows if r[i]]
            return [round(min(v)), round(sum(v) / len(v)), round(max(v))] if v else None
        return dict(samples=len(self.rows), sclk_mhz_min_mean_max=col(0, 1e6), mclk_mhz_min_mean_max=col(1, 1e6),
                    power_w_min_mean_max=col(2, 1e6))


# ---- roofline helpers ---------------------------------------------------------------------------------------------
# the translation unit, shared header and build recipe of every kernel profiles/pmc_traffic.json covers (the attention and
# regional kernels all live in mos_attn.hip): the PMC numbers stay valid exactly as long as these files are unchanged
PMC_SOURCE_FILES = ('mos_attn.hip', 'mos_common.h', 'build.sh')


def kernel_source_fingerprint(files=PMC_SOURCE_FILES):
    """sha256 (16 hex) over the sources + build recipe of the kernels the PMC passes profiled (files=None: all of csrc/)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'mix-of-show_amd', 'csrc')
    for fn in sorted(os.listdir(d)):
        if fn.endswith(('.hip', '.h', '.inc', '.sh')) and (files is None or fn in files):
            with open(os.path.join(d, fn), 'rb') as f:
                h.update(fn.encode() + b'\0' + f.read())
    return h.hexdigest()[:16]


def _pmc_traffic(kernel_name, path=None):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json,
    written by tools/pmc_traffic.py: FETCH_SIZE / WRITE_SIZE collected in separate passes, gfx950-corrected) — but
    ONLY if that file was measured on the kernel sources that are in the tree now (fingerprint match); a number
    profiled on other code is stale and reported as null."""
    path = path or os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        with open(path) as f:
            d = json.load(f)
        if d.get('source_sha16') != kernel_source_fingerprint():
            return None
        ent = d.get('kernels', {}).get(kernel_name)
        return float(ent['total']) if ent else None
    except (OSError, ValueError, KeyError, TypeError):
        return None


def parity_figures(path=None):
    """The measured denoised-latent / epsilon errors of the HIP path against the oracle (exact attention and the reference's
    fp16 arithmetic; teacher-forced and free-running; max-abs AND rms) as the GPU parity tests wrote them
    (tests/test_gpu_end_to_end.py::_record_parity -> profiles/parity_latents.json), so that the 1e-3 claim is auditable from
    the bench line. `stale` = the kernel sources changed since they were measured."""
    path = path or os.path.join(ROOT, 'profiles', 'parity_latents.json')
    try:
        with open(path) as f:
            d = json.load(f)
        return dict(source='profiles/parity_latents.json', stale=d.get('kernel_source_sha16_all') != kernel_source_fingerprint(None),
                    tolerance_north_star=d.get('tolerance_north_star'), normalisation=d.get('normalisation'), cases=d.get('cases'))
    except (OSError, ValueError):
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


def dominant_by_kernel_name(records, per):
    """The per-(kernel, shape) record with the largest time is what `roofline` describes; the dominant kernel BY NAME (all
    shapes of one kernel summed) can be another one (round 2: dK/dV d40 by record, conv3x3 by name): reported beside it."""
    agg = {}
    for r in records:
        k = r['name'].split(' ')[0]
        a = agg.setdefault(k, dict(ms=0.0, flops=0.0, bytes=0.0, calls=0))
        a['ms'] += r['total_ms']
        a['flops'] += r['flops'] * r['calls']
        a['bytes'] += r['bytes'] * r['calls']
        a['calls'] += r['calls']
    lib_ms = sum(a['ms'] for a in agg.values())
    out = []
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])[:12]:
        sec = a['ms'] * 1e-3
        out.append(dict(kernel=k, ms=round(a['ms'] / per, 3), launches=a['calls'] / per,
                        share_of_library_gpu_time=round(a['ms'] / max(1e-9, lib_ms), 4),
                        tflops=round(a['flops'] / max(1e-12, sec) / 1e12, 2), gbps=round(a['bytes'] / max(1e-12, sec) / 1e9, 1),
                        frac_of_mfma_peak=round(a['flops'] / max(1e-12, sec) / 1e12 / PEAK_MFMA_TFLOPS, 5)))
    return out


# Algorithmic FLOPs of one ED-LoRA training step per image at 512^2 that are NOT seen by the library's profiler
# (SURVEY 8(d) level table): the GEGLU feed-forward GEMMs (hipBLASLt): 153.5 GFLOP per sample-forward, x2 (forward + dX;
# frozen weights: no dW).
FF_GEMM_GFLOP_PER_TRAINED_IMAGE = 2 * 153.5


def whole_step_utilisation(recs, per, images, step_ms):
    """Algorithmic FLOPs of everything timed by the library's own profiler (attention, projections, convolutions incl. the
    VAE encoder and the CLIP tower) + the analytic feed-forward GEMM FLOPs, over the measured step time. MIOpen's small
    convolutions (16x16 / 8x8 levels, stride 2, conv_in / conv_out) are not counted: a lower bound."""
    lib = sum(r['flops'] * r['calls'] for r in recs) / per / 1e12
    ff = FF_GEMM_GFLOP_PER_TRAINED_IMAGE * images / 1e3
    tf = (lib + ff) / max(1e-12, step_ms * 1e-3)
    return dict(library_kernels_tflop=round(lib, 3), ff_gemm_tflop_analytic=round(ff, 3), tflop_per_step=round(lib + ff, 3),
                achieved_tflops=round(tf, 1), frac_of_mfma_peak=round(tf / PEAK_MFMA_TFLOPS, 4))


ATTENTION_PATH_KERNELS = ('attn_', 'region_attn', 'gemm_nt', 'lora_')


def _tuning_switches():
    """The host-side dispatch switches in effect (defaults unless overridden in the environment; the LIBRARY reads no
    environment variable since round 5)."""
    from mixofshow.hip import functional as F_hip
    return dict(conv3x3_min_pixels=F_hip._conv_min_pixels,
                fuse_add_layernorm=bool(F_hip._fuse_add_ln), fuse_groupnorm_skip_grad=bool(F_hip._fuse_gn_res),
                batched_time_projections=os.environ.get('MOS_BATCH_TEMB', '1') != '0',
                ff2_residual_epilogue=bool(F_hip._ff2_own))


def attention_path_aggregate(gflop_per_unit, units, recs, per):
    """algorithmic attention-path FLOPs of the timed unit over the summed time of the library kernels ON
    that path (attention, fused projections incl. the 1x1 convs routed through gemm_nt, LoRA gradients/packing). The
    (f).1 kernels the library also runs (conv3x3, GroupNorm, LayerNorm, GEGLU, softmax) are reported beside it, not in it."""
    on_path = sum(r['total_ms'] for r in recs if r['name'].startswith(ATTENTION_PATH_KERNELS)) / per
    lib_ms = sum(r['total_ms'] for r in recs) / per
    tflop = gflop_per_unit * units / 1e3
    achieved = tflop / max(1e-12, on_path * 1e-3)
    return dict(algorithmic_tflop=round(tflop, 4), attention_path_kernel_ms=round(on_path, 3),
                other_library_kernel_ms=round(lib_ms - on_path, 3), achieved_tflops=round(achieved, 2),
                frac_of_mfma_peak=round(achieved / PEAK_MFMA_TFLOPS, 5))


def _kernel_table(recs, per, n=14):
    return [dict(name=r['name'], calls=r['calls'] / per, avg_us=round(r['avg_us'], 2),
                 ms=round(r['total_ms'] / per, 3), tflops=round(r['flops'] / (r['avg_us'] * 1e-6) / 1e12, 2),
                 gbps=round(r['bytes'] / (r['avg_us'] * 1e-6) / 1e9, 1)) for r in recs[:n]]


# ---- CPU baselines (oracle path on the host cores; bounded samples) ------------------------------------------------
class _Budget(Exception):
    pass


def _with_alarm(seconds, fn):
    import signal

    def _alarm(signum, frame):
        raise _Budget()

    old = signal.signal(signal.SIGALRM, _alarm)
    signal.alarm(int(seconds))
    try:
        return fn()
    except _Budget:
        _log('cpu_baseline: wall-clock bound hit')
        return None
    finally:
        signal.alarm(0)
        signal.signal(signal.SIGALRM, old)


def _cpu_threads():
    # torch's intra-op pool degrades when hundreds of threads fight over the many small ops of a UNet (GroupNorm, SiLU,
    # 77-token GEMMs): 32 threads are used; `cores` reports the threads USED, `host_cores` what the box has.
    threads = int(os.environ.get('MOS_CPU_BASELINE_THREADS', min(32, os.cpu_count() or 1)))
    torch.set_num_threads(threads)
    return threads


def cpu_baseline_train(trainer, size, budget_s=30.0):
    """Oracle path (plain torch fp32, full (B*H,N,77) maps, 3 launches per LoRA site) on the host cores, batch 1:
    1 warm-up + up to 4 timed forward+backward steps, median."""
    from oracle import trainer_ref
    threads = _cpu_threads()
    times = []

    def run():
        _log(f'cpu_baseline(train): building the oracle twin on the host ({threads} threads of {os.cpu_count()})')
        twin = trainer_ref.make_reference_twin(trainer, device='cpu', dtype=torch.float32)
        b = synthetic_batch(1, size, 'cpu', 123)
        params = trainer_ref.twin_parameters(twin)
        t_all = time.time()
        for i in range(5):
            for p in params:
                p.grad = None
            t0 = time.time()
            loss = trainer_ref.reference_forward(twin, b['images'], b['prompts'], b['masks'], b['img_masks'])
            loss.backward()
            dt = time.time() - t0
            _log(f'cpu_baseline(train): step {i} took {dt:.2f}s')
            if i > 0:
                times.append(dt)
            if time.time() - t_all + dt > budget_s:     # bounded sample: stop before the next step would overrun
                break

    _with_alarm(int(os.environ.get('MOS_CPU_BASELINE_TIMEOUT', 240)), run)
    if not times:
        return dict(value=None, unit='images/s', cores=threads, host_cores=os.cpu_count(), kind='port',
                    sample='oracle step did not finish inside the wall-clock bound')
    med = statistics.median(times)
    return dict(value=round(1.0 / med, 5), unit='images/s', cores=threads, host_cores=os.cpu_count(), kind='port',
                sample=f'median of {len(times)} forward+backward steps after 1 warm-up, batch 1, {size}x{size}, fp32 torch '
                       'oracle (oracle/trainer_ref.py: full probability maps, 3-GEMM LoRA); optimiser step excluded')


def cpu_baseline_regional(preset, H, W, budget_s=28.0):
    """One regional UNet call (CFG pair, oracle region processors, fp32) on the host cores, extrapolated x50."""
    from oracle import region_ref
    threads = _cpu_threads()
    times = []

    def run():
        pipe = build_regional_pipe(preset, torch.device('cpu'), dtype=torch.float32)
        region_ref.install_region_processors_ref(pipe.unet)
        prompt, neg = regional_prompt(H, W)
        with torch.no_grad():
            emb, region_list = pipe._encode_region_prompt(prompt, pipe.new_concept_cfg, 'cpu', 1, True, [neg], height=H,
                                                          width=W)
            cak = {'region_list': region_list, 'height': H, 'width': W}
            x = torch.randn((2, 4, H // 8, W // 8), generator=torch.manual_seed(14))
            t_all = time.time()
            for i in range(4):
                t0 = time.time()
                pipe.unet(x, torch.tensor(500), encoder_hidden_states=emb, cross_attention_kwargs=cak)
                dt = time.time() - t0
                _log(f'cpu_baseline(regional): UNet call {i} took {dt:.2f}s')
                if i > 0:
                    times.append(dt)
                if time.time() - t_all + dt > budget_s:
                    break

    _with_alarm(int(os.environ.get('MOS_CPU_BASELINE_TIMEOUT', 240)), run)
    if not times:
        return dict(value=None, unit='ms', cores=threads, host_cores=os.cpu_count(), kind='port',
                    sample='oracle UNet call did not finish inside the wall-clock bound')
    med = statistics.median(times)
    return dict(value=round(med * 50 * 1e3, 1), unit='ms', cores=threads, host_cores=os.cpu_count(), kind='port',
                sample=f'50 x median of {len(times)} regional UNet calls (CFG pair, {H}x{W}, 3 regions) after 1 warm-up, fp32 '
                       'torch oracle (oracle/region_ref.py); scheduler / text encoding excluded (negligible)')


# ---- part 1: training ----------------------------------------------------------------------------------------------
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
    next_batch, data_info, tmp_dir = (lambda i: batches[i % 2]), None, None
    if args.data == 'jpeg':
        # every timed step consumes a batch that was decoded from JPEG, pushed through the reference's
        # transform chain by DataLoader workers, collated, pinned and copied to the device inside the timed region
        import tempfile
        tmp_dir = tempfile.TemporaryDirectory(prefix='mos_jpeg_')
        loader = jpeg_loader(tmp_dir.name, B, size, args.workers, seed=rank)
        it = iter(loader)
        batches = [next(it) for _ in range(2)]                  # also spins the workers up
        n_probe = 8
        t_l = time.perf_counter()
        for _ in range(n_probe):                                # loader-only throughput (no training step consuming)
            next(it)
        loader_ips = B * n_probe / (time.perf_counter() - t_l)
        next_batch = lambda i: next(it)
        data_info = dict(source='synthetic JPEG concept (8 photos 768x768 q92, masks, captions)', workers=args.workers,
                         transforms=[t['type'] for t in JPEG_TRANSFORMS], pin_memory=bool(torch.cuda.is_available()),
                         loader_only_images_per_sec=round(loader_ips, 2))
        _log(f'data pipeline ready: {args.workers} workers, loader alone {loader_ips:.1f} images/s')
    graphed, capture_s = False, None
    if args.graph:
        try:
            t_cap = time.perf_counter()
            engine.enable_graph(batches[0])
            torch.cuda.synchronize()
            capture_s = round(time.perf_counter() - t_cap, 2)
            graphed = True
            _log(f'forward+backward captured in a hipGraph ({capture_s}s)')
        except Exception as e:  # capture is an optimisation of launch overhead only; eager runs the same kernels
            engine._graph = None
            torch.cuda.synchronize()
            import traceback
            _log(f'hipGraph capture failed ({type(e).__name__}); running eager\n' + traceback.format_exc())
    if data_info is not None:
        def to_dev(b):
            return {k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in b.items()}
        batches = [to_dev(b) for b in batches]                  # the eager profiled pass below takes device tensors
        if not graphed:                                         # (a replayed step copies pinned host batches straight into its
            next_batch = lambda i: to_dev(next(it))             #  static input buffers: one H2D, no staging allocation)
    for i in range(args.warmup):
        engine.step(next_batch(i))
        torch.cuda.synchronize()
        _log(f'warmup step {i} done')
    _sync_barrier(world)
    with ClockSampler(device.index or 0) as clocks:
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(args.steps):
            engine.step(next_batch(i))
            marks[i + 1].record()                   # device-side step boundaries: no host sync inside the region
        _sync_barrier(world)
        dt = _max_over_ranks(time.perf_counter() - t0, world, device)
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    spread = [round(per_step[0], 2), round(statistics.median(per_step), 2), round(per_step[-1], 2)] if per_step else None
    # what the HOST needs per step, outside the timed region: with the device idle at the start of a step nothing the host does
    # can block on a full launch queue (inside a pipelined loop a host that runs ahead waits in hipGraphLaunch, so its loop time
    # only mirrors the device's). A replayed step needs ~5 ms of host time against ~33 ms of device time (tools/host_step_breakdown.py)
    host_ms = 0.0
    for i in range(5):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        engine.step(next_batch(i))
        host_ms += (time.perf_counter() - t1) * 1e3 / 5
    torch.cuda.synchronize()
    _log(f'timed region: {args.steps} steps in {dt:.3f}s; device-side step ms min/median/max {spread}; host needs '
         f'{host_ms:.2f} ms/step (device idle at step start); clocks {clocks.summary()}')
    allreduce = None
    if world > 1:
        # the one collective of the path, on its own -- the flat fp32 gradient bucket over RCCL/xGMI (latency-bound at
        # 4.5 MB). Timed on a scratch copy so the optimiser state of the run is untouched; bus bandwidth = 2(N-1)/N x bytes / time.
        scratch = engine.bucket.flat.clone()
        for _ in range(5):
            torch.distributed.all_reduce(scratch)
        _sync_barrier(world)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n_ar = 50
        e0.record()
        for _ in range(n_ar):
            torch.distributed.all_reduce(scratch)
        e1.record()
        torch.cuda.synchronize()
        us = _max_over_ranks(e0.elapsed_time(e1) * 1e3 / n_ar, world, device)
        allreduce = dict(bytes=engine.bucket.nbytes, us=round(us, 1),
                         busbw_gbps=round(2.0 * (world - 1) / world * engine.bucket.nbytes / (us * 1e-6) / 1e9, 2),
                         peak_gbps_per_gpu=7 * 153.0, backend=torch.distributed.get_backend())
        _log(f'gradient-bucket all-reduce: {allreduce}')
        del scratch
    # profiled pass (same workload, same process): per-kernel HIP-event timings of the library kernels
    recs = []
    saved_graph, engine._graph = getattr(engine, '_graph', None), None   # events are recorded at launch: eager pass
    with profiler.profile(recs):
        for i in range(2):
            engine.step(batches[i % 2])
        torch.cuda.synchronize()
    engine._graph = saved_graph
    lib_ms = sum(r['total_ms'] for r in recs) / 2
    result = dict(
        metric='edlora_train_images_per_sec_512_sd15', value=round(B * world * args.steps / dt, 4), unit='images/s',
        n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 3),
        higher_is_better=True, scaling='weak', vs_baseline=None,
        dtype={'fp16': 'fp16', 'bf16': 'bf16'}.get(args.precision, 'fp16'),
        data='synthetic' if data_info is None else 'synthetic-jpeg (decode + transforms + H2D inside the timed region)',
        config=dict(workload='BASELINE.json configs[1]: single-concept ED-LoRA tune, SD-1.5 UNet/CLIP/VAE '
                             f'(random init, calibrated), {size}x{size}, LoRA rank 4 on Attention+CLIPAttention, '
                             f'attn_reg on, batch {B}/GPU', global_batch=B * world, per_gpu_batch=B, image_size=size,
                    parallelism=f'dp{world}', grad_bucket_bytes=engine.bucket.nbytes, preset=args.preset,
                    hipgraph=graphed, graph_capture_s=capture_s, channels_last=bool(args.channels_last),
                    host_cores=os.cpu_count(), kernel_source_sha16=kernel_source_fingerprint(),
                    box=box_info(device.index or 0), clocks_timed_region=clocks.summary(), allreduce=allreduce,
                    host_need_ms_per_step=round(host_ms, 2),
                    **_tuning_switches()),
        roofline=roofline_from_profile(recs) if recs else None,
        attention_path=attention_path_aggregate(ATTN_PATH_GFLOP_PER_TRAINED_IMAGE, B, recs, 2) if recs else None,
        dominant_kernels_by_name=dominant_by_kernel_name(recs, 2) if recs else None,
        whole_step=whole_step_utilisation(recs, 2, B, dt / args.steps * 1e3) if recs else None,
        kernels=_kernel_table(recs, 2), library_kernel_ms_per_step=round(lib_ms, 3), step_ms_spread=spread)
    if data_info is not None:
        result['data_pipeline'] = data_info
        del it, loader
        tmp_dir.cleanup()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result['cpu_baseline'] = cpu_baseline_train(trainer, size)
    del engine, trainer
    torch.cuda.empty_cache()
    return result


# ---- part 2: regional sampling ---------------------------------------------------------------------------------------
REGION_PX = [[2, 2, 512, 184], [7, 184, 512, 345], [1, 488, 512, 747]]  # regionally_sample.sh boxes x (1/2, 768/2048)
REGION_PX_SHIPPED = [[4, 6, 1024, 490], [14, 490, 1024, 920], [2, 1302, 1024, 1992]]   # regionally_sample.sh:64-74, at 1024x2048


def region_px(height, width):
    """Pixel boxes [h0, w0, h1, w1] of the three regions: the reference's shipped example as is at 1024x2048
    (regionally_sample.sh:52-90, N = 32768 at level 0), BASELINE configs[4]'s scaled boxes at 512x768, scaled otherwise."""
    if (height, width) == (1024, 2048):
        return REGION_PX_SHIPPED
    if (height, width) == (512, 768):
        return REGION_PX
    return [[round(b[0] * height / 1024), round(b[1] * width / 2048), round(b[2] * height / 1024), round(b[3] * width / 2048)]
            for b in REGION_PX_SHIPPED]


def regional_attention_gflop(height, width, n_regions_px):
    """SURVEY 8(d)'s attention-path count (QK^T, PV, q/k/v/out projections of the 16 transformer blocks, region K/V and boxed
    attention) for one CFG-pair regional UNet call at any size: 829.5 GFLOP at 512x768 with the three configs[4] boxes."""
    B, T = 2, 77
    levels = [(320, 1, 5), (640, 2, 5), (1280, 4, 5), (1280, 8, 1)]
    area = sum(max(0, b[2] - b[0]) * max(0, b[3] - b[1]) for b in n_regions_px) / float(height * width)
    R = len(n_regions_px)
    fl = 0.0
    for C, down, blocks in levels:
        N = (height // (8 * down)) * (width // (8 * down))
        self_core = 4.0 * N * N * C
        self_proj = 2.0 * N * C * C * 4
        cross_proj = 2.0 * N * C * C * 2
        cross_kv = 2.0 * T * 768 * C * 2
        cross_core = 4.0 * N * T * C
        region = R * cross_kv + 4.0 * (area * N) * T * C
        fl += blocks * B * (self_core + self_proj + cross_proj + cross_kv + cross_core + region)
    return fl / 1e9


def build_regional_pipe(preset, device, dtype=torch.float16):
    from mixofshow.pipelines.pipeline_regionally_t2iadapter import RegionallyT2IAdapterPipeline
    pipe = RegionallyT2IAdapterPipeline.from_pretrained(f'synthetic://{preset}?seed=0', torch_dtype=dtype)
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
    regions = [(p, neg, [b[0] / height, b[1] / width, b[2] / height, b[3] / width]) for p, b in zip(regs, region_px(height, width))]
    return [(ctx, regions)], neg


def synthetic_adapter_states(pipe, height, width, device, dtype, seed=15):
    """SURVEY 8(d) cfg #5: the reference cannot run without an adapter input (pipeline_regionally_t2iadapter.py:484); no
    T2I-Adapter weights exist offline, so the four feature levels an adapter would produce are seeded synthetic tensors
    (N(0, 0.05^2)), pushed through the product's region-weight rule (`_adapter_states`, reference :488-542) with the
    regionally_sample.sh style weights: keypose 1.0 everywhere, 0.6 inside the first region box."""
    chans = pipe.unet.config.block_out_channels
    g = torch.Generator().manual_seed(seed)
    feats = [torch.randn((1, c, height // (8 << min(i, len(chans) - 1)), width // (8 << min(i, len(chans) - 1))), generator=g)
             .mul_(0.05).to(device, dtype) for i, c in enumerate(chans)]

    class _Fixed(torch.nn.Module):          # stands in for the adapter network: returns the seeded feature list
        dtype = feats[0].dtype

        def forward(self, x):
            return feats

    b = region_px(height, width)[0]
    return pipe._adapter_states(_Fixed(), torch.zeros(1, 3, height, width), 1.0, f'[{b[0]}, {b[1]}, {b[2]}, {b[3]}]-0.6',
                                height, width)


def synthetic_keypose_adapter(pipe, height, width, device, dtype, seed=16):
    """A real `T2IAdapter` network (diffusers full_adapter layout, restated in the pipeline module) with seeded random weights,
    scaled so that its level-0 features have std 0.05 (no adapter checkpoints exist offline), and a seeded synthetic pose
    image: the adapter FORWARD runs inside every timed call, as in the reference (:474-546)."""
    from mixofshow.pipelines.pipeline_regionally_t2iadapter import T2IAdapter
    torch.manual_seed(seed)
    ad = T2IAdapter(channels=tuple(pipe.unet.config.block_out_channels)).to(device, dtype).eval()
    g = torch.Generator().manual_seed(seed)
    low = torch.rand((1, 3, height // 32, width // 32), generator=g)
    image = torch.nn.functional.interpolate(low, size=(height, width), mode='bilinear').to(device)    # (1,3,H,W) in [0,1]
    with torch.no_grad():
        f0 = ad(image.to(dtype))[0].float().std().item()
        ad.adapter.conv_in.weight.mul_(0.05 / max(f0, 1e-6))
        ad.adapter.conv_in.bias.mul_(0.05 / max(f0, 1e-6))
        stds = [round(f.float().std().item(), 4) for f in ad(image.to(dtype))]
    return ad, image, stds


def run_regional(args, rank, world, device, steps=None, warmup=None):
    """`value` = latency of the call AS THE REFERENCE'S CALL RETURNS IT (regionally_controlable_sampling.py:40-52,
    pipeline_regionally_t2iadapter.py:474-546, 582-595): T2I-Adapter forward on a pose image + region-weight rule, 50
    regional UNet steps, VAE decode at 512x768, PIL conversion on the host. `value_ms_latent` = the same 50 steps with
    precomputed adapter states and `output_type='latent'` (the round-1..3 line) for continuity."""
    from mixofshow.hip import profiler
    H, W = int(getattr(args, 'height', 512)), int(getattr(args, 'width', 768))
    steps = steps or args.steps
    warmup = args.warmup if warmup is None else warmup
    pipe = build_regional_pipe(args.preset, device)
    if args.channels_last:
        pipe.unet.to(memory_format=torch.channels_last)
        pipe.vae.to(memory_format=torch.channels_last)
    prompt, neg = regional_prompt(H, W)
    latents = torch.randn((1, 4, H // 8, W // 8), generator=torch.manual_seed(14))
    graph = None if args.regional_graph < 0 else bool(args.regional_graph)
    adapter_states = synthetic_adapter_states(pipe, H, W, device, torch.float16)
    pipe.keypose_adapter, pose, adapter_stds = synthetic_keypose_adapter(pipe, H, W, device, torch.float16)
    b = region_px(H, W)[0]
    region_w = f'[{b[0]}, {b[1]}, {b[2]}, {b[3]}]-0.6'

    def sample(g, image_out=True):
        if image_out:
            return pipe(prompt=prompt, negative_prompt=[neg], height=H, width=W, num_inference_steps=50,
                        guidance_scale=7.5, latents=latents.clone(), output_type='pil', hipgraph=g,
                        keypose_adapter_input=pose, keypose_adaptor_weight=1.0, region_keypose_adaptor_weight=region_w).images
        return pipe(prompt=prompt, negative_prompt=[neg], height=H, width=W, num_inference_steps=50,
                    guidance_scale=7.5, latents=latents.clone(), output_type='latent', hipgraph=g,
                    adapter_states=adapter_states).images

    def timed(n, **kw):
        _sync_barrier(world)
        t0 = time.perf_counter()
        for _ in range(n):
            o = sample(graph, **kw)
        _sync_barrier(world)
        return _max_over_ranks(time.perf_counter() - t0, world, device), o

    cold_s, _ = timed(1)                             # first call of this layout: eager step 0 + capture + 48 replays (+ MIOpen finds)
    for _ in range(max(0, warmup - 1)):
        sample(graph)
    dt, images = timed(steps)
    sample(graph, image_out=False)
    dt_lat, out = timed(steps, image_out=False)
    graphed = bool(getattr(pipe, 'last_call_graphed', False))
    eager_steps = int(getattr(pipe, 'last_call_replay_from', 1)) if graphed else 50     # (read before the profiled eager pass)
    import numpy as np
    img_ok = (len(images) == 1 and images[0].size == (W, H) and bool(np.isfinite(np.asarray(images[0], dtype=np.float32)).all()))
    recs = []
    with profiler.profile(recs):                     # HIP events are recorded at launch: profiled pass runs eagerly
        sample(False)
        torch.cuda.synchronize()
    lib_ms = sum(r['total_ms'] for r in recs)
    res = dict(metric=f'regional_sample_latency_ms_50step_{H}x{W}_3regions', value=round(dt / steps * 1e3, 2),
               unit='ms', n_gpus=world, steps=steps, warmup=warmup, ms_per_step=round(dt / steps * 1e3, 2),
               value_ms_image=round(dt / steps * 1e3, 2), value_ms_latent=round(dt_lat / steps * 1e3, 2),
               cold_call_ms=round(cold_s * 1e3, 1),
               higher_is_better=False, scaling='weak', vs_baseline=None, dtype='fp16', data='synthetic',
               config=dict(workload=('BASELINE.json configs[4]' if (H, W) == (512, 768) else "the reference's shipped example "
                                     '(regionally_sample.sh:52-90)' if (H, W) == (1024, 2048) else 'regional sample') +
                                    f': 3-region (potter/hermione/thanos) {H}x{W}, 50 '
                                    'DPM-Solver++(2M) steps, CFG 7.5, batch 1, SD-1.5 random init (calibrated); value / '
                                    'value_ms_image: T2I-Adapter network forward on a seeded pose image + region weight, VAE '
                                    'decode, PIL out (the reference call); value_ms_latent: precomputed seeded adapter states, '
                                    'latent out (SURVEY 8(d) cfg #5)', height=H, width=W, region_boxes_px=region_px(H, W),
                           replicas=world, preset=args.preset, finite=bool(torch.isfinite(out).all()) and img_ok,
                           hipgraph=graphed, graph_reused_across_calls=graphed,
                           steady_state_eager_steps=eager_steps,
                           timed_calls='steady state: same layout as the warm-up calls, UNet graph captured there and replayed '
                                       '(all 50 steps once the source K/V buffers are refilled explicitly); '
                                       'cold_call_ms = first call of the layout (eager step 0 + capture)',
                           adapter_feature_std=adapter_stds,
                           channels_last=bool(args.channels_last), host_cores=os.cpu_count(),
                           box=box_info(device.index or 0), **_tuning_switches()),
               launches_per_unet_call=round(sum(r['calls'] for r in recs) / 50.0, 1) if recs else None,
               roofline=roofline_from_profile(recs) if recs else None,
               attention_path=attention_path_aggregate(ATTN_PATH_GFLOP_PER_REGIONAL_CALL if (H, W) == (512, 768) else
                                                       regional_attention_gflop(H, W, region_px(H, W)), 50, recs, 1) if recs else None,
               dominant_kernels_by_name=dominant_by_kernel_name(recs, 1) if recs else None,
               kernels=_kernel_table(recs, 1), library_kernel_ms_per_sample=round(lib_ms, 3))
    del pipe
    torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_cpu_baseline and H * W <= 512 * 768:     # (a 1024x2048 oracle call takes minutes)
        res['cpu_baseline'] = cpu_baseline_regional(args.preset, H, W)
    return res


# ---- configs[3]: gradient fusion of 14 synthetic ED-LoRAs ------------------------------------------------------------
def synthetic_edlora_checkpoints(preset, n, out_dir):
    """SURVEY 8(d) cfg #4: n synthetic ED-LoRA dicts (keys per App. C, seeds 0..n-1, lora_up ~ N(0, 0.02^2),
    lora_down kaiming, alphas 1.0), two concept words each. Returns the fusion json path."""
    tr = build_trainer(preset, torch.device('cpu'))
    template = tr.delta_state_dict()
    words = ['potter', 'hermione', 'thanos', 'hinton', 'lecun', 'bengio', 'catA', 'dogA', 'dogB', 'chair', 'table', 'vase',
             'rock', 'pyramid', 'anime', 'style']
    entries = []
    for i in range(n):
        g = torch.Generator().manual_seed(i)
        d = {'new_concept_embedding': {}, 'text_encoder': {}, 'unet': {}}
        a, b = f'<{words[i % len(words)]}{i}a>', f'<{words[i % len(words)]}{i}b>'
        for name in (a, b):
            d['new_concept_embedding'][name] = torch.randn(16, tr.concept_embedding.shape[1], generator=g) * 0.013
        for part in ('text_encoder', 'unet'):
            for k, v in template[part].items():
                if 'lora_up' in k:
                    d[part][k] = torch.randn(v.shape, generator=g) * 0.02
                else:
                    bound = 1.0 / (v.shape[1] ** 0.5)
                    d[part][k] = (torch.rand(v.shape, generator=g) * 2 - 1) * bound
        p = os.path.join(out_dir, f'edlora_{i}.pth')
        torch.save({'params': d}, p)
        entries.append(dict(lora_path=p, unet_alpha=1.0, text_encoder_alpha=1.0, concept_name=f'{a} {b}'))
    cfg = os.path.join(out_dir, 'fuse.json')
    with open(cfg, 'w') as f:
        json.dump(entries, f)
    return cfg


PEAK_FP64_MFMA_TFLOPS = 78.6     # MI355X fp64 matrix (v_mfma_f64_16x16x4_f64), vendor figure; the guides give no measured one


def fusion_roofline(lsq, gram):
    """configs[3]: the kernel with the largest share of the job's library time is the L-BFGS closure (fp64 W.G - P on the fp64
    matrix cores, ~45 k launches); the Gram kernel (one call per layer and concept since round 4) is reported beside it."""
    out = None
    if lsq:
        top = lsq[0]
        sec = top['avg_us'] * 1e-6
        ach = top['flops'] / sec / 1e12
        out = dict(kernel=top['name'], bound='mfma', achieved=round(ach, 3), peak=PEAK_FP64_MFMA_TFLOPS, unit='TFLOP/s',
                   frac=round(ach / PEAK_FP64_MFMA_TFLOPS, 5), traffic=None, avg_us=round(top['avg_us'], 2), launches=top['calls'],
                   dtype='f64', algorithmic_flops_per_launch=top['flops'], algorithmic_bytes_per_launch=top['bytes'],
                   total_ms=round(sum(r['total_ms'] for r in lsq), 1))
    if gram and out is not None:
        g = gram[0]
        sec = g['avg_us'] * 1e-6
        out['gram'] = dict(kernel=g['name'], achieved_tflops=round(g['flops'] / sec / 1e12, 2), peak=PEAK_MFMA_TFLOPS,
                           frac=round(g['flops'] / sec / 1e12 / PEAK_MFMA_TFLOPS, 5), gbps=round(g['bytes'] / sec / 1e9, 1),
                           avg_us=round(g['avg_us'], 2), launches=g['calls'], total_ms=round(sum(r['total_ms'] for r in gram), 1))
    return out


def run_fusion(args, rank, world, device):
    """configs[3]: gradient_fusion.compose_concepts on 14 synthetic ED-LoRAs, iters 500 (CLIP, cross-K/V) / 50 (spatial)
    as fuse.sh:8-9; a "step" is one complete fusion. Roofline: the Gram kernel. cpu_baseline: ONE level-0 spatial layer
    (n = 81,920 rows = one concept, 320 -> 320) through the reference-semantics L-BFGS on stored features."""
    import tempfile
    import gradient_fusion as gf
    from mixofshow.hip import profiler
    n = args.concepts
    tmp = tempfile.mkdtemp(prefix='mos_fuse_')
    _log(f'writing {n} synthetic ED-LoRA checkpoints ({args.preset}) to {tmp}')
    cfg = synthetic_edlora_checkpoints(args.preset, n, tmp)
    recs, times = [], []
    for i in range(args.warmup + args.steps):
        profiling = i == args.warmup + args.steps - 1
        torch.manual_seed(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if profiling:
            profiler.begin()
        gf.compose_concepts(cfg, args.textenc_iters, args.unet_iters, f'synthetic://{args.preset}?seed=0', tmp, 'bench',
                            device, save=False)
        torch.cuda.synchronize()
        if profiling:
            recs.extend(profiler.end())
        dt = time.perf_counter() - t0
        _log(f'fusion pass {i}: {dt:.1f}s')
        if i >= args.warmup:
            times.append(dt)
    gram = sorted((r for r in recs if r['name'].startswith('gram')), key=lambda r: -r['flops'])      # largest call first
    lsq = sorted((r for r in recs if r['name'].startswith('lsq_loss_grad')), key=lambda r: -r['total_ms'])
    res = dict(metric='gradient_fusion_wall_seconds_14_edloras_sd15', value=round(statistics.mean(times), 2), unit='s',
               n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(statistics.mean(times) * 1e3, 1),
               stage_seconds_last_pass=dict(gf.STAGE_SECONDS), solve_seconds_last_pass=dict(gf.SOLVE_SECONDS),
               lockstep_solve=os.environ.get('MOS_FUSION_BATCH', '1') != '0',
               higher_is_better=False, scaling='weak', vs_baseline=None, dtype='fp16 features / fp64 Gram + L-BFGS',
               data='synthetic',
               config=dict(workload=f'BASELINE.json configs[3]: gradient_fusion of {n} synthetic ED-LoRAs into one SD-1.5 '
                                    f'UNet + CLIP, L-BFGS iters {args.textenc_iters} (text encoder, cross K/V) / '
                                    f'{args.unet_iters} (spatial), saving excluded', preset=args.preset, concepts=n,
                           host_cores=os.cpu_count(), box=box_info(device.index or 0)),
               roofline=fusion_roofline(lsq, gram), kernels=_kernel_table(recs, 1, 10))
    if not args.no_cpu_baseline:
        from oracle import fusion_ref
        threads = _cpu_threads()
        g = torch.Generator().manual_seed(0)
        X = torch.randn(81920, 320, generator=g).half().float()
        W0 = torch.randn(320, 320, generator=g) * 0.05
        Y = (X @ (W0 + 0.01 * torch.randn(320, 320, generator=g)).T).half().float()
        t0 = time.time()
        _with_alarm(240, lambda: fusion_ref.update_quasi_newton_ref(X, Y, W0.clone(), args.unet_iters))
        dt = time.time() - t0
        res['cpu_baseline'] = dict(value=round(dt, 2), unit='s', cores=threads, host_cores=os.cpu_count(), kind='port',
                                   sample=f'ONE level-0 spatial layer (n=81920 rows = 1 concept x 20 steps x 4096 tokens, '
                                          f'320->320) through the reference-semantics L-BFGS ({args.unet_iters} iters, chunked '
                                          f'fp32 closure, oracle/fusion_ref.py); the full job has 96 spatial + 32 '
                                          f'cross-K/V + 48 CLIP layers and {n}x the rows per spatial layer')
    return res


def _parity_summary(par):
    """<= 10 keys: worst case over the recorded cases (tests/test_gpu_end_to_end.py::_record_parity ->
    profiles/parity_latents.json) of the UN-normalised figures north_star's "1e-3 on denoised latents" is judged on, per
    pipeline kind: eps max-abs, latent max-abs and latent RMS of the HIP path against exact attention, and the latent max-abs of
    the reference's own fp16 arithmetic against exact (the yardstick). The full per-case record goes to the verbose file."""
    if not isinstance(par, dict) or not par.get('cases'):
        return None
    out = dict(source=par.get('source'), stale=par.get('stale'), tol=par.get('tolerance_north_star'))
    want = (('eps_max', 'abs_eps_max_hip_vs_exact'), ('latent_max', 'abs_latent_max_hip_vs_exact'),
            ('latent_max_ref_fp16', 'abs_latent_max_ref_fp16_vs_exact'), ('latent_rms', 'latent_rms_teacher_forced_hip_vs_exact'))
    for name, case in par['cases'].items():
        kind = 'fp16pipe' if 'fp16 pipeline' in name else ('fp32pipe_peaked' if 'peaked' in name else 'fp32pipe')
        flat = dict(case)
        for v in case.values():                      # eps figures live one level down in some cases
            if isinstance(v, dict):
                flat.update({k: x for k, x in v.items() if k not in flat})
        for short, key in want:
            alt = key.replace('ref_fp16', 'ref_path')
            v = flat.get(key, flat.get(alt))
            if isinstance(v, (int, float)):
                tag = f'{kind}.{short}'
                if tag not in out or v > out[tag]:
                    out[tag] = float(f'{v:.3g}')
    keep = ['source', 'stale', 'tol'] + sorted(k for k in out if k.startswith('fp32pipe.')) + \
        sorted(k for k in out if k.startswith('fp32pipe_peaked.'))[:2] + sorted(k for k in out if k.startswith('fp16pipe.'))[:2]
    return {k: out[k] for k in keep[:11]}


COMPACT_LIMIT = 4096      # bytes: the driver keeps a bounded tail of stdout and parses the LAST line of it


def _short(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 1] + '~'


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d} if isinstance(d, dict) else None


def compact_line(res, full_path=None):
    """The ONE stdout line of a run: the contract's keys + `roofline` + `cpu_baseline` + both halves of BASELINE's metric, in
    <= COMPACT_LIMIT bytes. Everything else `res` carries (kernel tables, per-case parity, the regional sub-record) is the
    VERBOSE record: written to gpurun_out/bench_full.json, never to stdout."""
    rl_keys = ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_us', 'launches',
               'algorithmic_flops_per_launch', 'algorithmic_bytes_per_launch')
    out = {k: res.get(k) for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
                                   'scaling', 'vs_baseline', 'dtype', 'data')}
    cfg = dict(res.get('config') or {})
    ccfg = _pick(cfg, ('global_batch', 'per_gpu_batch', 'image_size', 'parallelism', 'grad_bucket_bytes', 'preset', 'hipgraph',
                       'graph_capture_s', 'channels_last', 'host_cores', 'kernel_source_sha16', 'height', 'width', 'replicas',
                       'concepts', 'allreduce', 'box', 'clocks_timed_region', 'host_need_ms_per_step'))
    ccfg = dict(workload=_short(cfg.get('workload', ''), 200), **ccfg)
    out['config'] = ccfg
    out['roofline'] = _pick(res.get('roofline'), rl_keys + ('total_ms', 'dtype'))
    cb = res.get('cpu_baseline')
    if isinstance(cb, dict):
        out['cpu_baseline'] = dict(_pick(cb, ('value', 'unit', 'cores', 'host_cores', 'kind')), sample=_short(cb.get('sample'), 160))
    if isinstance(res.get('whole_step'), dict):
        out['whole_step'] = _pick(res['whole_step'], ('tflop_per_step', 'achieved_tflops', 'frac_of_mfma_peak'))
    if isinstance(res.get('attention_path'), dict):
        out['attention_path_frac'] = res['attention_path'].get('frac_of_mfma_peak')
    if res.get('library_kernel_ms_per_step') is not None:
        out['library_kernel_ms_per_step'] = res['library_kernel_ms_per_step']
    by_name = res.get('dominant_kernels_by_name')
    if by_name:
        out['by_name_ms_frac'] = {r['kernel']: [r['ms'], round(r['frac_of_mfma_peak'], 3)] for r in by_name[:5]}
    for k in ('stage_seconds_last_pass', 'solve_seconds_last_pass', 'step_ms_spread', 'data_pipeline'):
        if k in res:
            out[k] = res[k]
    reg = res.get('regional') if isinstance(res.get('regional'), dict) else (res if 'value_ms_image' in res else None)
    if reg:
        rf, rcb = reg.get('roofline') or {}, reg.get('cpu_baseline') or {}
        out['regional_ms_image'] = reg.get('value_ms_image')
        out['regional_ms_latent'] = reg.get('value_ms_latent')
        out['regional_cold_call_ms'] = reg.get('cold_call_ms')
        out['regional_roofline'] = _pick(rf, ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'avg_us', 'launches'))
        out['regional_roofline_frac'] = rf.get('frac')
        out['regional_attention_path_frac'] = (reg.get('attention_path') or {}).get('frac_of_mfma_peak')
        out['regional_cpu_baseline_ms'] = rcb.get('value')
        out['regional_launches_per_unet_call'] = reg.get('launches_per_unet_call')
        if reg is not res:
            out['regional_workload'] = _short((reg.get('config') or {}).get('workload', ''), 120)
            rn = reg.get('dominant_kernels_by_name')
            if rn:
                out['regional_by_name_ms_frac'] = {r['kernel']: [r['ms'], round(r['frac_of_mfma_peak'], 3)] for r in rn[:5]}
    out['parity'] = _parity_summary(res.get('parity'))
    out['full_record'] = full_path
    line = json.dumps(out)
    for drop in ('regional_by_name_ms_frac', 'by_name_ms_frac', 'regional_workload', 'parity', 'solve_seconds_last_pass'):
        if len(line) <= COMPACT_LIMIT:        # never reached with the shipped records; a guard, not a mechanism
            break
        out.pop(drop, None)
        line = json.dumps(out)
    assert len(line) <= COMPACT_LIMIT, len(line)
    return line


def write_full_record(res):
    """Verbose record -> gpurun_out/bench_full.json (scratch on the GPU box, merged back by gpurun). Not echoed: a 30 KB dump
    would push the timing lines out of the stderr tail the driver keeps."""
    path = None
    try:
        d = os.path.join(ROOT, 'gpurun_out')
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, 'bench_full.json')
        with open(path, 'w') as f:
            json.dump(res, f)
        path = 'gpurun_out/bench_full.json'
    except OSError:
        path = None
    _log(f'verbose record ({len(json.dumps(res))} bytes: kernel tables, per-case parity, regional sub-record) -> {path}')
    return path


def _self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one per GPU over RCCL), fail loudly
    if the node cannot run them."""
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n_dev < args.gpus:
        print(f'bench.py: --gpus {args.gpus} requested but this node exposes {n_dev} HIP device(s); refusing to print a '
              f'{args.gpus}-GPU line from fewer ranks', file=sys.stderr)
        sys.exit(2)
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    _log('launching ' + ' '.join(cmd))
    sys.exit(subprocess.run(cmd).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None, help='timed steps (default: 8; fusion: 2 complete fusions)')
    ap.add_argument('--warmup', type=int, default=None, help='untimed warm-up steps (default: 3; fusion: 1)')
    ap.add_argument('--mode', default='train', choices=['train', 'regional', 'fusion'])
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--preset', default='sd15')
    ap.add_argument('--precision', default='fp16', choices=['fp16', 'bf16'])
    ap.add_argument('--data', default='synthetic', choices=['synthetic', 'jpeg'],
                    help='train: device-resident synthetic batches (the metric) or, SURVEY 8(f).4, batches decoded from JPEG '
                         'through the dataset / transform chain by DataLoader workers inside the timed region')
    ap.add_argument('--workers', type=int, default=8, help='--data jpeg: DataLoader worker processes')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--height', type=int, default=512, help='regional mode: image height (1024 with --width 2048 = the shipped example)')
    ap.add_argument('--width', type=int, default=768)
    ap.add_argument('--no-regional', action='store_true', help='train mode: skip the regional-sample half of the metric')
    ap.add_argument('--channels-last', type=int, default=1, help='NHWC UNet/VAE (the product default)')
    ap.add_argument('--graph', type=int, default=1, help='train: forward+backward replayed from a hipGraph (the product '
                    'default, train_edlora.py); 0 = eager')
    ap.add_argument('--regional-graph', type=int, default=-1, help='regional: 1 = replay the UNet call from a hipGraph, '
                    '0 = eager, -1 = the product default (mixofshow.utils.hipgraph.sampling_default)')
    ap.add_argument('--concepts', type=int, default=14)
    ap.add_argument('--textenc-iters', type=int, default=500)
    ap.add_argument('--unet-iters', type=int, default=50)
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 2 if args.mode == 'fusion' else 8
    if args.warmup is None:
        args.warmup = 1 if args.mode == 'fusion' else 3
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        _self_launch(args)
    from mixofshow.parallel import dp
    rank, world, local = dp.init_distributed()
    if world != args.gpus:
        print(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}', file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print('bench.py needs a HIP device (MI355X); there is no CPU fallback', file=sys.stderr)
        sys.exit(2)
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)
    if args.mode == 'train':
        res = run_train(args, rank, world, device)
        if world == 1 and not args.no_regional:
            reg = run_regional(args, rank, world, device, steps=min(args.steps, 3), warmup=min(args.warmup, 1))
            res['regional'] = dict(value_ms=reg['value'], value_ms_image=reg['value_ms_image'],
                                   value_ms_latent=reg['value_ms_latent'], cold_call_ms=reg['cold_call_ms'],
                                   metric=reg['metric'], steps=reg['steps'], warmup=reg['warmup'],
                                   config=reg['config'], roofline=reg['roofline'], attention_path=reg['attention_path'],
                                   cpu_baseline=reg.get('cpu_baseline'), kernels=reg['kernels'],
                                   launches_per_unet_call=reg.get('launches_per_unet_call'),
                                   dominant_kernels_by_name=reg.get('dominant_kernels_by_name'),
                                   library_kernel_ms_per_sample=reg['library_kernel_ms_per_sample'])
        else:
            res['regional'] = None
    elif args.mode == 'regional':
        res = run_regional(args, rank, world, device)
    else:
        res = run_fusion(args, rank, world, device)
    res['parity'] = parity_figures()
    if rank == 0:
        full = write_full_record(res)
        sys.stderr.flush()
        print(compact_line(res, full), flush=True)      # the LAST stdout line, <= COMPACT_LIMIT bytes
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
