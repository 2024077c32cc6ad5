"""The bench line the driver parses: schema of the committed evidence (profiles/*bench*.json, written by `python bench.py`
on an MI355X) and of the pure-host helpers of bench.py."""
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check_roofline(r):
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] == ('TFLOP/s' if r['bound'] == 'mfma' else 'GB/s')
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-4 and 0 < r['frac'] < 1
    assert r['peak'] == (2500.0 if r['bound'] == 'mfma' else 8000.0)
    assert r['traffic'] is None or r['traffic'] > 0


def _check_cpu_baseline(c, unit):
    assert c['kind'] in ('reference', 'port') and c['cores'] >= 1 and c['value'] > 0 and c['unit'] == unit
    assert isinstance(c['sample'], str)


def _train_lines():
    return sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_bench_train_n1.json')))


@pytest.mark.parametrize('path', _train_lines())
def test_committed_train_bench_line_schema(path):
    d = json.load(open(path))
    base = json.load(open(os.path.join(ROOT, 'BASELINE.json')))
    assert d['metric'] == 'edlora_train_images_per_sec_512_sd15' and 'ED-LoRA train images/sec' in base['metric']
    for k, t in (('value', float), ('unit', str), ('n_gpus', int), ('steps', int), ('warmup', int),
                 ('ms_per_step', float), ('higher_is_better', bool), ('scaling', str), ('dtype', str), ('data', str)):
        assert isinstance(d[k], t), k
    assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert d['data'] == 'synthetic' and d['dtype'] in ('fp16', 'bf16') and d['n_gpus'] == 1
    assert isinstance(d['config']['workload'], str) and 'model' not in d['config']
    _check_roofline(d['roofline'])
    _check_cpu_baseline(d['cpu_baseline'], d['unit'])
    assert d['value'] > d['cpu_baseline']['value']
    assert abs(d['value'] - d['config']['global_batch'] / (d['ms_per_step'] * 1e-3)) < 0.02 * d['value']
    if os.path.basename(path) >= 'r02':          # round 2 on: one line carries BOTH halves of BASELINE's metric
        reg = d['regional']
        assert reg['metric'] == 'regional_sample_latency_ms_50step_512x768_3regions' and reg['value_ms'] > 0
        _check_roofline(reg['roofline'])
        _check_cpu_baseline(reg['cpu_baseline'], 'ms')
        assert reg['cpu_baseline']['value'] > reg['value_ms']
        assert d['config']['hipgraph'] is True and d['config']['host_cores'] >= d['cpu_baseline']['cores']
        a = d['attention_path']
        assert abs(a['algorithmic_tflop'] - 0.604 * d['config']['per_gpu_batch']) < 1e-3
        assert abs(a['frac_of_mfma_peak'] - a['achieved_tflops'] / 2500.0) < 1e-4


def test_product_default_is_the_benchmarked_mode():
    """bench.py measures the hipGraph step; the product entry point and the shipped recipe default to it."""
    import yaml
    src = open(os.path.join(ROOT, 'train_edlora.py')).read()
    assert "opt['train'].get('hipgraph', True)" in src
    opt = yaml.safe_load(open(os.path.join(ROOT, 'options/train/EDLoRA/synthetic/8101_EDLoRA_potter_synthetic_B4.yml')))
    assert opt['train']['hipgraph'] is True
    import bench
    import argparse  # noqa: F401
    assert 'default=1' in [l for l in open(os.path.join(ROOT, 'bench.py')) if "'--graph'" in l][0]
    assert bench.ATTN_PATH_GFLOP_PER_TRAINED_IMAGE == 604.0


def test_roofline_helper_and_pmc_traffic_staleness(tmp_path):
    import bench
    recs = [dict(name='attn_bwd_dkdv f16 d40 B4 H8 Nq4096 Nkv4096', calls=10, total_ms=3.0, avg_us=300.0,
                 flops=1.718e11, bytes=6.29e7),
            dict(name='lora_down(skinny_nt) M16384 K320', calls=60, total_ms=0.6, avg_us=10.0, flops=1.7e8, bytes=1.1e7)]
    r = bench.roofline_from_profile(recs)
    assert r['bound'] == 'mfma' and r['kernel'] == recs[0]['name']
    assert abs(r['achieved'] - 1.718e11 / 300e-6 / 1e12) < 1e-2 and abs(r['frac'] - r['achieved'] / 2500.0) < 1e-4
    r2 = bench.roofline_from_profile(recs[::-1])
    assert r2['bound'] == 'hbm' and r2['unit'] == 'GB/s'
    # traffic is only reported when the PMC file was measured on the kernel sources in the tree
    good = tmp_path / 'ok.json'
    good.write_text(json.dumps({'source_sha16': bench.kernel_source_fingerprint(),
                                'kernels': {recs[0]['name']: {'total': 1.27e8}}}))
    stale = tmp_path / 'stale.json'
    stale.write_text(json.dumps({'source_sha16': '0' * 16, 'kernels': {recs[0]['name']: {'total': 1.27e8}}}))
    assert bench._pmc_traffic(recs[0]['name'], str(good)) == 1.27e8
    assert bench._pmc_traffic(recs[0]['name'], str(stale)) is None
    assert bench._pmc_traffic('no such kernel', str(good)) is None
    agg = [dict(name='attn_fwd f16 d40', total_ms=30.0), dict(name='gemm_nt f16 M1', total_ms=13.4),
           dict(name='conv3x3 f16 B4', total_ms=20.0), dict(name='groupnorm_apply f16', total_ms=2.0)]
    a = bench.attention_path_aggregate(604.0, 4, agg, 2)
    assert abs(a['attention_path_kernel_ms'] - 21.7) < 1e-6 and abs(a['other_library_kernel_ms'] - 11.0) < 1e-6
    assert abs(a['achieved_tflops'] - 2.416 / 21.7e-3) < 0.5 and abs(a['frac_of_mfma_peak'] - a['achieved_tflops'] / 2500) < 1e-4


def test_gpus_flag_without_launcher_refuses_to_fake_ranks():
    """`python bench.py --gpus 8` with no WORLD_SIZE must start the ranks itself or fail loudly — never print an
    n_gpus line from one rank. Here there is no HIP device: it must exit non-zero with a clear message."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode != 0 and 'HIP device' in p.stderr and p.stdout.strip() == ''


def test_jpeg_data_pipeline_feeds_the_training_step(emulated_hip, tmp_path):
    """SURVEY 8(f).4 (`bench.py --data jpeg`): a synthetic JPEG concept in the reference's folder layout goes through
    LoraDataset + the shipped transform chain + DataLoader collate into the batch format TrainEngine.step consumes."""
    import torch
    import bench
    from mixofshow.pipelines.train_loop import TrainEngine
    from tests.test_host_cpu import _trainer
    loader = bench.jpeg_loader(str(tmp_path), batch=2, size=128, workers=0, n_images=3)
    assert sorted(os.listdir(tmp_path / 'image')) == ['000.jpg', '001.jpg', '002.jpg']
    it = iter(loader)
    b = next(it)
    assert b['images'].shape == (2, 3, 128, 128) and b['images'].dtype == torch.float32
    assert -1.0 <= float(b['images'].min()) < float(b['images'].max()) <= 1.0
    assert b['masks'].shape == b['img_masks'].shape == (2, 1, 16, 16) and 0.05 < float(b['masks'].mean()) < 0.6
    assert all('<potter1> <potter2>' in p and '<TOK>' not in p for p in b['prompts'])
    b2 = next(it)
    assert not torch.equal(b['images'], b2['images'])                 # shuffled photos / random crops
    tr = _trainer()
    opt = dict(optim_g=dict(type='AdamW', lr=0.0, weight_decay=0.01, betas=[0.9, 0.999]), emb_norm_threshold=0.55)
    engine = TrainEngine(tr, opt, total_iter=10, mixed_precision='no')
    out = engine.step(b)
    assert torch.isfinite(out['loss']) and engine.global_step == 1
