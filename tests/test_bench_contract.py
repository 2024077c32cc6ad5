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


def _check_compact_train_line(d, raw):
    """Round 6 on: the stdout line is the COMPACT record (<= 4096 bytes, flat regional keys); the verbose one is a file."""
    assert len(raw.encode()) <= 4096
    assert d['metric'] == 'edlora_train_images_per_sec_512_sd15' and d['unit'] == 'images/s'
    assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None and d['data'] == 'synthetic'
    assert isinstance(d['config']['workload'], str) and 'model' not in d['config'] and d['config']['hipgraph'] is True
    assert abs(d['value'] - d['config']['global_batch'] / (d['ms_per_step'] * 1e-3)) < 0.02 * d['value']
    _check_roofline(d['roofline'])
    _check_cpu_baseline(d['cpu_baseline'], d['unit'])
    assert d['value'] > d['cpu_baseline']['value']
    assert d['regional_ms_image'] > 0 and d['regional_ms_latent'] > 0 and 0 < d['regional_roofline_frac'] < 1
    assert d['regional_cpu_baseline_ms'] > d['regional_ms_image']
    assert 'id' in d['config']['box'] and 'kernels' not in d and 'regional' not in d


@pytest.mark.parametrize('path', _train_lines())
def test_committed_train_bench_line_schema(path):
    raw = open(path).read().strip().splitlines()[-1]
    d = json.loads(raw)
    if 'full_record' in d:
        return _check_compact_train_line(d, raw)
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


@pytest.mark.parametrize('path', ['profiles/r05_bench_train_n1.json', 'profiles/r05_bench_fusion.json',
                                  'profiles/r05_bench_regional_1024x2048.json'])
def test_stdout_line_is_compact_and_parseable(path):
    """the 30 KB round-5 line could not be parsed by the driver. The line bench.py prints is now built by
    `compact_line` from the verbose record; fed the largest verbose records on file it must stay under 4096 bytes, parse, and
    still carry the contract's keys, `roofline`, `cpu_baseline` and both halves of BASELINE's metric."""
    import bench
    full = json.loads(open(os.path.join(ROOT, path)).read().strip().splitlines()[-1])
    line = bench.compact_line(full, 'gpurun_out/bench_full.json')
    assert len(line.encode()) <= bench.COMPACT_LIMIT == 4096 and '\n' not in line
    d = json.loads(line)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'parity', 'full_record'):
        assert k in d, k
    assert d['value'] == full['value'] and d['ms_per_step'] == full['ms_per_step']
    assert d['roofline']['frac'] == full['roofline']['frac'] and d['roofline']['kernel'] == full['roofline']['kernel']
    assert 'workload' in d['config'] and 'model' not in d['config']
    assert not any(k in d for k in ('kernels', 'regional', 'dominant_kernels_by_name'))     # verbose tables stay in the file
    assert len(d['parity']) <= 11 and d['parity']['tol'] == 1e-3
    if full['metric'].startswith('edlora_train'):
        assert d['cpu_baseline']['value'] == full['cpu_baseline']['value'] and d['cpu_baseline']['kind'] == 'port'
        assert d['regional_ms_image'] == full['regional']['value_ms_image']
        assert d['regional_ms_latent'] == full['regional']['value_ms_latent']
        assert d['regional_roofline_frac'] == full['regional']['roofline']['frac']
        assert d['regional_cpu_baseline_ms'] == full['regional']['cpu_baseline']['value']
        assert d['whole_step']['frac_of_mfma_peak'] == full['whole_step']['frac_of_mfma_peak']
        assert 'fp32pipe.latent_rms' in d['parity'] and 'fp32pipe.latent_max' in d['parity']


def test_main_prints_exactly_one_compact_stdout_line():
    """bench.main() writes the verbose record to stderr / gpurun_out and ONLY the compact line to stdout."""
    src = open(os.path.join(ROOT, 'bench.py')).read()
    main = src[src.index('def main():'):]
    assert main.count('print(compact_line(res, full), flush=True)') == 1
    assert 'print(json.dumps(res))' not in src


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
