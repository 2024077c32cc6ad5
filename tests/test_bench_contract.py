"""The bench line the driver parses: schema of the committed evidence (profiles/r01_bench_*.json, written by
`python bench.py` on an MI355X) and of the pure-host helpers of bench.py."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('name,metric,hib', [
    ('r01_bench_train_n1.json', 'edlora_train_images_per_sec_512_sd15', True),
    ('r01_bench_regional_n1.json', 'regional_sample_latency_ms_50step_512x768_3regions', False)])
def test_committed_bench_line_schema(name, metric, hib):
    d = json.load(open(os.path.join(ROOT, 'profiles', name)))
    base = json.load(open(os.path.join(ROOT, 'BASELINE.json')))
    assert d['metric'] == metric and metric.split('_')[0] in json.dumps(base)
    for k, t in (('value', float), ('unit', str), ('n_gpus', int), ('steps', int), ('warmup', int),
                 ('ms_per_step', float), ('higher_is_better', bool), ('scaling', str), ('dtype', str), ('data', str)):
        assert isinstance(d[k], t), k
    assert d['higher_is_better'] is hib and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert d['data'] == 'synthetic' and d['dtype'] in ('fp16', 'bf16') and d['n_gpus'] == 1
    assert isinstance(d['config']['workload'], str) and 'model' not in d['config']
    r = d['roofline']
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] == ('TFLOP/s' if r['bound'] == 'mfma' else 'GB/s')
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-4 and 0 < r['frac'] < 1
    assert r['peak'] == (2500.0 if r['bound'] == 'mfma' else 8000.0)
    assert r['traffic'] is None or r['traffic'] > 0
    if metric.startswith('edlora'):
        c = d['cpu_baseline']
        assert c['kind'] in ('reference', 'port') and c['cores'] >= 1 and c['value'] > 0 and c['unit'] == d['unit']
        assert isinstance(c['sample'], str) and d['value'] > c['value']
        assert abs(d['value'] - d['config']['global_batch'] / (d['ms_per_step'] * 1e-3)) < 0.02 * d['value']


def test_roofline_helper_and_pmc_traffic():
    import bench
    recs = [dict(name='attn_bwd_dkdv f16 d40 B4 H8 Nq4096 Nkv4096', calls=10, total_ms=3.0, avg_us=300.0,
                 flops=1.718e11, bytes=6.29e7),
            dict(name='lora_down(skinny_nt) M16384 K320', calls=60, total_ms=0.6, avg_us=10.0, flops=1.7e8, bytes=1.1e7)]
    r = bench.roofline_from_profile(recs)
    assert r['bound'] == 'mfma' and r['kernel'] == recs[0]['name']
    assert abs(r['achieved'] - 1.718e11 / 300e-6 / 1e12) < 1e-2 and abs(r['frac'] - r['achieved'] / 2500.0) < 1e-4
    assert r['traffic'] == bench._pmc_traffic(recs[0]['name']) and r['traffic'] > recs[0]['bytes']
    r2 = bench.roofline_from_profile(recs[::-1])
    assert r2['bound'] == 'hbm' and r2['unit'] == 'GB/s' and r2['traffic'] is None
