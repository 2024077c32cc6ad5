"""Synthetic ED-LoRA checkpoints for the gradient-fusion parity tests (SURVEY 8d cfg #4 recipe: seeds 0.., lora_up ~
N(0, 0.02^2)) — shared by tests/test_fusion_cpu.py and tests/golden/make_golden.py (which passes the product's
build_trainer loaded under an alias, see there)."""
import json

import torch


def make_fusion_fixture(tmp_path, preset, n_concepts=2, up_std=0.02, build_trainer=None):
    if build_trainer is None:
        from bench import build_trainer
    names = [('<potter1>', '<potter2>'), ('<thanos1>', '<thanos2>'), ('<hermione1>', '<hermione2>')][:n_concepts]
    ckpts = []
    for i, (a, b) in enumerate(names):
        tr = build_trainer(preset, torch.device('cpu'), seed=i)
        torch.manual_seed(100 + i)
        with torch.no_grad():
            for l in list(tr.text_encoder_lora) + list(tr.unet_lora):
                l.lora_up.weight.normal_(0, up_std)
            tr.concept_embedding.add_(torch.randn_like(tr.concept_embedding) * 0.01)
        d = tr.delta_state_dict()
        d['new_concept_embedding'] = {a: d['new_concept_embedding']['<potter1>'], b: d['new_concept_embedding']['<potter2>']}
        p = str(tmp_path / f'c{i}.pth')
        torch.save({'params': d}, p)
        ckpts.append(dict(lora_path=p, unet_alpha=1.0 - 0.2 * i, text_encoder_alpha=0.9, concept_name=f'{a} {b}'))
    cfg = str(tmp_path / 'fuse.json')
    with open(cfg, 'w') as f:
        json.dump(ckpts, f)
    return cfg
