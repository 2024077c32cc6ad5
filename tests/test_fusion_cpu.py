"""gradient_fusion.py end to end on the CPU ('tiny' preset, HIP primitives emulated): the host logic of SURVEY rows
F1-F3 — checkpoint parsing, concept-token numbering, feature recording through the fused projections' taps, per-layer
Gram accumulation + L-BFGS, weight write-back, and the saved directory layout the regional pipeline loads."""
import json
import os

import torch


def test_compose_concepts_end_to_end(emulated_hip, tmp_path):
    import gradient_fusion as gf
    from bench import build_trainer
    from mixofshow.pipelines.pipeline_regionally_t2iadapter import RegionallyT2IAdapterPipeline
    ckpts = []
    for i, (a, b) in enumerate([('<potter1>', '<potter2>'), ('<thanos1>', '<thanos2>')]):
        tr = build_trainer('tiny', torch.device('cpu'), seed=i)
        torch.manual_seed(100 + i)
        with torch.no_grad():
            for l in list(tr.text_encoder_lora) + list(tr.unet_lora):
                l.lora_up.weight.normal_(0, 0.02)
        d = tr.delta_state_dict()
        d['new_concept_embedding'] = {a: d['new_concept_embedding']['<potter1>'], b: d['new_concept_embedding']['<potter2>']}
        p = str(tmp_path / f'c{i}.pth')
        torch.save({'params': d}, p)
        ckpts.append(dict(lora_path=p, unet_alpha=1.0, text_encoder_alpha=1.0, concept_name=f'{a} {b}'))
    cfg = str(tmp_path / 'fuse.json')
    with open(cfg, 'w') as f:
        json.dump(ckpts, f)
    base = RegionallyT2IAdapterPipeline.from_pretrained('synthetic://tiny?seed=0', torch_dtype=torch.float16)
    base_unet = {k: v.clone() for k, v in base.unet.state_dict().items()}
    pipe, new_cfg = gf.compose_concepts(cfg, 20, 8, 'synthetic://tiny?seed=0', str(tmp_path), 'base', torch.device('cpu'))
    # concept table: order of the json, 16 layer tokens per word, numbering advances by 16 per word
    assert list(new_cfg) == ['<potter1>', '<potter2>', '<thanos1>', '<thanos2>']
    assert [new_cfg[k]['concept_token_names'][0] for k in new_cfg] == ['<new0>', '<new16>', '<new32>', '<new48>']
    ids = [i for k in new_cfg for i in new_cfg[k]['concept_token_ids']]
    assert len(ids) == 64 and ids == list(range(ids[0], ids[0] + 64))
    # saved layout (what regionally_controlable_sampling.py loads)
    out = tmp_path / 'combined_model_base'
    assert (out / 'unet' / 'diffusion_pytorch_model.safetensors').exists() and (out / 'new_concept_cfg.json').exists()
    with open(out / 'new_concept_cfg.json') as f:
        assert json.load(f) == new_cfg
    # every fused tensor is finite; cross-attention K/V and the spatial projections moved, the rest did not
    fused = pipe.unet.state_dict()
    moved = [k for k in fused if not torch.equal(fused[k], base_unet[k])]
    assert moved and all(torch.isfinite(fused[k].float()).all() for k in fused)
    assert any('attn2.to_k.weight' in k for k in moved) and any('attn2.to_v.weight' in k for k in moved)
    assert any('attn1.to_q.weight' in k for k in moved)
    assert not any(k.startswith('conv_in') or 'norm' in k for k in moved)
    # the directory round-trips through from_pretrained with the enlarged token table
    again = RegionallyT2IAdapterPipeline.from_pretrained(str(out), torch_dtype=torch.float16)
    assert again.text_encoder.get_input_embeddings().weight.shape[0] >= ids[-1] + 1
    for k, v in again.unet.state_dict().items():
        assert torch.equal(v, fused[k]), k
    assert again.tokenizer.convert_tokens_to_ids('<new32>') == new_cfg['<thanos1>']['concept_token_ids'][0]
    # and through the sampling CLI's loader (adds the concept tokens again: must be idempotent)
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('mos_regional_cli2', os.path.join(root, 'regionally_controlable_sampling.py'))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    rp = cli.build_model(str(out), torch.device('cpu'))
    assert rp.new_concept_cfg == new_cfg
    assert rp.tokenizer.convert_tokens_to_ids('<new63>') == ids[-1]
    assert rp.scheduler.__class__.__name__ == 'DPMSolverMultistepScheduler'
    # the whole sampling CLI on the fused directory: two regions, 50 DPM-Solver++ steps at 64x64
    save_dir = tmp_path / 'samples'
    cli.main(['--pretrained_model', str(out), '--prompt', 'two people', '--negative_prompt', 'blurry',
              '--prompt_rewrite', '[a <potter1> <potter2>]-*-[blurry]-*-[0, 0, 64, 30]|[a <thanos1> <thanos2>]-*-[]-*-[0, 28, 64, 64]',
              '--height', '64', '--width', '64', '--seed', '3', '--suffix', 'cpu', '--save_dir', str(save_dir)])
    files = sorted(os.listdir(save_dir / 'seed_3'))
    assert len(files) == 2 and files[0].startswith('two_people---cpu---') and files[0].endswith('.png')
    with open(save_dir / 'seed_3' / files[1]) as f:
        assert json.load(f)['prompt_rewrite'].startswith('[a <potter1>')
