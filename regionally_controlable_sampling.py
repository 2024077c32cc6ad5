"""Regionally controllable sampling CLI — same flags, region grammar and output naming as the reference's
regionally_controlable_sampling.py (:55-187):

  --prompt_rewrite '[region prompt]-*-[region negative prompt]-*-[h0, w0, h1, w1]|...'   (pixel boxes)

The fused model directory is what gradient_fusion.py writes (`combined_model_<suffix>/` + new_concept_cfg.json).
T2I-Adapter weights are loaded from `--keypose_adapter_path` / `--sketch_adapter_path` when given (the reference
downloads them from the HF hub, :62-63; there is no network here). Without a condition image the height/width come
from `--height/--width` (the reference derives them from the condition image, :139).
"""
import argparse
import hashlib
import json
import os

import mos_path  # noqa: F401
import torch

from mixofshow.models.schedulers import DPMSolverMultistepScheduler
from mixofshow.pipelines.pipeline_regionally_t2iadapter import RegionallyT2IAdapterPipeline, T2IAdapter


def sample_image(pipe, input_prompt, input_neg_prompt=None, generator=None, num_inference_steps=50, guidance_scale=7.5,
                 sketch_adaptor_weight=1.0, region_sketch_adaptor_weight='', keypose_adaptor_weight=1.0,
                 region_keypose_adaptor_weight='', **extra_kargs):
    keypose_condition = extra_kargs.pop('keypose_condition')
    sketch_condition = extra_kargs.pop('sketch_condition')
    return pipe(prompt=input_prompt, negative_prompt=input_neg_prompt,
                keypose_adapter_input=[keypose_condition] * len(input_prompt) if keypose_condition is not None else None,
                keypose_adaptor_weight=keypose_adaptor_weight, region_keypose_adaptor_weight=region_keypose_adaptor_weight,
                sketch_adapter_input=[sketch_condition] * len(input_prompt) if sketch_condition is not None else None,
                sketch_adaptor_weight=sketch_adaptor_weight, region_sketch_adaptor_weight=region_sketch_adaptor_weight,
                generator=generator, guidance_scale=guidance_scale, num_inference_steps=num_inference_steps,
                **extra_kargs).images


def _load_adapter(path, in_channels, device):
    adapter = T2IAdapter(in_channels=in_channels)
    if path.endswith('.safetensors'):
        from safetensors.torch import load_file
        sd = load_file(path)
    else:
        sd = torch.load(path, map_location='cpu')
    adapter.load_state_dict(sd)
    return adapter.to(device, torch.float16)


def build_model(pretrained_model, device, keypose_adapter_path=None, sketch_adapter_path=None):
    pipe = RegionallyT2IAdapterPipeline.from_pretrained(pretrained_model, torch_dtype=torch.float16).to(device)
    cfg_path = os.path.join(pretrained_model, 'new_concept_cfg.json')
    assert os.path.exists(cfg_path), f'{cfg_path} not found (written by gradient_fusion.py)'
    with open(cfg_path, 'r') as f:
        new_concept_cfg = json.load(f)
    for cfg in new_concept_cfg.values():
        pipe.tokenizer.add_tokens(cfg['concept_token_names'])
    pipe.set_new_concept_cfg(new_concept_cfg)
    pipe.scheduler = DPMSolverMultistepScheduler()
    if keypose_adapter_path:
        pipe.keypose_adapter = _load_adapter(keypose_adapter_path, 3, device)
    if sketch_adapter_path:
        pipe.sketch_adapter = _load_adapter(sketch_adapter_path, 1, device)
    return pipe


def prepare_text(prompt, region_prompts, height, width):
    """'[subject]-*-[negative]-*-[h0, w0, h1, w1]|...' -> (prompt, [(subject, negative, fractional box)])
    (reference :67-94; an empty box means the whole image)."""
    collection = []
    for region in region_prompts.split('|'):
        if region == '':
            break
        prompt_region, neg_prompt_region, pos = region.split('-*-')
        prompt_region = prompt_region.replace('[', '').replace(']', '')
        neg_prompt_region = neg_prompt_region.replace('[', '').replace(']', '')
        pos = eval(pos)  # noqa: S307 — box literal, as in the reference
        if len(pos) == 0:
            pos = [0, 0, 1, 1]
        else:
            pos[0], pos[2] = pos[0] / height, pos[2] / height
            pos[1], pos[3] = pos[1] / width, pos[3] / width
        collection.append((prompt_region, neg_prompt_region, pos))
    return (prompt, collection)


def parse_args():
    parser = argparse.ArgumentParser('', add_help=False)
    parser.add_argument('--pretrained_model', required=True, type=str)
    parser.add_argument('--sketch_condition', default=None, type=str)
    parser.add_argument('--sketch_adaptor_weight', default=1.0, type=float)
    parser.add_argument('--region_sketch_adaptor_weight', default='', type=str)
    parser.add_argument('--keypose_condition', default=None, type=str)
    parser.add_argument('--keypose_adaptor_weight', default=1.0, type=float)
    parser.add_argument('--region_keypose_adaptor_weight', default='', type=str)
    parser.add_argument('--save_dir', default=None, type=str)
    parser.add_argument('--prompt', default='photo of a toy', type=str)
    parser.add_argument('--negative_prompt', default='', type=str)
    parser.add_argument('--prompt_rewrite', default='', type=str)
    parser.add_argument('--seed', default=16141, type=int)
    parser.add_argument('--suffix', default='', type=str)
    parser.add_argument('--keypose_adapter_path', default=None, type=str)
    parser.add_argument('--sketch_adapter_path', default=None, type=str)
    parser.add_argument('--height', default=512, type=int)
    parser.add_argument('--width', default=512, type=int)
    return parser.parse_args()


if __name__ == '__main__':
    from PIL import Image
    args = parse_args()
    device = torch.device('cuda')
    pipe = build_model(args.pretrained_model, device, args.keypose_adapter_path, args.sketch_adapter_path)
    sketch = keypose = None
    width, height = args.width, args.height
    if args.sketch_condition and os.path.exists(args.sketch_condition):
        sketch = Image.open(args.sketch_condition).convert('L')
        width, height = sketch.size
    if args.keypose_condition and os.path.exists(args.keypose_condition):
        keypose = Image.open(args.keypose_condition).convert('RGB')
        width, height = keypose.size
    input_prompt = [prepare_text(args.prompt, args.prompt_rewrite, height, width)]
    # latents from a CPU generator: identical across devices/platforms (the reference seeds a device generator, :157)
    image = sample_image(pipe, input_prompt=input_prompt, input_neg_prompt=[args.negative_prompt],
                         generator=torch.Generator('cpu').manual_seed(args.seed),
                         sketch_adaptor_weight=args.sketch_adaptor_weight,
                         region_sketch_adaptor_weight=args.region_sketch_adaptor_weight,
                         keypose_adaptor_weight=args.keypose_adaptor_weight,
                         region_keypose_adaptor_weight=args.region_keypose_adaptor_weight,
                         sketch_condition=sketch, keypose_condition=keypose, height=height, width=width)
    configs = [
        f'pretrained_model: {args.pretrained_model}\n', f'context_prompt: {args.prompt}\n',
        f'neg_context_prompt: {args.negative_prompt}\n', f'sketch_condition: {args.sketch_condition}\n',
        f'sketch_adaptor_weight: {args.sketch_adaptor_weight}\n',
        f'region_sketch_adaptor_weight: {args.region_sketch_adaptor_weight}\n',
        f'keypose_condition: {args.keypose_condition}\n', f'keypose_adaptor_weight: {args.keypose_adaptor_weight}\n',
        f'region_keypose_adaptor_weight: {args.region_keypose_adaptor_weight}\n', f'random seed: {args.seed}\n',
        f'prompt_rewrite: {args.prompt_rewrite}\n'
    ]
    hash_code = hashlib.sha256(''.join(configs).encode('utf-8')).hexdigest()[:8]
    save_name = f"{args.prompt.replace(' ', '_')}---{args.suffix}---{hash_code}.png"
    save_dir = os.path.join(args.save_dir, f'seed_{args.seed}')
    os.makedirs(save_dir, exist_ok=True)
    image[0].save(os.path.join(save_dir, save_name))
    with open(os.path.join(save_dir, save_name.replace('.png', '.txt')), 'w') as fw:
        fw.writelines(configs)
    print(f'save to: {save_dir}')
