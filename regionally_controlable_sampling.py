"""Regionally controllable sampling CLI (the entry point of SURVEY rows R1-R3).

Command line, region grammar and output layout follow the reference script of the same name (its :55-187), so the
reference's `regionally_sample.sh` recipes run unchanged:

  --prompt_rewrite '[region prompt]-*-[region negative prompt]-*-[h0, w0, h1, w1]|[...]-*-[...]-*-[...]'

with pixel boxes that are converted to fractions of the image (an empty box `[]` = whole image). The fused model
directory is what `gradient_fusion.py` writes (`combined_model_<suffix>/` with `new_concept_cfg.json`). Differences,
both forced by the offline environment: T2I-Adapter weights come from `--keypose_adapter_path` /
`--sketch_adapter_path` (the reference pulls them from the HF hub), and without a condition image the size comes from
`--height/--width` (the reference reads it off the condition image). Latents are drawn from a CPU generator so a seed
gives the same image on every device.
"""
import argparse
import ast
import hashlib
import json
import os

import mos_path  # noqa: F401
import torch

from mixofshow.models.schedulers import DPMSolverMultistepScheduler
from mixofshow.pipelines.pipeline_regionally_t2iadapter import RegionallyT2IAdapterPipeline, T2IAdapter

# (flag, type, default) — the reference's flag set plus the four offline additions at the end
CLI = [('pretrained_model', str, None), ('sketch_condition', str, None), ('sketch_adaptor_weight', float, 1.0),
       ('region_sketch_adaptor_weight', str, ''), ('keypose_condition', str, None), ('keypose_adaptor_weight', float, 1.0),
       ('region_keypose_adaptor_weight', str, ''), ('save_dir', str, None), ('prompt', str, 'photo of a toy'),
       ('negative_prompt', str, ''), ('prompt_rewrite', str, ''), ('seed', int, 16141), ('suffix', str, ''),
       ('keypose_adapter_path', str, None), ('sketch_adapter_path', str, None), ('height', int, 512), ('width', int, 512)]
ADAPTER_KINDS = {'keypose': ('RGB', 3), 'sketch': ('L', 1)}     # condition image mode, adapter input channels


def _strip_brackets(text):
    return text.replace('[', '').replace(']', '')


def prepare_text(prompt, region_prompts, height, width):
    """Region grammar -> `(prompt, [(region prompt, region negative prompt, [h0, w0, h1, w1] as fractions), ...])`,
    the structure `RegionallyT2IAdapterPipeline.__call__` takes per sample. Parsing stops at the first empty segment
    (a trailing '|' is allowed)."""
    regions = []
    for segment in region_prompts.split('|'):
        if not segment:
            break
        subject, negative, box_text = segment.split('-*-')
        box = list(ast.literal_eval(box_text.strip()))
        if box:
            box = [box[0] / height, box[1] / width, box[2] / height, box[3] / width]
        else:
            box = [0, 0, 1, 1]
        regions.append((_strip_brackets(subject), _strip_brackets(negative), box))
    return (prompt, regions)


def _read_state_dict(path):
    if path.endswith('.safetensors'):
        from safetensors.torch import load_file
        return load_file(path)
    return torch.load(path, map_location='cpu')


def build_model(pretrained_model, device, keypose_adapter_path=None, sketch_adapter_path=None):
    """Fused checkpoint + concept table + DPM-Solver++ scheduler (+ adapters when weight files are given)."""
    cfg_file = os.path.join(pretrained_model, 'new_concept_cfg.json')
    if not os.path.exists(cfg_file):
        raise FileNotFoundError(f'{cfg_file}: a fused model directory written by gradient_fusion.py is expected')
    pipe = RegionallyT2IAdapterPipeline.from_pretrained(pretrained_model, torch_dtype=torch.float16).to(device)
    with open(cfg_file) as f:
        concepts = json.load(f)
    for entry in concepts.values():
        pipe.tokenizer.add_tokens(entry['concept_token_names'])
    pipe.set_new_concept_cfg(concepts)
    pipe.scheduler = DPMSolverMultistepScheduler()
    for kind, path in (('keypose', keypose_adapter_path), ('sketch', sketch_adapter_path)):
        if path:
            adapter = T2IAdapter(in_channels=ADAPTER_KINDS[kind][1])
            adapter.load_state_dict(_read_state_dict(path))
            setattr(pipe, f'{kind}_adapter', adapter.to(device, torch.float16))
    return pipe


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    for name, typ, default in CLI:
        parser.add_argument(f'--{name}', type=typ, default=default, required=(name == 'pretrained_model'))
    return parser.parse_args(argv)


def main(argv=None):
    from PIL import Image
    args = parse_args(argv)
    device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')     # the kernels themselves need the GPU
    pipe = build_model(args.pretrained_model, device, args.keypose_adapter_path, args.sketch_adapter_path)
    height, width = args.height, args.width
    conditions = {}
    for kind, (mode, _) in ADAPTER_KINDS.items():
        path = getattr(args, f'{kind}_condition')
        conditions[kind] = None
        if path and os.path.exists(path):
            conditions[kind] = Image.open(path).convert(mode)
            width, height = conditions[kind].size            # a condition image fixes the output size
    sample = prepare_text(args.prompt, args.prompt_rewrite, height, width)
    images = pipe(prompt=[sample], negative_prompt=[args.negative_prompt], height=height, width=width,
                  keypose_adapter_input=[conditions['keypose']] if conditions['keypose'] is not None else None,
                  keypose_adaptor_weight=args.keypose_adaptor_weight,
                  region_keypose_adaptor_weight=args.region_keypose_adaptor_weight,
                  sketch_adapter_input=[conditions['sketch']] if conditions['sketch'] is not None else None,
                  sketch_adaptor_weight=args.sketch_adaptor_weight,
                  region_sketch_adaptor_weight=args.region_sketch_adaptor_weight,
                  generator=torch.Generator('cpu').manual_seed(args.seed), num_inference_steps=50,
                  guidance_scale=7.5).images
    # output: <save_dir>/seed_<seed>/<prompt>---<suffix>---<8 hex of the run record>.png + the record next to it
    record = {name: getattr(args, name) for name, _, _ in CLI if name != 'save_dir'}
    record_text = json.dumps(record, indent=1, sort_keys=True)
    tag = hashlib.sha256(record_text.encode('utf-8')).hexdigest()[:8]
    out_dir = os.path.join(args.save_dir or '.', f'seed_{args.seed}')
    os.makedirs(out_dir, exist_ok=True)
    stem = os.path.join(out_dir, f"{args.prompt.replace(' ', '_')}---{args.suffix}---{tag}")
    images[0].save(stem + '.png')
    with open(stem + '.txt', 'w') as f:
        f.write(record_text + '\n')
    print(f'save to: {out_dir}')


if __name__ == '__main__':
    main()
