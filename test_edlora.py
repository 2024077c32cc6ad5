"""Sampling / visual validation entry point (reference test_edlora.py:24-101): EDLoRAPipeline on the HIP path with
the PromptDataset's CPU-seeded latents; writes one PNG per (prompt, sample)."""
import argparse
import os
import os.path as osp

import mos_path  # noqa: F401
import torch

from mixofshow.data.prompt_dataset import PromptDataset
from mixofshow.pipelines.pipeline_edlora import EDLoRAPipeline, StableDiffusionPipeline
from mixofshow.utils.convert_edlora_to_diffusers import convert_edlora
from mixofshow.utils.options import load_options


def visual_validation(pipe, dataloader, tag, opt, rank=0):
    out_dir = osp.join(opt['path']['visualization'], tag)
    os.makedirs(out_dir, exist_ok=True)
    for batch in dataloader:
        latents = batch['latents'].to(pipe.device) if opt['val'].get('use_fixed_latents', True) else None
        images = pipe(prompt=list(batch['prompts']), latents=latents,
                      num_inference_steps=opt['val']['sample'].get('num_inference_steps', 50),
                      guidance_scale=opt['val']['sample'].get('guidance_scale', 7.5)).images
        for img, prompt, idx in zip(images, batch['prompts'], batch['indices']):
            img.save(osp.join(out_dir, f"{prompt.replace(' ', '_')[:80]}---G_7.5_S_50---{int(idx)}---{tag}.png"))


def test(root_path, args):
    opt = load_options(args.opt)
    opt.setdefault('path', {})
    opt['path']['visualization'] = osp.join(root_path, 'results', opt['name'], 'visualization')
    device = 'cuda' if torch.cuda.is_available() else 'cpu'
    pipeclass = EDLoRAPipeline if opt['models']['enable_edlora'] else StableDiffusionPipeline     # reference :90
    pipe = pipeclass.from_pretrained(opt['models']['pretrained_path'], torch_dtype=torch.float16).to(device)
    pipe, cfg = convert_edlora(pipe, torch.load(opt['path']['lora_path'], weights_only=False),
                               enable_edlora=opt['models']['enable_edlora'], alpha=opt['models'].get('alpha', 1.0))
    pipe.set_new_concept_cfg(cfg)
    valset_cfg = opt['datasets']['val_vis']
    loader = torch.utils.data.DataLoader(PromptDataset(valset_cfg), batch_size=valset_cfg['batch_size_per_gpu'])
    visual_validation(pipe, loader, f"validation_{opt['models'].get('alpha', 1.0)}", opt)


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('-opt', type=str, required=True)
    test(osp.abspath(osp.join(__file__, osp.pardir)), parser.parse_args())
