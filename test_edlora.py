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


class _SingleProcess:
    """What `visual_validation` needs of an `accelerate.Accelerator` (the reference passes one, test_edlora.py:24): the product
    launches one process per GPU with torch.distributed (mixofshow.parallel.dp) instead."""

    @property
    def is_main_process(self):
        from mixofshow.parallel import dp
        return dp.get_rank() == 0

    def wait_for_everyone(self):
        from mixofshow.parallel import dp
        dp.barrier()


def visual_validation(accelerator, pipe, dataloader, current_iter, opt):
    """reference test_edlora.py:24-57, same signature and file layout: every prompt of the validation set sampled from its fixed
    latents with the reference's negative prompt, one PNG per sample at
    `<visualization>/<dataset name>/<current_iter>/<prompt>---G_<scale>_S_<steps>---<index>---<current_iter>.png`, then (option
    val.compose_visualize) the comparison grid. `accelerator`: anything with `wait_for_everyone()` / `is_main_process`, or None."""
    from mixofshow.utils.util import NEGATIVE_PROMPT, compose_visualize, pil_imwrite
    accelerator = accelerator if accelerator is not None else _SingleProcess()
    dataset_name = dataloader.dataset.opt['name']
    pipe.unet.eval()
    pipe.text_encoder.eval()
    steps = opt['val']['sample'].get('num_inference_steps', 50)
    scale = opt['val']['sample'].get('guidance_scale', 7.5)
    save_img_path = None
    for val_data in dataloader:
        prompts = list(val_data['prompts'])
        latents = val_data['latents'].to(pipe.device, torch.float16) if opt['val'].get('use_fixed_latents', True) else None
        output = pipe(prompt=prompts, latents=latents, negative_prompt=[NEGATIVE_PROMPT] * len(prompts),
                      num_inference_steps=steps, guidance_scale=scale).images
        for img, prompt, indice in zip(output, prompts, val_data['indices']):
            img_name = f"{prompt.replace(' ', '_')}---G_{scale}_S_{steps}---{indice}"
            save_img_path = osp.join(opt['path']['visualization'], dataset_name, f'{current_iter}',
                                     f'{img_name}---{current_iter}.png')
            pil_imwrite(img, save_img_path)
        del output
    if hasattr(pipe, 'clear_sampling_graphs'):
        pipe.clear_sampling_graphs()          # the graphs kept across the prompts of one validation run are not needed afterwards
    accelerator.wait_for_everyone()
    if opt['val'].get('compose_visualize') and save_img_path is not None and accelerator.is_main_process:
        compose_visualize(osp.dirname(save_img_path))


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
    visual_validation(None, pipe, loader, f"validation_{opt['models'].get('alpha', 1.0)}", opt)       # reference :101


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('-opt', type=str, required=True)
    test(osp.abspath(osp.join(__file__, osp.pardir)), parser.parse_args())
