"""ED-LoRA sampling pipeline — same entry points as reference mixofshow/pipelines/pipeline_edlora.py:
`bind_concept_prompt` (:18-29), `EDLoRAPipeline` (:32-322) and the `StableDiffusionPipeline` name that
train_edlora.py:18 imports. diffusers is not available, so `StableDiffusionPipeline` here is a minimal local
base class (component registry, device/dtype moves, latent preparation / decoding, save/load in the diffusers
directory layout); the denoising loop drives the local UNet whose attention layers run on the HIP kernels.
"""
import json
import os
from types import SimpleNamespace

import numpy as np
import torch

from mixofshow.models.edlora import (revise_edlora_unet_attention_controller_forward,
                                     revise_edlora_unet_attention_forward)
from mixofshow.utils import pretrained


def bind_concept_prompt(prompts, new_concept_cfg):
    """Expand each prompt into 16 per-layer prompts; in copy i every concept name becomes its i-th token."""
    if isinstance(prompts, str):
        prompts = [prompts]
    bound = []
    for prompt in prompts:
        layers = [prompt] * 16
        for concept_name, cfg in new_concept_cfg.items():
            layers = [p.replace(concept_name, tok) for p, tok in zip(layers, cfg['concept_token_names'])]
        bound.extend(layers)
    return bound


class StableDiffusionPipelineOutput(SimpleNamespace):
    pass


class StableDiffusionPipeline:
    """Minimal stand-in for diffusers.StableDiffusionPipeline (text-to-image, no safety checker)."""

    components = ('vae', 'text_encoder', 'tokenizer', 'unet', 'scheduler')

    def __init__(self, vae, text_encoder, tokenizer, unet, scheduler, safety_checker=None, feature_extractor=None,
                 requires_safety_checker=False):
        self.register_modules(vae=vae, text_encoder=text_encoder, tokenizer=tokenizer, unet=unet, scheduler=scheduler)
        self.safety_checker = None
        self.vae_scale_factor = 2**(len(self.vae.config.block_out_channels) - 1)
        self._progress = {}

    def register_modules(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    @classmethod
    def from_pretrained(cls, pretrained_path, scheduler=None, torch_dtype=None, **kw):
        pipe = cls(vae=pretrained.load_vae(pretrained_path), text_encoder=pretrained.load_text_encoder(pretrained_path),
                   tokenizer=pretrained.load_tokenizer(pretrained_path), unet=pretrained.load_unet(pretrained_path),
                   scheduler=scheduler or pretrained.load_scheduler(pretrained_path, 'dpm'))
        cfg = os.path.join(str(pretrained_path), 'new_concept_cfg.json')
        if torch_dtype is not None:
            pipe.to(dtype=torch_dtype)
        if os.path.isfile(cfg) and hasattr(pipe, 'set_new_concept_cfg'):
            pass  # callers load it explicitly (regionally_controlable_sampling.py:57-60)
        return pipe

    def save_pretrained(self, path):
        """diffusers directory layout; weights as safetensors (reference gradient_fusion.py:810-811)."""
        from safetensors.torch import save_file
        for name, fname in (('unet', 'diffusion_pytorch_model.safetensors'), ('vae', 'diffusion_pytorch_model.safetensors'),
                            ('text_encoder', 'model.safetensors')):
            os.makedirs(os.path.join(path, name), exist_ok=True)
            sd = {k: v.detach().cpu().contiguous() for k, v in getattr(self, name).state_dict().items()}
            save_file(sd, os.path.join(path, name, fname))
            pretrained.save_model_config(getattr(self, name), os.path.join(path, name))
        os.makedirs(os.path.join(path, 'tokenizer'), exist_ok=True)
        added = getattr(self.tokenizer, 'added', None)
        if added is not None:
            with open(os.path.join(path, 'tokenizer', 'added_tokens.json'), 'w') as f:
                json.dump(added, f)
        elif hasattr(self.tokenizer, 'save_pretrained'):
            self.tokenizer.save_pretrained(os.path.join(path, 'tokenizer'))

    def to(self, device=None, dtype=None):
        if isinstance(device, torch.dtype):
            device, dtype = None, device
        for name in ('vae', 'text_encoder', 'unet'):
            m = getattr(self, name)
            if device is not None:
                m.to(device)
            if dtype is not None:
                m.to(dtype)
        if device is not None and torch.device(device).type == 'cuda':
            # NHWC on the device: the token-major attention path and MIOpen's fp16 convolutions are channels-last
            self.unet.to(memory_format=torch.channels_last)
        for extra in ('keypose_adapter', 'sketch_adapter'):
            m = getattr(self, extra, None)
            if m is not None:
                m.to(device=device, dtype=dtype)
        return self

    @property
    def device(self):
        return next(self.unet.parameters()).device

    @property
    def _execution_device(self):
        return self.device

    def set_progress_bar_config(self, **kw):
        self._progress = kw

    def progress_bar(self, total=None):
        from tqdm import tqdm
        return tqdm(total=total, disable=self._progress.get('disable', True))

    def check_inputs(self, prompt, height, width, callback_steps, negative_prompt=None, prompt_embeds=None,
                     negative_prompt_embeds=None):
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f'`height` and `width` have to be divisible by 8 but are {height} and {width}.')
        if prompt is None and prompt_embeds is None:
            raise ValueError('Provide either `prompt` or `prompt_embeds`.')
        if prompt is not None and prompt_embeds is not None:
            raise ValueError('Cannot forward both `prompt` and `prompt_embeds`.')

    def prepare_latents(self, batch_size, num_channels, height, width, dtype, device, generator, latents=None):
        shape = (batch_size, num_channels, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if latents is None:
            gdev = generator.device if generator is not None else device
            latents = torch.randn(shape, generator=generator, device=gdev, dtype=dtype).to(device)
        else:
            latents = latents.to(device=device, dtype=dtype)
        return latents * self.scheduler.init_noise_sigma

    def prepare_extra_step_kwargs(self, generator, eta):
        return {}

    def decode_latents(self, latents):
        latents = latents / self.vae.config.scaling_factor
        image = self.vae.decode(latents.to(self.vae.dtype)).sample
        image = (image / 2 + 0.5).clamp(0, 1)
        return image.cpu().permute(0, 2, 3, 1).float().numpy()

    @staticmethod
    def numpy_to_pil(images):
        from PIL import Image
        if images.ndim == 3:
            images = images[None]
        return [Image.fromarray((im * 255).round().astype(np.uint8)) for im in images]


    # ---- sampling (shared by the plain and the ED-LoRA pipeline) -------------------------------------------
    _requires_concept_cfg = False
    new_concept_cfg = None

    def set_new_concept_cfg(self, new_concept_cfg=None):
        self.new_concept_cfg = new_concept_cfg

    def _encode_prompt(self, prompt, new_concept_cfg, device, num_images_per_prompt, do_classifier_free_guidance,
                       negative_prompt=None, prompt_embeds=None, negative_prompt_embeds=None):
        """ED-LoRA: (B,16,77,768) layer-wise embeddings, negative embeddings repeated over the 16 layers and
        concatenated in front for classifier-free guidance (reference :111-190). Plain LoRA / no concepts (one token
        per concept word, or no table): ordinary (B,77,768) embeddings, what diffusers' StableDiffusionPipeline —
        the class the reference picks when `enable_edlora` is false (test_edlora.py:90) — computes."""
        assert num_images_per_prompt == 1, 'only support num_images_per_prompt=1 now'
        if isinstance(prompt, str):
            batch_size = 1
        elif isinstance(prompt, list):
            batch_size = len(prompt)
        else:
            batch_size = prompt_embeds.shape[0]
        tok = self.tokenizer
        if prompt_embeds is None:
            if new_concept_cfg:
                texts = bind_concept_prompt(prompt, new_concept_cfg)      # 16 (ED-LoRA) or 1 (LoRA) per prompt
            else:
                texts = [prompt] if isinstance(prompt, str) else list(prompt)
            ids = tok(texts, padding='max_length', max_length=tok.model_max_length, truncation=True,
                      return_tensors='pt').input_ids
            prompt_embeds = self.text_encoder(ids.to(device))[0]
            prompt_embeds = prompt_embeds.reshape(batch_size, -1, *prompt_embeds.shape[1:])
        prompt_embeds = prompt_embeds.to(dtype=self.text_encoder.dtype, device=device)
        if prompt_embeds.dim() == 3:
            prompt_embeds = prompt_embeds[:, None]
        _, layer_num, seq_len, _ = prompt_embeds.shape
        if do_classifier_free_guidance and negative_prompt_embeds is None:
            if negative_prompt is None:
                uncond = [''] * batch_size
            elif type(prompt) is not type(negative_prompt):
                raise TypeError(f'`negative_prompt` should be the same type to `prompt`, but got {type(negative_prompt)}'
                                f' != {type(prompt)}.')
            elif isinstance(negative_prompt, str):
                uncond = [negative_prompt]
            elif batch_size != len(negative_prompt):
                raise ValueError(f'`negative_prompt` has batch size {len(negative_prompt)}, `prompt` has {batch_size}.')
            else:
                uncond = negative_prompt
            ids = tok(uncond, padding='max_length', max_length=seq_len, truncation=True, return_tensors='pt').input_ids
            negative_prompt_embeds = self.text_encoder(ids.to(device))[0]
        if do_classifier_free_guidance:
            n = negative_prompt_embeds.to(dtype=self.text_encoder.dtype, device=device)
            n = n.view(batch_size, 1, n.shape[1], -1).repeat(1, layer_num, 1, 1)
            prompt_embeds = torch.cat([n, prompt_embeds])
        return prompt_embeds[:, 0] if layer_num == 1 else prompt_embeds

    def clear_sampling_graphs(self):
        """Release the captured UNet graphs kept across calls (their memory pools stay pinned otherwise, e.g. between the
        validation calls of a training run)."""
        from mixofshow.utils import hipgraph as hipgraph_util
        return hipgraph_util.clear_sampling_graphs(self)

    @torch.no_grad()
    def __call__(self, prompt=None, height=None, width=None, num_inference_steps=50, guidance_scale=7.5,
                 negative_prompt=None, num_images_per_prompt=1, eta=0.0, generator=None, latents=None,
                 prompt_embeds=None, negative_prompt_embeds=None, output_type='pil', return_dict=True, callback=None,
                 callback_steps=1, cross_attention_kwargs=None, hipgraph=None):
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        self.check_inputs(prompt, height, width, callback_steps, negative_prompt, prompt_embeds, negative_prompt_embeds)
        batch_size = 1 if isinstance(prompt, str) else (len(prompt) if isinstance(prompt, list) else prompt_embeds.shape[0])
        device = self._execution_device
        do_cfg = guidance_scale > 1.0
        assert self.new_concept_cfg is not None or not self._requires_concept_cfg
        prompt_embeds = self._encode_prompt(prompt, self.new_concept_cfg, device, num_images_per_prompt, do_cfg,
                                            negative_prompt, prompt_embeds=prompt_embeds,
                                            negative_prompt_embeds=negative_prompt_embeds)
        self.scheduler.set_timesteps(num_inference_steps, device=device)
        timesteps = self.scheduler.timesteps
        latents = self.prepare_latents(batch_size * num_images_per_prompt, self.unet.in_channels, height, width,
                                       prompt_embeds.dtype, device, generator, latents)

        # hipGraph replay (`hipgraph=None` -> hipgraph_util.sampling_default(), on; not with an attention-recording
        # controller or forward hooks, which keep Python-side state per call): step 0 is eager, the UNet call is
        # captured at step 1 and replayed afterwards. The graph is kept ACROSS calls of the same shape (a validation loop
        # samples many prompts at one resolution): everything it reads besides (latents, t) is the prompt embedding -- a
        # static tensor that a later call refills in place, together with its layer-major copy -- so a later call replays
        # all steps instead of paying an eager step and a capture per prompt.
        from mixofshow.models import edlora
        from mixofshow.utils import hipgraph as hipgraph_util
        if hipgraph is None:
            hipgraph = hipgraph_util.sampling_default()
        hipgraph = (bool(hipgraph) and hipgraph_util.graphs_usable(device) and len(timesteps) >= 4
                    and not hasattr(self, 'controller') and not hipgraph_util.has_forward_hooks(self.unet)
                    and not hipgraph_util.has_python_controllers(self.unet))
        graphed, ent, replay_from = None, None, 1
        if hipgraph and cross_attention_kwargs is None:
            # (model epoch: see RegionallyT2IAdapterPipeline.__call__ -- weights, derived weight copies, processor objects)
            epoch = (hipgraph_util.model_epoch(self.unet),
                     tuple(id(getattr(m, 'processor', None)) for m in self.unet.modules() if hasattr(m, 'processor')))
            gkey = (tuple(prompt_embeds.shape), prompt_embeds.dtype, height, width, bool(do_cfg), tuple(latents.shape))
            cache = self.__dict__.setdefault('_sampling_graphs', {})
            ent = cache.get(gkey)
            if ent is not None and ent.epoch != epoch:
                cache.pop(gkey)
                ent = None
            if ent is None:
                ent = SimpleNamespace(pe=prompt_embeds.clone(), graphed=None, epoch=epoch)
                while len(cache) >= 2:                       # at most two shapes resident (graphs pin their memory pools)
                    cache.pop(next(iter(cache)))
                cache[gkey] = ent
            else:
                ent.pe.copy_(prompt_embeds)
                edlora.refresh_layer_major_states(ent.pe)
            prompt_embeds, graphed = ent.pe, ent.graphed
            if graphed is not None:
                replay_from = 0

        def unet_call(x, t):
            return self.unet(x, t, encoder_hidden_states=prompt_embeds,
                             cross_attention_kwargs=cross_attention_kwargs).sample

        self.last_call_graphed = graphed is not None
        self.last_call_replay_from = replay_from
        for i, t in enumerate(timesteps):
            model_in = torch.cat([latents] * 2) if do_cfg else latents
            model_in = self.scheduler.scale_model_input(model_in, t)
            if hipgraph and i == 1 and graphed is None:
                graphed = hipgraph_util.try_capture(unet_call, model_in, t)
                hipgraph = graphed is not None
                self.last_call_graphed = hipgraph
                if ent is not None:
                    ent.graphed = graphed
                    if graphed is None:
                        self._sampling_graphs.pop(gkey, None)
            use_graph = graphed is not None and i >= replay_from
            noise_pred = graphed(model_in, t) if use_graph else unet_call(model_in, t)
            if do_cfg:
                uncond, text = noise_pred.chunk(2)
                noise_pred = uncond + guidance_scale * (text - uncond)
            latents = self.scheduler.step(noise_pred, t, latents).prev_sample
            if hasattr(self, 'controller'):
                latents = self.controller.step_callback(latents).to(latents.dtype)
            if callback is not None and i % callback_steps == 0:
                callback(i, t, latents)
        if output_type == 'latent':
            image = latents
        else:
            image = self.decode_latents(latents)
            if output_type == 'pil':
                image = self.numpy_to_pil(image)
        if not return_dict:
            return image
        return StableDiffusionPipelineOutput(images=image, nsfw_content_detected=None)


class EDLoRAPipeline(StableDiffusionPipeline):

    def __init__(self, vae, text_encoder, tokenizer, unet, scheduler, safety_checker=None, feature_extractor=None,
                 requires_safety_checker=False):
        super().__init__(vae, text_encoder, tokenizer, unet, scheduler)
        revise_edlora_unet_attention_forward(unet)  # reference :93
        self.new_concept_cfg = None

    _requires_concept_cfg = True

    def set_controller(self, controller):
        self.controller = controller
        revise_edlora_unet_attention_controller_forward(self.unet, controller)

