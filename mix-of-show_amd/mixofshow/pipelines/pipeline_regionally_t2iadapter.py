"""Regionally controllable sampling on the HIP path — same names as the reference's
mixofshow/pipelines/pipeline_regionally_t2iadapter.py: `RegionT2I_AttnProcessor` (:27-145),
`revise_regionally_t2iadapter_attention_forward` (:148-163), `RegionallyT2IAdapterPipeline` (:166-608).

The reference, per cross-attention layer and per denoising step: recomputes every region's K/V, builds an overlap
count mask ON THE CPU, indexes CUDA tensors with it (nonzero + H2D + sync), runs an einsum/softmax/einsum per
region on a cropped query box and scatter-adds the result (:32-86) — ~(2+R) host syncs x 16 layers x 50 steps.
Here: box rounding happens once on the host (pure integer arithmetic, identical ceil/floor rule), the K/V of the
context prompt and of all regions come from ONE fused [to_k; to_v] GEMM per layer and are cached across the 50
steps (they are step-invariant), and one kernel (`mos_region_cross_attn_fwd`) computes, for every query, the base
attention or the count-normalised sum of the covering regions' attentions — no masks, no syncs, no scatter.
"""
import ast
import math
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from mixofshow.hip import functional as F_hip
from mixofshow.models.attention import _check_plain, fused_attention_layer, project
from mixofshow.pipelines.pipeline_edlora import StableDiffusionPipeline, bind_concept_prompt
from mixofshow.utils import hipgraph as hipgraph_util


def region_feature_boxes(region_fracs, feat_h, feat_w):
    """ceil on the starts, floor on the ends of fractional coords x feature size (reference :38-39, :68-69)."""
    return [(math.ceil(b[0] * feat_h), math.ceil(b[1] * feat_w), math.floor(b[2] * feat_h), math.floor(b[3] * feat_w))
            for b in region_fracs]


class RegionT2I_AttnProcessor:

    def __init__(self, cross_attention_idx, attention_op=None):
        self.attention_op = attention_op
        self.cross_attention_idx = cross_attention_idx
        self._kv_key = None
        self._kv = None
        self._cd = None

    def reset_cache(self):
        self._kv_key = None
        self._kv = None

    def refresh_source_kv(self, attn, encoder_hidden_states, region_list, cd):
        """Recompute the cached source K/V for new prompt / region embeddings WITHOUT running the layer: what step 0 of a call
        does as a side effect. The pipeline calls it before replaying a graph captured by an earlier call (the graph reads the
        buffer `_kv` by address; `_source_kv` refills it in place)."""
        with torch.no_grad():
            return self._source_kv(attn, self._layer_states(encoder_hidden_states), region_list, cd)

    def _layer_states(self, states):
        return states[:, self.cross_attention_idx] if states.dim() == 4 else states

    def _source_kv(self, attn, context, region_list, cd):
        """(S, B, 77, 2C) keys|values of [context, region_1..R] — one GEMM, cached while the inputs are unchanged."""
        srcs = [context] + [self._layer_states(r[0]) for r in region_list]
        key = tuple((s.data_ptr(), s._version, tuple(s.shape)) for s in srcs) + (
            attn.to_k.weight._version, attn.to_v.weight._version, attn.to_k.weight.data_ptr(), cd)
        if key != self._kv_key:
            stacked = torch.stack([s.to(cd) for s in srcs])                     # (S, B, 77, Cc)
            S, B, M, Cc = stacked.shape
            kv = project(attn, 'kv', [attn.to_k, attn.to_v], stacked.reshape(S * B, M, Cc), cd).reshape(S, B, M, -1)
            if self._kv is not None and self._kv.shape == kv.shape and self._kv.dtype == kv.dtype:
                self._kv.copy_(kv)               # in place: a captured graph may hold this address (the pipeline's graph
                                                 # cache re-attaches the buffer, stale, before step 0 of a call)
            else:
                self._kv = kv
            self._kv_key = key
        return self._kv

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 **cross_attention_kwargs):
        _check_plain(attn)
        assert attention_mask is None
        if encoder_hidden_states is None:
            return fused_attention_layer(attn, hidden_states, None)[0]
        region_list = cross_attention_kwargs['region_list']          # KeyError if absent, like the reference (:120)
        height, width = cross_attention_kwargs['height'], cross_attention_kwargs['width']
        context = self._layer_states(encoder_hidden_states)
        if len(region_list) == 0:
            return fused_attention_layer(attn, hidden_states, context)[0]
        n_tok = hidden_states.shape[1]
        downscale = math.sqrt(height * width / n_tok)
        feat_h, feat_w = int(height // downscale), int(width // downscale)
        assert feat_h * feat_w == n_tok, f'{n_tok} tokens do not form a {feat_h}x{feat_w} map'
        cd = F_hip.compute_dtype_for(hidden_states)
        self._cd = cd
        with torch.no_grad():
            kv = self._source_kv(attn, context, region_list, cd)
        C = kv.shape[-1] // 2
        region = dict(k_src=kv[..., :C], v_src=kv[..., C:], feat_h=feat_h, feat_w=feat_w,
                      boxes=region_feature_boxes([r[-1] for r in region_list], feat_h, feat_w))
        return fused_attention_layer(attn, hidden_states, context, region=region)[0]


def revise_regionally_t2iadapter_attention_forward(unet):
    """Every Attention (attn1 AND attn2) gets the processor; the layer index advances on attn2 (reference :148-163)."""

    def visit(module, count):
        for name, layer in module.named_children():
            if layer.__class__.__name__ == 'Attention':
                layer.set_processor(RegionT2I_AttnProcessor(count))
                if 'attn2' in name:
                    count += 1
            else:
                count = visit(layer, count)
        return count

    n = visit(unet.down_blocks, 0)
    n = visit(unet.mid_block, n)
    return visit(unet.up_blocks, n)


# ---- T2I-Adapter (diffusers `T2IAdapter(adapter_type='full_adapter')`, restated; plumbing) ---------------------
class _AdapterResnetBlock(nn.Module):

    def __init__(self, c):
        super().__init__()
        self.block1 = nn.Conv2d(c, c, 3, padding=1)
        self.act = nn.ReLU()
        self.block2 = nn.Conv2d(c, c, 1)

    def forward(self, x):
        return self.block2(self.act(self.block1(x))) + x


class _AdapterBlock(nn.Module):

    def __init__(self, cin, cout, n_res, down):
        super().__init__()
        self.downsample = nn.AvgPool2d(2, stride=2, ceil_mode=True) if down else None
        self.in_conv = nn.Conv2d(cin, cout, 1) if cin != cout else None
        self.resnets = nn.Sequential(*[_AdapterResnetBlock(cout) for _ in range(n_res)])

    def forward(self, x):
        if self.downsample is not None:
            x = self.downsample(x)
        if self.in_conv is not None:
            x = self.in_conv(x)
        return self.resnets(x)


class _FullAdapter(nn.Module):

    def __init__(self, in_channels=3, channels=(320, 640, 1280, 1280), num_res_blocks=2, downscale_factor=8):
        super().__init__()
        self.unshuffle = nn.PixelUnshuffle(downscale_factor)
        self.conv_in = nn.Conv2d(in_channels * downscale_factor**2, channels[0], 3, padding=1)
        self.body = nn.ModuleList([_AdapterBlock(channels[0], channels[0], num_res_blocks, False)] + [
            _AdapterBlock(channels[i - 1], channels[i], num_res_blocks, True) for i in range(1, len(channels))])
        self.total_downscale_factor = downscale_factor * 2**(len(channels) - 1)

    def forward(self, x):
        x = self.conv_in(self.unshuffle(x))
        feats = []
        for blk in self.body:
            x = blk(x)
            feats.append(x)
        return feats


class T2IAdapter(nn.Module):

    def __init__(self, in_channels=3, channels=(320, 640, 1280, 1280), num_res_blocks=2, downscale_factor=8):
        super().__init__()
        self.adapter = _FullAdapter(in_channels, channels, num_res_blocks, downscale_factor)

    @property
    def dtype(self):
        return self.adapter.conv_in.weight.dtype

    def forward(self, x):
        return self.adapter(x)


def _preprocess_adapter_image(image, height, width):
    """PIL image(s) / tensor -> (B, C, H, W) float in [0, 1] (diffusers `_preprocess_adapter_image`)."""
    import numpy as np
    if torch.is_tensor(image):
        return image
    if not isinstance(image, (list, tuple)):
        image = [image]
    arrs = []
    for im in image:
        a = np.array(im.resize((width, height)))
        a = a[None, ..., None] if a.ndim == 2 else a[None]
        arrs.append(a)
    a = np.concatenate(arrs, 0).astype(np.float32) / 255.0
    return torch.from_numpy(a.transpose(0, 3, 1, 2))


class StableDiffusionAdapterPipelineOutput(SimpleNamespace):
    pass


class RegionallyT2IAdapterPipeline(StableDiffusionPipeline):

    def __init__(self, vae, text_encoder, tokenizer, unet, scheduler, safety_checker=None, feature_extractor=None,
                 requires_safety_checker=False):
        super().__init__(vae, text_encoder, tokenizer, unet, scheduler)
        self.new_concept_cfg = None
        self.keypose_adapter = None   # attached after construction (regionally_controlable_sampling.py:62-63)
        self.sketch_adapter = None
        revise_regionally_t2iadapter_attention_forward(self.unet)

    def set_new_concept_cfg(self, new_concept_cfg=None):
        self.new_concept_cfg = new_concept_cfg

    def _encode(self, prompts, device):
        tok = self.tokenizer
        ids = tok(prompts, padding='max_length', max_length=tok.model_max_length, truncation=True,
                  return_tensors='pt').input_ids
        return self.text_encoder(ids.to(device), attention_mask=None)[0]

    def _encode_region_prompt(self, prompt, new_concept_cfg, device, num_images_per_prompt, do_classifier_free_guidance,
                              negative_prompt=None, prompt_embeds=None, negative_prompt_embeds=None, height=512, width=512):
        """reference :215-299 — context (2,16,77,768) = cat[neg, pos]; per region (cat[neg_r, pos_r], frac box)."""
        batch_size = 1 if isinstance(prompt, str) else (len(prompt) if isinstance(prompt, list) else prompt_embeds.shape[0])
        assert batch_size == 1, 'only sample one prompt once in this version'
        region_list = []
        if prompt_embeds is None:
            context_prompt, regions = prompt[0][0], prompt[0][1]
            # ONE text-encoder forward over all prompts of the call (the reference runs 2 + 2R of them, :237-296: the same
            # rows, a sequence's embedding does not depend on its batch neighbours) -- an eager CLIP forward is ~130 launches
            groups, texts, spans = [], [], {}

            def want(strs):
                strs = [strs] if isinstance(strs, str) else list(strs)
                key = tuple(strs)
                if key not in spans:                         # (the negative prompts of the regions are usually one string)
                    spans[key] = (len(texts), len(texts) + len(strs))
                    texts.extend(strs)
                groups.append(spans[key])

            want(bind_concept_prompt([context_prompt], new_concept_cfg))
            want(negative_prompt if negative_prompt is not None else [''] * batch_size)
            for region_prompt, region_neg, _ in regions:
                want(bind_concept_prompt([region_prompt], new_concept_cfg))
                want(region_neg if region_neg is not None else [''] * batch_size)
            emb = self._encode(texts, device)
            parts = [emb[a:b] for a, b in groups]
            pos = parts[0].reshape(batch_size, -1, *emb.shape[1:])
            layer_num, seq_len = pos.shape[1], pos.shape[2]
            neg = parts[1].view(batch_size, 1, seq_len, -1).repeat(1, layer_num, 1, 1)
            prompt_embeds = torch.cat([neg, pos])
            for k, (_, _, box) in enumerate(regions):
                rp = parts[2 + 2 * k].reshape(batch_size, -1, *emb.shape[1:])
                rn = parts[3 + 2 * k].view(batch_size, 1, seq_len, -1).repeat(1, layer_num, 1, 1)
                region_list.append((torch.cat([rn, rp]), box))
        return prompt_embeds, region_list

    def _adapter_states(self, adapter, adapter_input, weight, region_weight, height, width):
        """Region-weighted adapter features (reference :488-542)."""
        if adapter_input is None or adapter is None:
            return None
        x = _preprocess_adapter_image(adapter_input, height, width).to(self.device).to(adapter.dtype)
        feats = adapter(x)
        out = []
        for f in feats:
            w = weight * torch.ones(*f.shape[2:], dtype=f.dtype, device=f.device)
            if region_weight != '':
                for item in region_weight.split('|'):
                    region, rw = item.split('-')
                    region, rw = ast.literal_eval(region), float(ast.literal_eval(rw))   # grammar of reference :500-501
                    fh, fw = f.shape[2:]
                    h0, w0, h1, w1 = region
                    h0, h1, w0, w1 = h0 / height, h1 / height, w0 / width, w1 / width
                    w[math.ceil(h0 * fh):math.floor(h1 * fh), math.ceil(w0 * fw):math.floor(w1 * fw)] = rw
            out.append(w * f)
        return out

    def clear_sampling_graphs(self):
        """Release the captured UNet graphs kept across calls (their memory pools stay pinned otherwise, e.g. between the
        validation calls of a training run)."""
        from mixofshow.utils import hipgraph as hipgraph_util
        return hipgraph_util.clear_sampling_graphs(self)

    @torch.no_grad()
    def __call__(self, prompt=None, keypose_adapter_input=None, keypose_adaptor_weight=1.0,
                 region_keypose_adaptor_weight='', sketch_adapter_input=None, sketch_adaptor_weight=1.0,
                 region_sketch_adaptor_weight='', height=None, width=None, num_inference_steps=50, guidance_scale=7.5,
                 negative_prompt=None, num_images_per_prompt=1, eta=0.0, generator=None, latents=None,
                 prompt_embeds=None, negative_prompt_embeds=None, output_type='pil', return_dict=True, callback=None,
                 callback_steps=1, cross_attention_kwargs=None, adapter_states=None, hipgraph=None):
        device = self._execution_device
        self.check_inputs(prompt, height, width, callback_steps, negative_prompt, prompt_embeds, negative_prompt_embeds)
        batch_size = 1 if isinstance(prompt, str) else (len(prompt) if isinstance(prompt, list) else prompt_embeds.shape[0])
        do_cfg = guidance_scale > 1.0
        assert self.new_concept_cfg is not None
        prompt_embeds, region_list = self._encode_region_prompt(
            prompt, self.new_concept_cfg, device, num_images_per_prompt, do_cfg, negative_prompt,
            prompt_embeds=prompt_embeds, negative_prompt_embeds=negative_prompt_embeds, height=height, width=width)
        self.scheduler.set_timesteps(num_inference_steps, device=device)
        timesteps = self.scheduler.timesteps
        latents = self.prepare_latents(batch_size * num_images_per_prompt, self.unet.config.in_channels, height, width,
                                       prompt_embeds.dtype, device, generator, latents)
        # adapter features, once per call (reference :474-546). `adapter_states` lets callers pass precomputed
        # features (e.g. zeros / synthetic) when no adapter network is attached.
        if adapter_states is None:
            kp = self._adapter_states(self.keypose_adapter, keypose_adapter_input, keypose_adaptor_weight,
                                      region_keypose_adaptor_weight, height, width)
            sk = self._adapter_states(self.sketch_adapter, sketch_adapter_input, sketch_adaptor_weight,
                                      region_sketch_adaptor_weight, height, width)
            if kp is not None and sk is not None:
                adapter_states = [a + b for a, b in zip(kp, sk)]
            else:
                adapter_states = kp if kp is not None else sk
        if adapter_states is not None and do_cfg:
            adapter_states = [torch.cat([s] * 2, dim=0) if s.shape[0] == batch_size else s for s in adapter_states]
        if adapter_states is not None and getattr(self.unet, 'channels_last', False):
            # the layout of the activations they are added to, once per call (a mixed-layout add is a strided kernel, x 4 x 50)
            adapter_states = [s.contiguous(memory_format=torch.channels_last) for s in adapter_states]
        # hipGraph replay (`hipgraph=None` -> hipgraph_util.sampling_default(), on): step 0 runs eagerly (it fills the
        # per-layer source K/V caches), the UNet call is captured at step 1 and replayed from then on: 544 vs 774 ms per
        # 50-step sample. Not with forward hooks on the UNet (they would only run at capture time).
        if hipgraph is None:
            hipgraph = hipgraph_util.sampling_default()
        hipgraph = (bool(hipgraph) and hipgraph_util.graphs_usable(device) and len(timesteps) >= 4
                    and not hipgraph_util.has_forward_hooks(self.unet)
                    and not hipgraph_util.has_python_controllers(self.unet))
        # The captured graph is kept ACROSS calls of the same shape (prompts of the same layout, same boxes, same
        # resolution): everything it reads besides (latents, t) -- prompt embeddings, region embeddings, adapter features,
        # the processors' K/V caches -- lives in static tensors that a later call refreshes in place (step 0 of every call
        # still runs eagerly: it recomputes the K/V caches into the buffers the graph holds).
        graphed = None
        ent = None
        procs = [m.processor for m in self.unet.modules() if isinstance(getattr(m, 'processor', None), RegionT2I_AttnProcessor)]
        if hipgraph:
            # model epoch: the graph also bakes in the addresses of DERIVED buffers (fused / cast weight copies, fp32 affine
            # copies) that are re-allocated when a weight's version changes (load_state_dict of another fused model, an
            # in-place LoRA merge, .half()) and the K/V buffers of the processor objects it was captured with: a change of
            # any of them is a key miss, and the stale entry is dropped
            epoch = (hipgraph_util.model_epoch(self.unet), tuple(id(p) for p in procs))
            gkey = (tuple(prompt_embeds.shape), prompt_embeds.dtype, height, width, bool(do_cfg),
                    tuple((tuple(r[0].shape), tuple(float(v) for v in r[1])) for r in region_list),
                    None if adapter_states is None else tuple(tuple(a.shape) for a in adapter_states),
                    tuple(latents.shape))
            cache = self.__dict__.setdefault('_sampling_graphs', {})
            ent = cache.get(gkey)
            if ent is not None and ent.epoch != epoch:
                cache.pop(gkey)
                ent = None
            if ent is None:
                ent = SimpleNamespace(pe=prompt_embeds.clone(), rl=[(r[0].clone(), r[1]) for r in region_list],
                                      ad=None if adapter_states is None else [a.clone() for a in adapter_states], graphed=None,
                                      epoch=epoch)
                while len(cache) >= 2:                       # at most two shapes resident (graphs pin their memory pools)
                    cache.pop(next(iter(cache)))
                cache[gkey] = ent
            else:
                ent.pe.copy_(prompt_embeds)
                for (dst, _), (src, _) in zip(ent.rl, region_list):
                    dst.copy_(src)
                if ent.ad is not None:
                    for dst, src in zip(ent.ad, adapter_states):
                        dst.copy_(src)
            prompt_embeds, region_list, adapter_states, graphed = ent.pe, ent.rl, ent.ad, ent.graphed
        cak = {'region_list': region_list, 'height': height, 'width': width}
        # the per-layer source K/V caches are keyed on tensor identity (address / version / shape); a new call with new
        # prompts can re-use the very same addresses (caching allocator), so every call starts with stale caches
        for proc in procs:
            proc.reset_cache()
        replay_from = 1
        if graphed is not None:
            # the graph reads the K/V buffers it was captured with (an eager call in between gave the processors new ones):
            # hand them back, stale, and refill exactly those from this call's embeddings. With every cross-attention layer
            # refreshed that way step 0 is a replay as well (an eager UNet call is ~1.4 k launches, dispatcher-bound);
            # otherwise (no regions: the layers keep no source cache) step 0 runs eagerly as in the capturing call.
            for proc, buf, attn, cd in ent.kv:
                proc._kv = buf
                if len(region_list) > 0:
                    proc.refresh_source_kv(attn, prompt_embeds, region_list, cd)
            if len(region_list) > 0 and ent.kv and len(ent.kv) == ent.n_cross:
                replay_from = 0

        def unet_call(x, t):
            # (the UNet pops the LIST and adds the tensors to its activations -- `sample += r`, `x + r` -- it never writes to
            # them: a fresh list per call, no clones; the reference clones because diffusers' UNet may add in place into them)
            residuals = list(adapter_states) if adapter_states is not None else None
            return self.unet(x, t, encoder_hidden_states=prompt_embeds, cross_attention_kwargs=cak,
                             down_block_additional_residuals=residuals).sample

        self.last_call_graphed = graphed is not None
        self.last_call_replay_from = replay_from if graphed is not None else 1     # first step served by the graph
        for i, t in enumerate(timesteps):
            model_in = torch.cat([latents] * 2) if do_cfg else latents
            model_in = self.scheduler.scale_model_input(model_in, t)
            if hipgraph and i == 1 and graphed is None:
                graphed = hipgraph_util.try_capture(unet_call, model_in, t)
                hipgraph = graphed is not None
                self.last_call_graphed = hipgraph
                if ent is not None:
                    ent.graphed = graphed
                    ent.kv = [(m.processor, m.processor._kv, m, m.processor._cd) for m in self.unet.modules()
                              if isinstance(getattr(m, 'processor', None), RegionT2I_AttnProcessor)
                              and m.processor._kv is not None]                        # buffers owned by the entry
                    ent.n_cross = sum(1 for n, m in self.unet.named_modules()
                                      if isinstance(getattr(m, 'processor', None), RegionT2I_AttnProcessor) and n.endswith('attn2'))
                    if graphed is None:
                        self._sampling_graphs.pop(gkey, None)
            use_graph = graphed is not None and i >= replay_from
            noise_pred = graphed(model_in, t) if use_graph else unet_call(model_in, t)
            if do_cfg:
                uncond, text = noise_pred.chunk(2)
                noise_pred = uncond + guidance_scale * (text - uncond)
            latents = self.scheduler.step(noise_pred, t, latents).prev_sample
            if callback is not None and i % callback_steps == 0:
                callback(i, t, latents)
        if output_type == 'latent':
            image = latents
        else:
            image = self.decode_latents(latents)
            if output_type == 'pil':
                image = self.numpy_to_pil(image)
        if not return_dict:
            return (image, None)
        return StableDiffusionAdapterPipelineOutput(images=image, nsfw_content_detected=None)
