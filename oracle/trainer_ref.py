"""ORACLE — test infrastructure. The reference training forward (mixofshow/pipelines/trainer_edlora.py:202-261)
and one optimisation step (train_edlora.py:105-143) restated on CPU in plain torch.

The non-attention plumbing (UNet skeleton, CLIP tower, VAE, DDPM add_noise) is shared with the product —
BASELINE.json's north_star asks for "the same non-attention ops as the comparison path" — while every piece of the
hot path is the oracle's: LoRALinearLayerRef on each LoRA site, EDLoRA(_Control)_AttnProcessorRef on attn2,
PlainAttnProcessorRef on attn1, AttentionStoreRef keeping full (B*H, N, 77) maps and cal_attn_reg_ref.
This is also the "CPU diffusers reference path" stand-in that bench.py times as `cpu_baseline`.
"""
import torch
import torch.nn.functional as F

from oracle import edlora_ref as R


def make_reference_twin(trainer, device='cpu', dtype=torch.float32, round_frozen_to=None):
    """Rebuild the trainer's modules from the same (synthetic, seeded or on-disk) weights, copy its trainable
    state, and put the ORACLE attention path on them. Returns a dict of modules + metadata."""
    import mos_path  # noqa: F401
    from mixofshow.utils import pretrained
    src = trainer.pretrained_path
    unet = pretrained.load_unet(src).to(device, dtype)
    te = pretrained.load_text_encoder(src).to(device, dtype)
    vae = pretrained.load_vae(src).to(device, dtype)
    for p in list(unet.parameters()) + list(te.parameters()) + list(vae.parameters()):
        p.requires_grad_(False)
    if round_frozen_to is not None:
        # fp16-emulating mode: autocast rounds the frozen Linear/Conv weights and biases to half at every
        # use; the twin keeps fp32 arithmetic but sees the same weight VALUES
        for root in (unet, te, vae):
            for m in root.modules():
                if isinstance(m, (torch.nn.Linear, torch.nn.Conv2d)):
                    for p in (m.weight, m.bias):
                        if p is not None:
                            p.data = p.data.to(round_frozen_to).to(dtype)
    te.resize_token_embeddings(len(trainer.tokenizer))
    concept = torch.nn.Parameter(trainer.concept_embedding.detach().to(device, dtype).clone())
    emb = te.text_model.embeddings
    object.__setattr__(emb, 'concept_rows', concept)
    emb.concept_base = trainer.get_all_concept_token_ids()[0]

    def wrap(root, loras):
        out = []
        mods = dict(root.named_modules())
        for lora in loras:
            ref = R.LoRALinearLayerRef(lora.name, mods[lora.name], rank=lora.lora_down.weight.shape[0],
                                       alpha=float(lora.alpha))
            ref.lora_down.weight.data.copy_(lora.lora_down.weight.detach().to(device, dtype))
            ref.lora_up.weight.data.copy_(lora.lora_up.weight.detach().to(device, dtype))
            out.append(ref.to(device, dtype))
        return out

    te_lora = wrap(te, trainer.text_encoder_lora)
    unet_lora = wrap(unet, trainer.unet_lora)
    store = R.AttentionStoreRef(training=True) if trainer.attn_reg_weight is not None else None
    for m in unet.modules():
        if m.__class__.__name__ == 'Attention':
            m.set_processor(R.PlainAttnProcessorRef())
    R.install_ref_processors(unet, controller=store, control=store is not None)
    return dict(unet=unet, text_encoder=te, vae=vae, concept=concept, te_lora=te_lora, unet_lora=unet_lora,
                store=store, trainer=trainer)


def reference_forward(twin, images, prompts, masks, img_masks, noise=None, timesteps=None, latents=None,
                      latent_noise=None):
    """trainer_edlora.py:202-261."""
    tr = twin['trainer']
    if latents is None:
        latents = twin['vae'].encode(images).latent_dist.sample(noise=latent_noise) * 0.18215
    bsz = latents.shape[0]
    if noise is None:
        noise = torch.randn_like(latents)
        if tr.noise_offset is not None:
            noise = noise + tr.noise_offset * torch.randn((bsz, latents.shape[1], 1, 1), device=latents.device)
    if timesteps is None:
        timesteps = torch.randint(0, tr.scheduler.config.num_train_timesteps, (bsz, ), device=latents.device)
    noisy = tr.scheduler.add_noise(latents, noise.to(latents.dtype), timesteps.long())
    if tr.enable_edlora:
        prompts = R.bind_concept_prompt_ref(prompts, tr.new_concept_cfg)
    ids = tr.tokenizer(prompts, padding='max_length', max_length=tr.tokenizer.model_max_length, truncation=True,
                       return_tensors='pt').input_ids.to(latents.device)
    ehs = twin['text_encoder'](ids)[0]
    if tr.enable_edlora:
        ehs = ehs.reshape(bsz, -1, *ehs.shape[1:])
    pred = twin['unet'](noisy, timesteps, ehs).sample
    assert tr.scheduler.config.prediction_type == 'epsilon'
    loss = R.masked_mse_ref(pred, noise, masks if tr.use_mask_loss else img_masks)
    if twin['store'] is not None:
        maps = twin['store'].get_average_attention()
        reg = R.cal_attn_reg_ref(maps, masks, ids, tr.get_all_concept_token_ids(), tr.attn_reg_weight,
                                 tr.reg_full_identity, strict_resolutions=False)
        if not torch.isnan(reg):
            loss = loss + reg
        twin['store'].reset()
    return loss


def twin_parameters(twin):
    """Same order as EDLoRATrainer.trainable_parameters(): concept rows, text-encoder LoRAs, UNet LoRAs."""
    ps = [twin['concept']]
    for l in twin['te_lora'] + twin['unet_lora']:
        ps += [l.lora_down.weight, l.lora_up.weight]
    return ps


def full_table_adamw_reference(table, concept_ids, grads_per_step, lr, weight_decay=0.01, betas=(0.9, 0.999)):
    """train_edlora.py:123-136 restated on a bare embedding table: AdamW over the WHOLE table, then every
    non-concept row is restored to its original value after each step. Returns the table after the steps."""
    table = torch.nn.Parameter(table.clone())
    original = table.detach().clone()
    opt = torch.optim.AdamW([table], lr=lr, weight_decay=weight_decay, betas=betas)
    keep = torch.ones(table.shape[0], dtype=torch.bool)
    keep[concept_ids] = False
    for g in grads_per_step:
        table.grad = g.clone()
        opt.step()
        opt.zero_grad()
        with torch.no_grad():
            table[keep] = original[keep]
    return table.detach()
