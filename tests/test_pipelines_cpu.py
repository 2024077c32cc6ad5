"""CPU tests of the sampling pipelines' HOST logic (prompt binding/encoding, region list construction, CFG, the
DPM-Solver++ loop, processor dispatch): the product pipelines on the 'tiny' preset in fp32 with the HIP primitives
emulated (fixture `emulated_hip`) against the SAME modules running the oracle's restatement of the reference
processors. The product rounds the attention operands to half (its kernels' compute type) even in an fp32 model, so
the two paths agree to a few half ulps per layer; with 4 steps on the tiny model that stays ~3e-4 of the latent range
(the 50-step fp16 GPU runs are dominated by chaotic amplification instead)."""
import torch


def _concept_cfg(tokenizer, text_encoder, names):
    cfg = {}
    for i, n in enumerate(names):
        toks = [f'<new{16 * i + l}>' for l in range(16)]
        tokenizer.add_tokens(toks)
        cfg[n] = {'concept_token_ids': [tokenizer.convert_tokens_to_ids(t) for t in toks], 'concept_token_names': toks}
    text_encoder.resize_token_embeddings(len(tokenizer))
    with torch.no_grad():
        g = torch.Generator().manual_seed(11)
        table = text_encoder.get_input_embeddings().weight
        table[49408:] = (torch.randn(table.shape[0] - 49408, table.shape[1], generator=g) * 0.02).to(table)
    return cfg


def _close(a, b, what):
    scale = max(1.0, b.abs().max().item())
    err = (a - b).abs().max().item()
    assert torch.isfinite(a).all() and err <= 2e-3 * scale, f'{what}: max|d|={err:.3e} on scale {scale:.2f}'


def test_edlora_pipeline_matches_reference_processors(emulated_hip):
    from mixofshow.pipelines.pipeline_edlora import EDLoRAPipeline
    from oracle import edlora_ref as R
    pipe = EDLoRAPipeline.from_pretrained('synthetic://tiny?seed=0', torch_dtype=torch.float32)
    pipe.set_new_concept_cfg(_concept_cfg(pipe.tokenizer, pipe.text_encoder, ['<potter1>', '<potter2>']))
    latents = torch.randn((2, 4, 8, 8), generator=torch.manual_seed(1))
    kw = dict(prompt=['a <potter1> <potter2> in the park', 'a photo of a dog'], negative_prompt=['blurry', ''],
              height=64, width=64, num_inference_steps=4, guidance_scale=7.5, output_type='latent')
    out = pipe(latents=latents.clone(), **kw).images
    assert not pipe.last_call_graphed                   # no device, no graph
    for m in pipe.unet.modules():
        if m.__class__.__name__ == 'Attention':
            m.set_processor(R.PlainAttnProcessorRef())
    R.install_ref_processors(pipe.unet)
    ref = pipe(latents=latents.clone(), **kw).images
    _close(out, ref, 'edlora pipeline latents')
    # guidance_scale <= 1 disables the CFG pair; a different step count changes the result
    one = pipe(latents=latents.clone(), **dict(kw, guidance_scale=1.0)).images
    assert one.shape == out.shape and not torch.allclose(one, ref)


def test_regional_pipeline_matches_reference_processors(emulated_hip):
    from mixofshow.pipelines.pipeline_regionally_t2iadapter import RegionallyT2IAdapterPipeline
    from oracle import region_ref
    H, W = 64, 96
    pipe = RegionallyT2IAdapterPipeline.from_pretrained('synthetic://tiny?seed=0', torch_dtype=torch.float32)
    pipe.set_new_concept_cfg(_concept_cfg(pipe.tokenizer, pipe.text_encoder,
                                          ['<potter1>', '<potter2>', '<hermione1>', '<hermione2>']))
    neg = 'lowres, bad anatomy'
    regions = [('a <potter1> <potter2>, in uniform', neg, [0.0, 0.0, 1.0, 0.45]),
               ('a <hermione1> <hermione2>, girl', neg, [0.1, 0.4, 0.9, 1.0]),       # overlaps region 1
               ('a castle', neg, [0.5, 0.2, 0.75, 0.7])]                               # covered by both others
    latents = torch.randn((1, 4, H // 8, W // 8), generator=torch.manual_seed(14))

    def run(prompt):
        return pipe(prompt=prompt, negative_prompt=[neg], height=H, width=W, num_inference_steps=4,
                    guidance_scale=7.5, latents=latents.clone(), output_type='latent').images

    with_regions = run([('three people near the castle', regions)])
    no_regions = run([('three people near the castle', [])])          # empty region list: plain cross attention
    region_ref.install_region_processors_ref(pipe.unet)
    _close(with_regions, run([('three people near the castle', regions)]), 'regional latents')
    _close(no_regions, run([('three people near the castle', [])]), 'regional latents, no regions')
    assert not torch.allclose(with_regions, no_regions)
