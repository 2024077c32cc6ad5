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


def test_regional_pipeline_with_t2i_adapters(emulated_hip):
    """Keypose + sketch T2I-Adapter features with per-region weights (reference :474-546): the adapters' residuals are
    added into the UNet's down path; weight 0 must reproduce the adapter-free sample, region weights must matter."""
    import numpy as np
    from PIL import Image
    from mixofshow.pipelines.pipeline_regionally_t2iadapter import RegionallyT2IAdapterPipeline, T2IAdapter
    H, W = 64, 96
    pipe = RegionallyT2IAdapterPipeline.from_pretrained('synthetic://tiny?seed=0', torch_dtype=torch.float32)
    pipe.set_new_concept_cfg(_concept_cfg(pipe.tokenizer, pipe.text_encoder, ['<potter1>', '<potter2>']))
    torch.manual_seed(0)
    pipe.keypose_adapter = T2IAdapter(in_channels=3, channels=(32, 64))
    pipe.sketch_adapter = T2IAdapter(in_channels=1, channels=(32, 64))
    rng = np.random.default_rng(0)
    pose = Image.fromarray(rng.integers(0, 255, (H, W, 3), dtype=np.uint8), 'RGB')
    sketch = Image.fromarray(rng.integers(0, 255, (H * 2, W * 2), dtype=np.uint8), 'L')     # resized by the pipeline
    prompt = [('two people', [('a <potter1> <potter2>', '', [0.0, 0.0, 1.0, 0.5])])]
    latents = torch.randn((1, 4, H // 8, W // 8), generator=torch.manual_seed(2))

    def run(**kw):
        return pipe(prompt=prompt, negative_prompt=[''], height=H, width=W, num_inference_steps=3, guidance_scale=7.5,
                    latents=latents.clone(), output_type='latent', **kw).images

    plain = run()
    zero = run(keypose_adapter_input=[pose], keypose_adaptor_weight=0.0, sketch_adapter_input=[sketch],
               sketch_adaptor_weight=0.0)
    torch.testing.assert_close(zero, plain, rtol=0, atol=0)
    both = run(keypose_adapter_input=[pose], keypose_adaptor_weight=1.0, sketch_adapter_input=[sketch],
               sketch_adaptor_weight=0.5)
    only_pose = run(keypose_adapter_input=[pose], keypose_adaptor_weight=1.0)
    regional = run(keypose_adapter_input=[pose], keypose_adaptor_weight=1.0,
                   region_keypose_adaptor_weight='[0, 0, 64, 48]-0.0')           # switch the left half off
    for x in (both, only_pose, regional):
        assert torch.isfinite(x).all() and not torch.allclose(x, plain)
    assert not torch.allclose(both, only_pose) and not torch.allclose(regional, only_pose)
    # precomputed adapter features take the same path
    kp = pipe._adapter_states(pipe.keypose_adapter, [pose], 1.0, '', H, W)
    torch.testing.assert_close(run(adapter_states=kp), only_pose, rtol=1e-5, atol=1e-5)


def test_region_kv_cache_is_dropped_at_every_call(emulated_hip):
    """the per-layer source K/V cache is keyed on tensor identity; a second call with other prompts may get
    the same addresses from the caching allocator. Every pipeline call must start with empty caches: poison them with
    a wrong entry under a key that WOULD match and check the result is unaffected."""
    from mixofshow.pipelines.pipeline_regionally_t2iadapter import RegionT2I_AttnProcessor, RegionallyT2IAdapterPipeline
    H, W = 64, 64
    pipe = RegionallyT2IAdapterPipeline.from_pretrained('synthetic://tiny?seed=0', torch_dtype=torch.float32)
    pipe.set_new_concept_cfg(_concept_cfg(pipe.tokenizer, pipe.text_encoder, ['<potter1>', '<potter2>']))
    latents = torch.randn((1, 4, H // 8, W // 8), generator=torch.manual_seed(3))

    def run(text):
        prompt = [('two people', [(text, '', [0.0, 0.0, 1.0, 0.6])])]
        return pipe(prompt=prompt, negative_prompt=[''], height=H, width=W, num_inference_steps=2, guidance_scale=7.5,
                    latents=latents.clone(), output_type='latent').images

    a = run('a <potter1> <potter2>')
    procs = [m.processor for m in pipe.unet.modules() if isinstance(getattr(m, 'processor', None), RegionT2I_AttnProcessor)]
    cached = [p for p in procs if p._kv is not None]
    assert cached, 'cross-attention layers cache their source K/V during a call'

    class _AlwaysEqual(tuple):
        def __eq__(self, other):
            return True

        def __ne__(self, other):
            return False

        __hash__ = tuple.__hash__

    for p in cached:                                  # stale entry whose key matches anything
        p._kv = torch.zeros_like(p._kv)
        p._kv_key = _AlwaysEqual()
    b = run('a <potter1> <potter2>')
    torch.testing.assert_close(a, b, rtol=0, atol=0)


def test_adapter_region_weighting_vs_reference_golden(golden):
    """(f2) T2I-Adapter region-weight rule: the product's `_adapter_states` + keypose/sketch sum against the outputs of the
    reference's own source lines (pipeline_regionally_t2iadapter.py:484-542, executed by tests/golden/make_golden.py)."""
    from mixofshow.pipelines.pipeline_regionally_t2iadapter import RegionallyT2IAdapterPipeline
    a = golden['adapter']
    H, W = a['height'], a['width']

    class _Fixed(torch.nn.Module):          # an "adapter" that returns the seeded features the golden was made with
        def __init__(self, feats):
            super().__init__()
            self.feats = feats
            self.p = torch.nn.Parameter(torch.zeros(1))

        @property
        def dtype(self):
            return torch.float32

        def forward(self, x):
            return [f.clone() for f in self.feats]

    pipe = RegionallyT2IAdapterPipeline.from_pretrained('synthetic://tiny?seed=0', torch_dtype=torch.float32)
    dummy = torch.zeros(1, 3, H, W)
    for name, case in a['cases'].items():
        c = case['spec']
        kp = pipe._adapter_states(_Fixed(a['keypose']) if c['kp'] else None, dummy if c['kp'] else None, c['kw'], c['rk'], H, W)
        sk = pipe._adapter_states(_Fixed(a['sketch']) if c['sk'] else None, dummy if c['sk'] else None, c['sw'], c['rs'], H, W)
        if kp is not None and sk is not None:
            got = [x + y for x, y in zip(kp, sk)]
        else:
            got = kp if kp is not None else sk
        assert len(got) == len(case['out']) == 4
        for g, r in zip(got, case['out']):
            torch.testing.assert_close(g, r, rtol=1e-6, atol=1e-6, msg=name)


def test_region_prompts_are_encoded_in_one_forward_with_the_rows_of_separate_forwards(emulated_hip):
    """`_encode_region_prompt` runs ONE text-encoder forward over every prompt of the call (the reference: 2 + 2R forwards,
    pipeline_regionally_t2iadapter.py:237-296); the rows must be the ones separate forwards give, in the reference's layout:
    context (2, 16, 77, C) = cat[neg, pos], per region (cat[neg_r, pos_r], box)."""
    from mixofshow.pipelines.pipeline_edlora import bind_concept_prompt
    from mixofshow.pipelines.pipeline_regionally_t2iadapter import RegionallyT2IAdapterPipeline
    pipe = RegionallyT2IAdapterPipeline.from_pretrained('synthetic://tiny?seed=0', torch_dtype=torch.float32)
    cfg = _concept_cfg(pipe.tokenizer, pipe.text_encoder, ['<potter1>', '<potter2>', '<hermione1>', '<hermione2>'])
    pipe.set_new_concept_cfg(cfg)
    neg = 'lowres, bad anatomy'
    regions = [('a <potter1> <potter2>, in uniform', neg, [0.0, 0.0, 1.0, 0.45]),
               ('a <hermione1> <hermione2>, girl', 'blurry', [0.1, 0.4, 0.9, 1.0]),
               ('a castle', None, [0.5, 0.2, 0.75, 0.7])]
    calls = []
    real = pipe.text_encoder.forward
    pipe.text_encoder.forward = lambda *a, **k: (calls.append(a[0].shape[0]), real(*a, **k))[1]
    emb, region_list = pipe._encode_region_prompt([('three people near the castle', regions)], cfg, 'cpu', 1, True, [neg])
    pipe.text_encoder.forward = real
    assert calls == [16 + 1 + 16 + 16 + 1 + 16 + 1]            # one forward; the repeated negative prompt is encoded once
    enc = lambda strs: pipe._encode(strs, 'cpu')               # noqa: E731
    pos = enc(bind_concept_prompt(['three people near the castle'], cfg))
    assert emb.shape == (2, 16, 77, pos.shape[-1]) and len(region_list) == 3
    torch.testing.assert_close(emb[1], pos, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(emb[0], enc([neg]).expand(16, -1, -1), rtol=1e-5, atol=1e-6)
    for (r_emb, box), (text, rneg, rbox) in zip(region_list, regions):
        assert box == rbox and r_emb.shape == emb.shape
        torch.testing.assert_close(r_emb[1], enc(bind_concept_prompt([text], cfg)), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(r_emb[0], enc([rneg if rneg is not None else '']).expand(16, -1, -1), rtol=1e-5, atol=1e-6)


def test_cached_sampling_graph_replays_step_0_after_an_explicit_kv_refresh(emulated_hip, monkeypatch):
    """A later call of the same layout refills the processors' source-K/V buffers from ITS embeddings
    (`refresh_source_kv`) and replays the captured UNet call for every step, step 0 included. The replay is emulated by a
    callable that -- like a hipGraph -- runs no host code of the processors' cache logic: it reads the K/V buffers that were
    current at capture time, whatever they hold. The second call must equal an eager call with ITS prompts."""
    from mixofshow.pipelines import pipeline_regionally_t2iadapter as P
    from mixofshow.utils import hipgraph as G
    H, W = 64, 96
    pipe = P.RegionallyT2IAdapterPipeline.from_pretrained('synthetic://tiny?seed=0', torch_dtype=torch.float32)
    pipe.set_new_concept_cfg(_concept_cfg(pipe.tokenizer, pipe.text_encoder, ['<potter1>', '<potter2>']))
    latents = torch.randn((1, 4, H // 8, W // 8), generator=torch.manual_seed(5))
    replays = []

    class _FakeGraph:
        def __init__(self, fn, *example):
            self.fn = fn
            self.frozen = [(m.processor, m.processor._kv) for m in pipe.unet.modules()
                           if isinstance(getattr(m, 'processor', None), P.RegionT2I_AttnProcessor) and m.processor._kv is not None]

        def __call__(self, x, t):
            replays.append(int(t))
            real = P.RegionT2I_AttnProcessor._source_kv
            bufs = {id(p): b for p, b in self.frozen}
            monkeypatch.setattr(P.RegionT2I_AttnProcessor, '_source_kv', lambda self_, *a: bufs[id(self_)])
            try:
                return self.fn(x, t)
            finally:
                monkeypatch.setattr(P.RegionT2I_AttnProcessor, '_source_kv', real)

    monkeypatch.setattr(G, 'graphs_usable', lambda device: True)
    monkeypatch.setattr(G, 'try_capture', lambda fn, *ex: _FakeGraph(fn, *ex))

    def run(text, graph):
        prompt = [('two people', [(text, '', [0.0, 0.0, 1.0, 0.6]), ('a dog', 'blurry', [0.2, 0.5, 0.9, 1.0])])]
        return pipe(prompt=prompt, negative_prompt=[''], height=H, width=W, num_inference_steps=5, guidance_scale=7.5,
                    latents=latents.clone(), output_type='latent', hipgraph=graph).images

    a_graph = run('a <potter1> <potter2>', True)
    n_first = len(replays)
    assert pipe.last_call_graphed and n_first == 4                 # capturing call: step 0 eager, steps 1..4 replayed
    b_graph = run('a red car', True)
    assert len(replays) - n_first == 5, 'a cached graph replays every step, step 0 included'
    b_eager = run('a red car', False)
    a_eager = run('a <potter1> <potter2>', False)
    torch.testing.assert_close(b_graph, b_eager, rtol=0, atol=0)
    torch.testing.assert_close(a_graph, a_eager, rtol=0, atol=0)
    assert not torch.allclose(a_eager, b_eager)
    # no regions: the layers keep no source cache, so step 0 of a cached layout stays eager (and the result is the eager one)
    plain = [('two people', [])]
    kw = dict(negative_prompt=[''], height=H, width=W, num_inference_steps=5, guidance_scale=7.5, output_type='latent')
    p1 = pipe(prompt=plain, latents=latents.clone(), hipgraph=True, **kw).images
    n = len(replays)
    p2 = pipe(prompt=plain, latents=latents.clone(), hipgraph=True, **kw).images
    assert len(replays) - n == 4
    torch.testing.assert_close(p1, p2, rtol=0, atol=0)


def test_edlora_pipeline_keeps_its_sampling_graph_across_calls(gpu_branches, monkeypatch):
    """EDLoRAPipeline / StableDiffusionPipeline: a later call of the same shape refills the static prompt embedding AND its
    layer-major copy in place and replays the captured UNet call for every step. The replay is emulated by a callable that
    reads the layer-major slices that existed at capture time (a hipGraph holds their addresses), whatever they contain."""
    from mixofshow.models import edlora
    from mixofshow.pipelines.pipeline_edlora import EDLoRAPipeline
    from mixofshow.utils import hipgraph as G
    pipe = EDLoRAPipeline.from_pretrained('synthetic://tiny?seed=0', torch_dtype=torch.float32)
    pipe.set_new_concept_cfg(_concept_cfg(pipe.tokenizer, pipe.text_encoder, ['<potter1>', '<potter2>']))
    latents = torch.randn((1, 4, 8, 8), generator=torch.manual_seed(7))
    replays = []

    class _FakeGraph:
        def __init__(self, fn, *example):
            self.fn = fn
            self.frozen = next(iter(pipe._sampling_graphs.values())).pe._mos_layers[1]

        def __call__(self, x, t):
            replays.append(int(t))
            real_sel, real_att = edlora._select_layer_states, edlora.attach_layer_major_states
            monkeypatch.setattr(edlora, '_select_layer_states', lambda states, idx: self.frozen[idx])
            monkeypatch.setattr(edlora, 'attach_layer_major_states', lambda states: states)
            try:
                return self.fn(x, t)
            finally:
                monkeypatch.setattr(edlora, '_select_layer_states', real_sel)
                monkeypatch.setattr(edlora, 'attach_layer_major_states', real_att)

    monkeypatch.setattr(G, 'graphs_usable', lambda device: True)
    monkeypatch.setattr(G, 'try_capture', lambda fn, *ex: _FakeGraph(fn, *ex))

    def run(text, graph, **kw):
        return pipe(prompt=text, negative_prompt='blurry', height=64, width=64, num_inference_steps=5, guidance_scale=7.5,
                    latents=latents.clone(), output_type='latent', hipgraph=graph, **kw).images

    a_graph = run('a <potter1> <potter2> in the park', True)
    assert pipe.last_call_graphed and len(replays) == 4 and pipe.last_call_replay_from == 1
    b_graph = run('a photo of a dog', True)
    assert len(replays) == 9 and pipe.last_call_replay_from == 0 and len(pipe._sampling_graphs) == 1
    b_eager = run('a photo of a dog', False)
    a_eager = run('a <potter1> <potter2> in the park', False)
    torch.testing.assert_close(b_graph, b_eager, rtol=0, atol=0)
    torch.testing.assert_close(a_graph, a_eager, rtol=0, atol=0)
    assert not torch.allclose(a_eager, b_eager)
    # a changed weight is a new model epoch: the entry is dropped and the call captures again (step 0 eager)
    with torch.no_grad():
        next(pipe.unet.parameters()).add_(0.0)
    run('a photo of a dog', True)
    assert pipe.last_call_replay_from == 1 and len(pipe._sampling_graphs) == 1
    # cross_attention_kwargs: not cached (nothing is known about what they reference), still graphed within the call
    run('a photo of a dog', True, cross_attention_kwargs={})
    assert pipe.last_call_graphed and pipe.last_call_replay_from == 1


def test_sampling_graph_cache_over_mixed_layouts_and_eager_calls(emulated_hip, monkeypatch):
    """The regional pipeline's graph cache under a mixed call sequence: four layouts (two region lists of one shape, no
    regions, one region), eager calls in between, eviction at two resident graphs. Every call must give EXACTLY what a fresh
    pipeline computes eagerly for its prompts (the emulated replay reads the K/V buffers that were current at its capture)."""
    from mixofshow.pipelines import pipeline_regionally_t2iadapter as P
    from mixofshow.utils import hipgraph as G
    H, W = 64, 96

    def make():
        p = P.RegionallyT2IAdapterPipeline.from_pretrained('synthetic://tiny?seed=0', torch_dtype=torch.float32)
        p.set_new_concept_cfg(_concept_cfg(p.tokenizer, p.text_encoder, ['<potter1>', '<potter2>']))
        return p

    pipe, ref = make(), make()
    latents = torch.randn((1, 4, H // 8, W // 8), generator=torch.manual_seed(5))

    class _FakeGraph:
        def __init__(self, fn, *example):
            self.fn = fn
            self.frozen = {id(m.processor): m.processor._kv for m in pipe.unet.modules()
                           if isinstance(getattr(m, 'processor', None), P.RegionT2I_AttnProcessor) and m.processor._kv is not None}

        def __call__(self, x, t):
            real = P.RegionT2I_AttnProcessor._source_kv
            if self.frozen:
                monkeypatch.setattr(P.RegionT2I_AttnProcessor, '_source_kv', lambda s, *a: self.frozen[id(s)])
            try:
                return self.fn(x, t)
            finally:
                monkeypatch.setattr(P.RegionT2I_AttnProcessor, '_source_kv', real)

    monkeypatch.setattr(G, 'graphs_usable', lambda device: True)
    monkeypatch.setattr(G, 'try_capture', lambda fn, *ex: _FakeGraph(fn, *ex))
    prompts = {
        'A': [('two people', [('a <potter1> <potter2>', '', [0.0, 0.0, 1.0, 0.6]), ('a dog', 'blurry', [0.2, 0.5, 0.9, 1.0])])],
        'B': [('a street', [('a red car', '', [0.0, 0.0, 1.0, 0.6]), ('a cat', 'blurry', [0.2, 0.5, 0.9, 1.0])])],
        'C': [('two people', [])],
        'D': [('a forest', [('a <potter1> <potter2>, hat', '', [0.1, 0.1, 0.8, 0.5])])],
    }

    def run(p, key, graph):
        return p(prompt=prompts[key], negative_prompt=[''], height=H, width=W, num_inference_steps=5, guidance_scale=7.5,
                 latents=latents.clone(), output_type='latent', hipgraph=graph).images

    want = {k: run(ref, k, False) for k in prompts}
    seen = []
    for key, graph in zip('ABCADCBDABCD', (1, 1, 1, 0, 1, 1, 1, 1, 1, 0, 1, 1)):
        got = run(pipe, key, bool(graph))
        torch.testing.assert_close(got, want[key], rtol=0, atol=0, msg=f'{key} graph={graph}')
        seen.append((key, pipe.last_call_replay_from if graph else None))
        assert len(pipe._sampling_graphs) <= 2
    assert ('B', 0) in seen and ('D', 0) in seen and ('A', 0) in seen          # cached layouts replay every step ...
    assert all(r == 1 for k, r in seen if k == 'C' and r is not None)            # ... the region-free one keeps its eager step 0
