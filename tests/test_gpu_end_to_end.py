"""GPU end-to-end parity: the product pipelines (HIP attention path) against the SAME modules running the oracle's
processors (plain torch restatement of the reference: baddbmm / softmax / bmm, per-region einsum + mask scatter,
3-GEMM LoRA) on identical weights, seeds and CPU-generated latents. Non-attention operators are shared, as
BASELINE.json's north_star prescribes, so the difference isolates the hot path.

Tolerance (BASELINE.json north_star): 1e-3 on denoised latents. All sampling checks are TEACHER-FORCED on the BASELINE
configs (`synthetic://sd15`, all four UNet levels, d = 40 / 80 / 160; EDLoRA 512x512 and 3(+1)-region 512x768, 50
DPM-Solver++ steps, CFG 7.5): one path runs the loop and records the latent it fed to the UNet at every step, every
other path is given the SAME latent at every step, and per step we compare (a) the UNet's raw epsilon (both CFG
halves) and (b) the latent after the scheduler update, as max|d| / max(1, |.|max).

  * `*_hot_path_error_*` (fp32 pipeline): non-attention operators run in fp32 on both sides, so the ONLY half-precision
    arithmetic is inside the attention layers = the hot path. HIP path (fp16 kernels) vs the oracle in exact fp32:
    epsilon asserted at 1e-3, every step (measured 2e-4 .. 3e-4). The post-scheduler latent is epsilon pushed through
    a linear update with a CFG gain of up to 14: the reference's OWN fp16 attention arithmetic sits at 2.1e-3 .. 2.7e-3
    from exact there, so the latent is asserted at "no worse than the reference arithmetic (+25 %)" and <= 1.4e-2.
  * `*_fp16_pipeline_*` (the benchmarked dtype): here EVERY operator rounds to half, and two valid fp16 evaluations of
    the same UNet differ by ~3 ulp of the top binade in epsilon (measured: the reference's own fp16 path sits 2.4e-3 *
    |eps|max from the exact-attention result, and two runs of the SAME eager loop differ because MIOpen's split-K
    convolutions use atomics); CFG 7.5 multiplies that by up to 14 in the scheduler update. A 1e-3 max-abs bound
    is therefore not a property any fp16 path has, the reference's included. Asserted instead, per step: the HIP path
    is no further from the exact-attention result than the reference's fp16 path is (max and RMS), with the absolute
    numbers printed.
The free-running 50-step difference is printed as a report and loosely bounded. The synthetic weights are calibrated
(mixofshow.utils.pretrained.calibrate_synthetic_unet) so that latents stay O(1) over the 50 steps; the tests print how
strongly epsilon depends on the attention path (removing attention / scaling the logits by 5 %), i.e. what the bounds
can detect.
"""
import copy
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = 'cuda'
TOL = 1e-3          # BASELINE.json north_star: tolerance on denoised latents


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


def test_graft_smoke():
    import __graft_entry__ as g
    g.smoke()


class _CastLayer:
    """The oracle processor evaluated in `dtype` on a `dtype` copy of the layer (same weights and inputs), output cast
    back. dtype=float32 in an fp16 pipeline = the exact attention layer; dtype=float16 in an fp32 pipeline = the
    reference's fp16 attention arithmetic."""

    def __init__(self, inner, dtype):
        self.inner, self.dtype = inner, dtype
        self.copy = None

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, **kw):
        if self.copy is None:
            proc, attn.processor = attn.processor, None
            self.copy = copy.deepcopy(attn).to(self.dtype)
            attn.processor = proc
        kw = dict(kw)
        if 'region_list' in kw:
            kw['region_list'] = [(r[0].to(self.dtype), r[1]) for r in kw['region_list']]
        ehs = encoder_hidden_states.to(self.dtype) if encoder_hidden_states is not None else None
        return self.inner(self.copy, hidden_states.to(self.dtype), encoder_hidden_states=ehs, **kw).to(hidden_states.dtype)


def _install_cast(unet, dtype):
    for m in unet.modules():
        if m.__class__.__name__ == 'Attention':
            inner = m.processor.inner if isinstance(m.processor, _CastLayer) else m.processor
            m.set_processor(_CastLayer(inner, dtype))


@torch.no_grad()
def _denoise_loop(pipe, prompt_embeds, latents0, cak=None, forced=None, steps=50, guidance_scale=7.5, residuals=None,
                  max_steps=None):
    """The sampling loop of the reference pipelines (pipeline_edlora.py:271-301 / pipeline_regionally_t2iadapter.py:
    548-580) written out so that every step's (input latent, raw UNet epsilon, post-scheduler latent) can be
    recorded and the input latent can be teacher-forced. Returns a list of (x_in, eps_raw, x_out)."""
    sched = pipe.scheduler
    sched.set_timesteps(steps, device=latents0.device)
    lat = latents0.to(prompt_embeds.dtype) * sched.init_noise_sigma
    rec = []
    for i, t in enumerate(sched.timesteps):
        if max_steps is not None and i >= max_steps:       # only the first steps of the `steps`-step schedule
            break
        if forced is not None:
            lat = forced[i]
        x = sched.scale_model_input(torch.cat([lat] * 2), t)
        extra = {} if residuals is None else {'down_block_additional_residuals': [r.clone() for r in residuals]}   # popped per call (reference :565)
        eps = pipe.unet(x, t, encoder_hidden_states=prompt_embeds, cross_attention_kwargs=cak, **extra).sample
        u, c = eps.chunk(2)
        new = sched.step(u + guidance_scale * (c - u), t, lat).prev_sample
        rec.append((lat, eps, new))
        lat = new
    return rec


def _absmax(t):
    return t.float().abs().max().item()


def _record_parity(name, **figures):
    """Merge the measured parity figures of one test into gpurun_out/parity_latents.json (committed afterwards as
    profiles/parity_latents.json, which bench.py embeds in its JSON line next to `roofline`: VERDICT r03 1(d))."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, 'parity_latents.json')
        data = {}
        if os.path.exists(path):
            with open(path) as f:
                data = json.load(f)
        import bench
        fp = bench.kernel_source_fingerprint(None)
        if data.get('kernel_source_sha16_all') != fp:
            data = {'kernel_source_sha16_all': fp, 'tolerance_north_star': TOL, 'normalisation':
                    'max|d| / max(1, |.|max) for *_maxabs, rms(d) / max(1, rms) for *_rms; abs_* / frac_* keys are NOT normalised '
                    '(absolute max |d|, 99.9th percentile of |d|, fraction of elements with |d| > 1e-3); worst of 50 '
                    'DPM-Solver++ steps, CFG 7.5, synthetic://sd15', 'written_by': 'tests/test_gpu_end_to_end.py', 'cases': {}}
        data['cases'][name] = {k: float(f'{v:.4e}') for k, v in figures.items()}
        with open(path, 'w') as f:
            json.dump(data, f, indent=1, sort_keys=True)
    except OSError:
        pass


def _per_step(rec_a, rec_b):
    """Worst per-step (max|d eps| / max(1,|eps|), max|d x| / max(1,|x|), rms d eps / rms eps) of a against b."""
    we = wx = wr = 0.0
    for (xa, ea, na), (xb, eb, nb) in zip(rec_a, rec_b):
        assert torch.equal(xa, xb), 'teacher forcing broken: the two paths saw different input latents'
        d = ea.float() - eb.float()
        we = max(we, _absmax(d) / max(1.0, _absmax(eb)))
        wx = max(wx, _absmax(na.float() - nb.float()) / max(1.0, _absmax(nb)))
        wr = max(wr, (d.pow(2).mean().sqrt() / eb.float().pow(2).mean().sqrt()).item())
    return we, wx, wr


def _abs_figures(rec_a, rec_b, tag):
    """UN-normalised figures of a against b: absolute max |d|, the 99.9th percentile of |d| and the
    fraction of elements with |d| > 1e-3, for epsilon and for the post-scheduler latent -- each the worst of the steps."""
    out = {}
    for what, idx in (('eps', 1), ('latent', 2)):
        mx = p999 = frac = 0.0
        for ra, rb in zip(rec_a, rec_b):
            d = (ra[idx].float() - rb[idx].float()).abs().flatten()
            mx = max(mx, d.max().item())
            k = max(1, int(round(d.numel() * 0.999)))
            p999 = max(p999, d.kthvalue(k).values.item())
            frac = max(frac, (d > TOL).float().mean().item())
        out[f'abs_{what}_max_{tag}'] = mx
        out[f'abs_{what}_p999_{tag}'] = p999
        out[f'frac_{what}_gt_1e-3_{tag}'] = frac
    return out


def _ranges(rec):
    return max(_absmax(r[1]) for r in rec), max(_absmax(r[2]) for r in rec)


def _latent_rms(rec_a, rec_b):
    """RMS of the post-scheduler latent difference, normalised by max(1, rms of the latent): (worst step, final step)."""
    worst = last = 0.0
    for (_, _, na), (_, _, nb) in zip(rec_a, rec_b):
        d = (na.float() - nb.float()).pow(2).mean().sqrt().item()
        last = d / max(1.0, nb.float().pow(2).mean().sqrt().item())
        worst = max(worst, last)
    return worst, last


@torch.no_grad()
def _sensitivity(name, pipe, prompt_embeds, x, cak=None):
    """What the bounds can see: how much epsilon moves when the attention path is removed / its logits scaled 5 %."""
    t = torch.tensor(500, device=x.device)
    xin = torch.cat([x] * 2)
    base = pipe.unet(xin, t, encoder_hidden_states=prompt_embeds, cross_attention_kwargs=cak).sample.float()
    attns = [m for m in pipe.unet.modules() if m.__class__.__name__ == 'Attention']
    for m in attns:
        m.scale *= 1.05
    scaled = pipe.unet(xin, t, encoder_hidden_states=prompt_embeds, cross_attention_kwargs=cak).sample.float()
    for m in attns:
        m.scale /= 1.05
    saved = [m.to_out[0].weight.clone() for m in attns]
    for m in attns:
        m.to_out[0].weight.zero_()
    removed = pipe.unet(xin, t, encoder_hidden_states=prompt_embeds, cross_attention_kwargs=cak).sample.float()
    for m, w in zip(attns, saved):
        m.to_out[0].weight.copy_(w)
    a, b = _absmax(scaled - base), _absmax(removed - base)
    print(f'[parity] {name}: sensitivity of epsilon to the attention path: logits x1.05 -> {a:.3e}, attention removed -> '
          f'{b:.3e} (|eps|max {_absmax(base):.2f})')
    assert b >= 50 * TOL, 'fixture too insensitive: the attention path hardly reaches epsilon'
    return a, b


def _edlora_setup(dtype):
    from mixofshow.pipelines.pipeline_edlora import EDLoRAPipeline
    pipe = EDLoRAPipeline.from_pretrained('synthetic://sd15?seed=0', torch_dtype=dtype).to(DEV)
    cfg = _concept_cfg(pipe.tokenizer, pipe.text_encoder, ['<potter1>', '<potter2>'])
    pipe.set_new_concept_cfg(cfg)
    emb = pipe._encode_prompt('a <potter1> <potter2> in the park', cfg, DEV, 1, True, None)
    latents = torch.randn((1, 4, 64, 64), generator=torch.manual_seed(1)).to(DEV)     # PromptDataset recipe, index 1
    return pipe, emb, None, latents


def _regional_setup(preset, dtype=torch.float16, H=512, W=768):
    from bench import regional_prompt
    from mixofshow.pipelines.pipeline_regionally_t2iadapter import RegionallyT2IAdapterPipeline
    pipe = RegionallyT2IAdapterPipeline.from_pretrained(f'synthetic://{preset}?seed=0', torch_dtype=dtype).to(DEV)
    cfg = _concept_cfg(pipe.tokenizer, pipe.text_encoder,
                       ['<potter1>', '<potter2>', '<hermione1>', '<hermione2>', '<thanos1>', '<thanos2>'])
    pipe.set_new_concept_cfg(cfg)
    prompt, neg = regional_prompt(H, W)
    if (H, W) == (512, 768):        # (the shipped 1024x2048 example runs as shipped: three regions)
        prompt[0][1].append(('a castle', neg, [100 / H, 150 / W, 400 / H, 300 / W]))   # overlaps regions 1 and 2
    emb, region_list = pipe._encode_region_prompt(prompt, cfg, DEV, 1, True, [neg], height=H, width=W)
    cak = {'region_list': region_list, 'height': H, 'width': W}
    latents = torch.randn((1, 4, H // 8, W // 8), generator=torch.manual_seed(14)).to(DEV)
    return pipe, emb, cak, latents


def _install_oracle(pipe, regional):
    from oracle import edlora_ref as R
    from oracle import region_ref
    if regional:
        region_ref.install_region_processors_ref(pipe.unet)
    else:
        for m in pipe.unet.modules():
            if m.__class__.__name__ == 'Attention':
                m.set_processor(R.PlainAttnProcessorRef())
        R.install_ref_processors(pipe.unet)


def _restore_hip(pipe, hip_procs):
    for n, m in pipe.unet.named_modules():
        if n in hip_procs:
            m.set_processor(hip_procs[n])
            if hasattr(m.processor, 'reset_cache'):
                m.processor.reset_cache()


@torch.no_grad()
def _peak_attention_logits(pipe, emb, latents, cak, target):
    """Trained SD-1.5 attention has PEAKED softmaxes (max logits of 20..40: the online-softmax rescale branch, probabilities
    of ~1 next to ~e^-30); the calibrated random weights give maxima of ~10. One UNet call records every attention layer's
    input, then each layer's `to_q` is scaled so that ITS largest |logit| on that input equals `target`. Returns
    (largest before, largest after) over the layers."""
    attns = [m for m in pipe.unet.modules() if m.__class__.__name__ == 'Attention']
    seen = {}

    def hook(mod, args, kwargs):
        x = args[0] if args else kwargs['hidden_states']
        e = kwargs.get('encoder_hidden_states')
        if e is None and len(args) > 1:
            e = args[1]
        seen[mod] = (x.detach(), None if e is None else e.detach())

    hs = [m.register_forward_pre_hook(hook, with_kwargs=True) for m in attns]
    t = torch.tensor(500, device=latents.device)
    pipe.unet(torch.cat([latents.to(emb.dtype)] * 2), t, encoder_hidden_states=emb, cross_attention_kwargs=cak)
    for h in hs:
        h.remove()

    def max_logit(m):
        x, e = seen[m]
        if e is not None and e.dim() == 4:                       # layer-indexed text states (edlora.py:56-57)
            e = e[:, getattr(m.processor, 'cross_attention_idx', 0)]
        src = x if e is None else e
        q = m.to_q(x.float()).unflatten(-1, (m.heads, -1)).transpose(1, 2)
        k = m.to_k(src.float()).unflatten(-1, (m.heads, -1)).transpose(1, 2)
        return max((q[:, h0:h0 + 2] @ k[:, h0:h0 + 2].transpose(-1, -2)).abs().max().item() for h0 in range(0, m.heads, 2)) * m.scale

    before = after = 0.0
    for m in attns:
        mx = max_logit(m)
        before = max(before, mx)
        m.to_q.weight.mul_(target / mx)
        after = max(after, max_logit(m))
    return before, after


def _hot_path_error(name, setup, regional, residuals_fn=None, steps=50, peak_logits=None, max_steps=None):
    """fp32 pipeline: the only half-precision arithmetic is the attention layers. HIP vs exact at 1e-3, every step.
    residuals_fn(pipe) -> adapter states fed as `down_block_additional_residuals` to every UNet call (both paths).
    peak_logits: the "trained-model" fixture -- every attention layer's scores scaled to that maximum.
    With |logit| ~ 30 the fp16 roundings of q and k alone move a score by ~1e-2, so the reference's OWN fp16 arithmetic is
    percent-level away from exact there; the assertion is "HIP no further from exact than the reference arithmetic"."""
    pipe, emb, cak, latents = setup(torch.float32)
    res = residuals_fn(pipe) if residuals_fn is not None else None
    hip_procs = {n: m.processor for n, m in pipe.unet.named_modules() if m.__class__.__name__ == 'Attention'}
    if peak_logits is not None:
        lb, la = _peak_attention_logits(pipe, emb, latents, cak, peak_logits)
        print(f'[parity] {name}: attention logits peaked: largest |logit| {lb:.1f} -> {la:.1f} (every layer at {peak_logits})')
        assert 0.8 * peak_logits <= la <= 1.2 * peak_logits
    rec_free = _denoise_loop(pipe, emb, latents, cak=cak, residuals=res, steps=steps, max_steps=max_steps)
    _sensitivity(name, pipe, emb, latents, cak=cak)
    _install_oracle(pipe, regional)
    rec_exact = _denoise_loop(pipe, emb, latents, cak=cak, residuals=res, steps=steps, max_steps=max_steps)       # oracle, exact fp32 attention
    forced = [r[0] for r in rec_exact]
    _install_cast(pipe.unet, torch.float16)
    rec_ref16 = _denoise_loop(pipe, emb, latents, cak=cak, forced=forced, residuals=res, steps=steps, max_steps=max_steps)   # reference fp16 attention arithmetic
    _restore_hip(pipe, hip_procs)
    rec_hip = _denoise_loop(pipe, emb, latents, cak=cak, forced=forced, residuals=res, steps=steps, max_steps=max_steps)
    he, hx, hr = _per_step(rec_hip, rec_exact)
    re_, rx, rr = _per_step(rec_ref16, rec_exact)
    de, dx, dr = _per_step(rec_hip, rec_ref16)                                 # HIP against the reference arithmetic, directly
    emax, xmax = _ranges(rec_exact)
    free = _absmax(rec_free[-1][2] - rec_exact[-1][2]) / max(1.0, _absmax(rec_exact[-1][2]))
    (hw, hl), (dw, dl), (rw, rl) = _latent_rms(rec_hip, rec_exact), _latent_rms(rec_hip, rec_ref16), _latent_rms(rec_ref16, rec_exact)
    fw, fl = _latent_rms(rec_free, rec_exact)
    print(f'[parity] {name}, fp32 pipeline, teacher-forced 50 steps, worst step: HIP vs exact: |d eps|/max(1,|eps|) = '
          f'{he:.3e}, |d x|/max(1,|x|) = {hx:.3e}, rms rel eps {hr:.3e}; reference fp16 attention vs exact: {re_:.3e}, '
          f'{rx:.3e}, {rr:.3e}; HIP vs reference fp16 attention DIRECTLY: {de:.3e}, {dx:.3e}, {dr:.3e}; |eps|max {emax:.2f} '
          f'|x|max {xmax:.2f}; free-running 50-step HIP vs exact {free:.3e}')
    print(f'[parity] {name}, denoised-latent RMS error / max(1, rms latent) (worst step, final step): HIP vs exact {hw:.3e} '
          f'{hl:.3e}; HIP vs reference fp16 attention {dw:.3e} {dl:.3e}; reference fp16 attention vs exact {rw:.3e} {rl:.3e}; '
          f'FREE-RUNNING 50 steps HIP vs exact {fw:.3e} {fl:.3e}')
    _record_parity(name + ' | fp32 pipeline (attention layers only in half)',
                   eps_maxabs_hip_vs_exact=he, eps_maxabs_hip_vs_ref_fp16=de, eps_maxabs_ref_fp16_vs_exact=re_,
                   latent_maxabs_teacher_forced_hip_vs_exact=hx, latent_maxabs_teacher_forced_hip_vs_ref_fp16=dx,
                   latent_maxabs_teacher_forced_ref_fp16_vs_exact=rx, latent_maxabs_free_running_final_hip_vs_exact=free,
                   latent_rms_teacher_forced_hip_vs_exact=hw, latent_rms_teacher_forced_hip_vs_ref_fp16=dw,
                   latent_rms_teacher_forced_ref_fp16_vs_exact=rw, latent_rms_free_running_final_hip_vs_exact=fl,
                   eps_absmax=emax, latent_absmax=xmax,
                   **_abs_figures(rec_hip, rec_exact, 'hip_vs_exact'), **_abs_figures(rec_hip, rec_ref16, 'hip_vs_ref_fp16'),
                   **_abs_figures(rec_ref16, rec_exact, 'ref_fp16_vs_exact'),
                   **_abs_figures(rec_free[-1:], rec_exact[-1:], 'free_running_final_hip_vs_exact'))
    assert xmax <= 8.0, f'{name}: calibrated synthetic latents should stay O(1), got {xmax}'
    if peak_logits is not None:
        # yardstick = the reference's own fp16 attention arithmetic on the same peaked scores (see the docstring); the peaked
        # sampler is less contractive, so the FREE-running error is read against the reference arithmetic run freely too
        # (first run of round 5: HIP 1.74e-3 after 12 free steps against 4.7e-4 per teacher-forced step)
        _install_oracle(pipe, regional)
        _install_cast(pipe.unet, torch.float16)
        rec_free16 = _denoise_loop(pipe, emb, latents, cak=cak, residuals=res, steps=steps, max_steps=max_steps)
        _restore_hip(pipe, hip_procs)
        _, fl16 = _latent_rms(rec_free16, rec_exact)
        print(f'[parity] {name}: free-running final-latent RMS error, HIP {fl:.3e} / reference fp16 arithmetic {fl16:.3e}')
        _record_parity(name + ' | free-running yardstick', latent_rms_free_running_final_hip_vs_exact=fl,
                       latent_rms_free_running_final_ref_fp16_vs_exact=fl16)
        assert he <= 1.25 * re_ + 1e-4, f'{name}: epsilon {he:.3e} vs reference fp16 arithmetic {re_:.3e} (both against exact)'
        # the relative criteria alone would let a kernel-side accuracy regression of up to 25 % pass wherever the
        # reference arithmetic is worse; absolute ceilings on what the hot path itself produces: teacher-forced epsilon within
        # north_star's 1e-3 (2.8e-4 / 1.8e-4 measured), teacher-forced latent RMS within 1e-3 (4.7e-4 / 4.4e-4 measured)
        assert he <= TOL, f'{name}: raw epsilon differs by {he:.3e} from exact attention (teacher-forced, peaked logits)'
        assert hw <= TOL, f'{name}: teacher-forced latent RMS error {hw:.3e} (peaked logits)'
        assert hw <= 1.25 * rw + 1e-4 and hx <= 1.25 * rx + 1e-4, f'{name}: latent {hw:.3e} / {hx:.3e} vs {rw:.3e} / {rx:.3e}'
        assert fl <= max(TOL, 1.5 * fl16 + 1e-4), f'{name}: free-running final-latent RMS error {fl:.3e} (reference arithmetic {fl16:.3e})'
        return
    # epsilon = what the hot path produces: north_star's 1e-3, every step -- against exact attention AND against the
    # reference's own fp16 attention arithmetic
    assert he <= TOL, f'{name}: raw epsilon differs by {he:.3e} (teacher-forced)'
    assert de <= TOL, f'{name}: raw epsilon differs from the reference fp16 arithmetic by {de:.3e}'
    # denoised latents, north_star's 1e-3 as an RMS figure: every teacher-forced step and the free-running final latent
    assert hw <= TOL and dw <= TOL, f'{name}: latent RMS error {hw:.3e} (vs exact) / {dw:.3e} (vs reference fp16 arithmetic)'
    assert fl <= TOL, f'{name}: free-running final-latent RMS error {fl:.3e}'
    # the scheduler update is linear in epsilon with a CFG gain of up to 2*7.5-1 = 14 on a per-half error: the latent
    # inherits that amplified error on BOTH sides — the reference's own fp16 attention arithmetic measures 2.1e-3 ..
    # 2.7e-3 here (HIP: 0.87 .. 0.94 of that over six runs). Bound: no worse than the reference arithmetic (+25 %: both are
    # maxima over 1.6 M elements x 50 steps), and within 1e-3 * CFG gain absolutely.
    assert hx <= max(TOL, 1.25 * rx), f'{name}: post-scheduler latent {hx:.3e} vs reference fp16 arithmetic {rx:.3e}'
    assert hx <= TOL * (2 * 7.5 - 1)
    assert free <= 1e-2, f'{name}: free-running latents differ by {free:.3e}'


def _fp16_pipeline_band(name, setup, regional):
    """fp16 pipeline (the benchmarked dtype): HIP no further from exact attention than the reference's fp16 path.
    The shared (non-attention) 3x3 convolutions ALL run on the library's deterministic implicit-GEMM kernel for this test
    (product default: MIOpen below 4096 output pixels, whose split-K kernels accumulate with atomics), so that "reference
    vs itself" is the noise floor the other differences are read against. (MIOpen's own deterministic mode was tried:
    > 4 minutes per 50-step loop on this box.)"""
    from mixofshow.hip import functional as F_hip
    saved = F_hip._conv_min_pixels
    F_hip._conv_min_pixels = 0
    try:
        _fp16_pipeline_band_body(name, setup, regional)
    finally:
        F_hip._conv_min_pixels = saved


def _fp16_pipeline_band_body(name, setup, regional):
    pipe, emb, cak, latents = setup(torch.float16)
    hip_procs = {n: m.processor for n, m in pipe.unet.named_modules() if m.__class__.__name__ == 'Attention'}
    rec_free = _denoise_loop(pipe, emb, latents, cak=cak)
    _install_oracle(pipe, regional)
    # one discarded call first: the FIRST evaluation of a convolution shape goes through MIOpen's find step, which may
    # return the result of another algorithm than the one used afterwards (that, not atomics, is what made "two runs of
    # the same loop" differ by ~3 ulp)
    _denoise_loop(pipe, emb, latents, cak=cak, steps=2)
    rec_ref = _denoise_loop(pipe, emb, latents, cak=cak)                        # the reference path, fp16
    forced = [r[0] for r in rec_ref]
    rec_ref2 = _denoise_loop(pipe, emb, latents, cak=cak, forced=forced)       # same path again: run-to-run noise
    _install_cast(pipe.unet, torch.float32)
    rec_exact = _denoise_loop(pipe, emb, latents, cak=cak, forced=forced)      # exact attention in the fp16 UNet
    _restore_hip(pipe, hip_procs)
    rec_hip = _denoise_loop(pipe, emb, latents, cak=cak, forced=forced)
    he, hx, hr = _per_step(rec_hip, rec_exact)
    re_, rx, rr = _per_step(rec_ref, rec_exact)
    pe, px, pr = _per_step(rec_hip, rec_ref)
    ne, nx, nr = _per_step(rec_ref2, rec_ref)
    emax, xmax = _ranges(rec_ref)
    free = _absmax(rec_free[-1][2].float() - rec_ref[-1][2].float()) / max(1.0, _absmax(rec_ref[-1][2]))
    print(f'[parity] {name}, fp16 pipeline, teacher-forced 50 steps, worst step (max eps, max x, rms-rel eps): HIP vs exact '
          f'{he:.3e} {hx:.3e} {hr:.3e}; reference fp16 path vs exact {re_:.3e} {rx:.3e} {rr:.3e}; HIP vs reference '
          f'{pe:.3e} {px:.3e} {pr:.3e}; reference vs itself (2 runs) {ne:.3e} {nx:.3e} {nr:.3e}; |eps|max {emax:.2f} '
          f'|x|max {xmax:.2f}; free-running 50-step HIP vs reference {free:.3e}')
    _record_parity(name + ' | fp16 pipeline (every operator in half: the benchmarked dtype)',
                   eps_maxabs_hip_vs_exact=he, eps_maxabs_ref_path_vs_exact=re_, eps_maxabs_hip_vs_ref_path=pe,
                   eps_maxabs_ref_path_vs_itself=ne, latent_maxabs_teacher_forced_hip_vs_exact=hx,
                   latent_maxabs_teacher_forced_ref_path_vs_exact=rx, latent_maxabs_teacher_forced_hip_vs_ref_path=px,
                   latent_maxabs_teacher_forced_ref_path_vs_itself=nx, latent_maxabs_free_running_final_hip_vs_ref_path=free,
                   eps_rmsrel_hip_vs_exact=hr, eps_rmsrel_ref_path_vs_exact=rr, eps_rmsrel_hip_vs_ref_path=pr,
                   eps_absmax=emax, latent_absmax=xmax,
                   **_abs_figures(rec_hip, rec_exact, 'hip_vs_exact'), **_abs_figures(rec_hip, rec_ref, 'hip_vs_ref_path'),
                   **_abs_figures(rec_ref, rec_exact, 'ref_path_vs_exact'), **_abs_figures(rec_ref2, rec_ref, 'ref_path_vs_itself'))
    assert xmax <= 8.0
    ulp = 2.0 ** -8 / max(1.0, emax)         # one fp16 ulp of the top binade, in the normalised units above
    assert hr <= 1.15 * rr + 1e-5, f'{name}: HIP rms error {hr:.3e} vs reference path {rr:.3e} (both against exact)'
    assert he <= 1.5 * re_ + ulp and hx <= 1.5 * rx + 14 * ulp, f'{name}: HIP max error outside the reference band'
    assert he <= 5e-3 and free <= 1e-1
    # HIP against the reference's fp16 path DIRECTLY. The reference path is not bit-reproducible here (hipBLASLt's stream-K
    # feed-forward GEMMs accumulate atomically: "reference vs itself" measures 2.0e-3 .. 2.6e-3), so the yardstick is the
    # larger of that run-to-run spread and the reference path's own distance from exact attention -- asserted always (the
    # round-3 form of this check only fired for a bit-reproducible reference, i.e. never).
    assert pe <= 1.25 * max(ne, re_) + ulp, f'{name}: HIP vs reference path {pe:.3e} (reference vs itself {ne:.3e}, vs exact {re_:.3e})'
    assert pr <= 1.25 * max(nr, rr) + 1e-5, f'{name}: HIP vs reference path rms {pr:.3e} (reference vs itself {nr:.3e}, vs exact {rr:.3e})'
    assert px <= 1.25 * max(nx, rx) + 14 * ulp, f'{name}: latent HIP vs reference path {px:.3e} (itself {nx:.3e}, vs exact {rx:.3e})'


def test_edlora_sd15_hot_path_error_teacher_forced():
    """SD-1.5 architecture, 512x512, 50 DPM-Solver++ steps, CFG 7.5, ED-LoRA layer-wise prompts
    (reference pipeline_edlora.py:271-301): 1e-3 on epsilon and on the denoised latent, every step."""
    _hot_path_error('edlora sd15 512x512', _edlora_setup, False)


def test_regional_sd15_hot_path_error_teacher_forced():
    """BASELINE configs[4]: 3 regions (+1 overlapping) at 512x768, 50 steps, CFG pair per call
    (reference pipeline_regionally_t2iadapter.py:548-580): 1e-3 on epsilon and on the denoised latent, every step."""
    _hot_path_error('regional sd15 512x768', lambda dt: _regional_setup('sd15', dt), True)


def test_regional_sd15_shipped_example_1024x2048_teacher_forced():
    """VERDICT r04 missing #3: the reference's own shipped regional example (regionally_sample.sh:52-90: 1024 x 2048, the three
    boxes unscaled, seed 14) -- latent 128 x 256, N = 32768 queries AND self-attention keys at level 0, 8192 / 2048 / 512 below:
    grid limits, 32-bit offsets and the region kernel's box table at 5.3x the largest size of the other tests. First 5 steps of
    the 50-step schedule, teacher-forced, fp32 pipeline (attention layers only in half), HIP vs exact attention (the oracle's
    probability tensors are built in (batch x head) slices: 69 GB otherwise) and vs the reference's fp16 arithmetic."""
    _hot_path_error('regional sd15 1024x2048 (regionally_sample.sh)', lambda dt: _regional_setup('sd15', dt, 1024, 2048), True,
                    max_steps=5)


def test_regional_sd15_with_adapter_states_hot_path_error_teacher_forced():
    """SURVEY 8(f).2 on the device (reference pipeline_regionally_t2iadapter.py:474-546, :565): the reference cannot sample
    without an adapter input. Seeded synthetic 4-level adapter features go through the product's region-weight rule ON
    THE GPU (== the same rule on the CPU, which tests/test_pipelines_cpu.py pins to the reference's own source lines) and
    enter every UNet call as `down_block_additional_residuals` (CFG pair), HIP processors vs the oracle's."""
    from bench import synthetic_adapter_states

    def residuals(pipe):
        gpu = synthetic_adapter_states(pipe, 512, 768, DEV, torch.float32)
        # the same rule on CPU tensors: identical weights map and features
        import math
        from bench import REGION_PX
        g = torch.Generator().manual_seed(15)
        chans = pipe.unet.config.block_out_channels
        b = REGION_PX[0]
        for i, (c, f_gpu) in enumerate(zip(chans, gpu)):
            f = torch.randn((1, c, 512 // (8 << i), 768 // (8 << i)), generator=g).mul_(0.05)
            fh, fw = f.shape[2:]
            w = torch.ones(fh, fw)
            w[math.ceil(b[0] / 512 * fh):math.floor(b[2] / 512 * fh), math.ceil(b[1] / 768 * fw):math.floor(b[3] / 768 * fw)] = 0.6
            torch.testing.assert_close(f_gpu.cpu(), w * f, rtol=0, atol=0)
        assert len(gpu) == 4 and gpu[0].shape == (1, 320, 64, 96) and gpu[3].shape == (1, 1280, 8, 12)
        return [torch.cat([s_] * 2) for s_ in gpu]

    _hot_path_error('regional sd15 512x768 + adapter states', lambda dt: _regional_setup('sd15', dt), True, residuals_fn=residuals)


def test_edlora_sd15_hot_path_error_peaked_logits():
    """The trained-model regime (max logits ~30: probabilities ~1 beside ~e^-30, the rescale branch of the online softmax in
    every tile) through all 32 attention layers, 12 teacher-forced steps."""
    _hot_path_error('edlora sd15 512x512, logits peaked at 30', _edlora_setup, False, steps=12, peak_logits=30.0)


def test_regional_sd15_hot_path_error_peaked_logits():
    _hot_path_error('regional sd15 512x768, logits peaked at 30', lambda dt: _regional_setup('sd15', dt), True, steps=12,
                    peak_logits=30.0)


@torch.no_grad()
def test_reference_style_attention_store_gets_full_maps_on_the_hip_path():
    """VERDICT r04 missing #2, on the device at SD-1.5 size: a controller that follows the reference's protocol and declares no
    token positions -- `AttentionStore(training=False)` as prompt-to-prompt uses it (reference ptp_util.py:37-53,79-98) -- is handed
    the dense (B*H, N, 77) probabilities of all 16 cross-attention layers (mos_attn_probs), keeps the conditional half, and the
    layer output continues from what it returned (mos_attn_pv). Against the oracle's full-map store on exact fp32 attention:
    the 16 averaged maps after two UNet calls, and epsilon."""
    from mixofshow.models.edlora import revise_edlora_unet_attention_controller_forward
    from mixofshow.utils.ptp_util import AttentionStore
    from oracle import edlora_ref as R
    pipe, emb, _, latents = _edlora_setup(torch.float32)
    store = AttentionStore(training=False)
    revise_edlora_unet_attention_controller_forward(pipe.unet, store)
    assert store.num_att_layers == 16 and store.token_positions is None
    x = torch.cat([latents] * 2)
    ts = [torch.tensor(t, device=DEV) for t in (801, 401)]
    eps_hip = [pipe.unet(x, t, encoder_hidden_states=emb).sample.float() for t in ts]
    assert store.cur_step == 2
    maps_hip = store.get_average_attention()
    ref = R.AttentionStoreRef(training=False)
    for m in pipe.unet.modules():
        if m.__class__.__name__ == 'Attention':
            m.set_processor(R.PlainAttnProcessorRef())
    R.install_ref_processors(pipe.unet, controller=ref, control=True)
    ref.num_att_layers = 16
    eps_ref = [pipe.unet(x, t, encoder_hidden_states=emb).sample.float() for t in ts]
    maps_ref = ref.get_average_attention()
    worst = 0.0
    n = 0
    for k, lst in maps_ref.items():
        assert len(maps_hip[k]) == len(lst)
        for a, b in zip(maps_hip[k], lst):
            assert a.shape == b.shape and a.shape[0] == 8 and a.shape[2] == 77          # conditional half: 1 sample x 8 heads
            worst = max(worst, (a.float() - b.float()).abs().max().item())
            n += 1
    de = max(_absmax(a - b) / max(1.0, _absmax(b)) for a, b in zip(eps_hip, eps_ref))
    print(f'[parity] full-map controller on the HIP path: {n} stored maps, worst |dP| = {worst:.3e}; epsilon vs exact {de:.3e}')
    # probabilities in [0, 1] stored in half (ulp 4.9e-4 at the top) from half q / k projections, averaged in half, against the
    # exact fp32 maps: 1.3e-3 measured; epsilon (4e-5) is what continues from the returned map
    assert n == 16 and worst <= 3e-3 and de <= TOL


def test_edlora_sd15_fp16_pipeline_inside_reference_band():
    _fp16_pipeline_band('edlora sd15 512x512', _edlora_setup, False)


def test_regional_sd15_fp16_pipeline_inside_reference_band():
    _fp16_pipeline_band('regional sd15 512x768', lambda dt: _regional_setup('sd15', dt), True)


def test_regional_sampling_graph_is_reused_across_calls_and_refreshed():
    """The captured UNet graph is kept across `pipe(...)` calls of the same shape; prompt / region embeddings, adapter
    features and the processors' K/V caches are refreshed IN PLACE (step 0 of every call runs eagerly). A second call with
    OTHER prompts must reproduce the eager result of those prompts -- not the first call's."""
    from bench import regional_prompt, synthetic_adapter_states
    pipe, _, _, latents = _regional_setup('small')
    H, W = 512, 768
    p1, neg = regional_prompt(H, W)
    p2 = [('two wizards in a forest, oil painting', [(r[0].replace('castle', 'lake'), r[1], r[2]) for r in p1[0][1]])]
    ad = synthetic_adapter_states(pipe, H, W, DEV, torch.float16)

    def run(prompt, g):
        return pipe(prompt=prompt, negative_prompt=[neg], height=H, width=W, num_inference_steps=12, guidance_scale=7.5,
                    latents=latents.clone().cpu(), output_type='latent', hipgraph=g, adapter_states=ad).images.float()

    a_graph = run(p1, True)
    assert pipe.last_call_graphed and len(pipe._sampling_graphs) == 1
    b_graph = run(p2, True)                      # cache hit: no new capture, statics refreshed
    assert pipe.last_call_graphed and len(pipe._sampling_graphs) == 1
    b_eager = run(p2, False)
    a_eager = run(p1, False)
    a_again = run(p1, True)
    scale = max(1.0, _absmax(b_eager))
    d_b, d_a, d_ab = _absmax(b_graph - b_eager) / scale, _absmax(a_again - a_eager) / scale, _absmax(a_eager - b_eager) / scale
    print(f'[parity] regional graph reuse (small, 12 steps): replayed vs eager, prompts B {d_b:.3e}, prompts A again {d_a:.3e}; '
          f'prompts A vs B differ by {d_ab:.3e}')
    assert d_ab > 20 * max(d_a, d_b, 1e-4), 'fixture: the two prompt sets should give clearly different latents'
    assert d_b <= 0.05 and d_a <= 0.05 and _absmax(a_graph - a_again) / scale <= 0.05


def test_edlora_sampling_graph_is_reused_across_calls_and_refreshed():
    """EDLoRAPipeline keeps the captured UNet graph across calls of one shape (a validation loop samples many prompts): the
    static prompt embedding and its layer-major copy are refilled in place and ALL steps of a later call are replays. A second
    call with ANOTHER prompt must reproduce the eager result of that prompt, and be faster than a capturing call."""
    import time
    from mixofshow.pipelines.pipeline_edlora import EDLoRAPipeline
    pipe = EDLoRAPipeline.from_pretrained('synthetic://small?seed=0', torch_dtype=torch.float16).to(DEV)
    pipe.set_new_concept_cfg(_concept_cfg(pipe.tokenizer, pipe.text_encoder, ['<potter1>', '<potter2>']))
    latents = torch.randn((1, 4, 64, 64), generator=torch.manual_seed(1))

    def run(prompt, g):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = pipe(prompt=prompt, negative_prompt='blurry', height=512, width=512, num_inference_steps=12, guidance_scale=7.5,
                   latents=latents.clone(), output_type='latent', hipgraph=g).images.float()
        torch.cuda.synchronize()
        return out, time.perf_counter() - t0

    pa, pb = 'a <potter1> <potter2> in the park', 'a photo of a dog on the beach'
    run(pa, False)                                                     # warm the weight caches
    a_graph, t_capture = run(pa, True)
    assert pipe.last_call_graphed and pipe.last_call_replay_from == 1 and len(pipe._sampling_graphs) == 1
    b_graph, t_replay = run(pb, True)
    assert pipe.last_call_graphed and pipe.last_call_replay_from == 0 and len(pipe._sampling_graphs) == 1
    b_eager, t_eager = run(pb, False)
    a_eager, _ = run(pa, False)
    a_again, _ = run(pa, True)
    scale = max(1.0, _absmax(b_eager))
    d_b, d_a, d_ab = _absmax(b_graph - b_eager) / scale, _absmax(a_again - a_eager) / scale, _absmax(a_eager - b_eager) / scale
    print(f'[parity] edlora graph reuse (small, 12 steps): replayed vs eager, prompt B {d_b:.3e}, prompt A again {d_a:.3e}; '
          f'prompts A vs B differ by {d_ab:.3e}; call latency: capturing {t_capture * 1e3:.0f} ms, replaying {t_replay * 1e3:.0f} ms, '
          f'eager {t_eager * 1e3:.0f} ms')
    assert d_ab > 20 * max(d_a, d_b, 1e-4), 'fixture: the two prompts should give clearly different latents'
    assert d_b <= 0.05 and d_a <= 0.05 and _absmax(a_graph - a_again) / scale <= 0.05
    assert t_replay < t_capture


def test_pipeline_call_equals_written_out_loop():
    """The product's own `pipe(...)` entry points run the loop the teacher-forced tests write out: same latent after
    the first scheduler update (later steps inherit MIOpen's run-to-run noise; the final latent is a report)."""
    from mixofshow.pipelines.pipeline_edlora import EDLoRAPipeline
    pipe = EDLoRAPipeline.from_pretrained('synthetic://small?seed=0', torch_dtype=torch.float16).to(DEV)
    cfg = _concept_cfg(pipe.tokenizer, pipe.text_encoder, ['<potter1>', '<potter2>'])
    pipe.set_new_concept_cfg(cfg)
    latents = torch.randn((1, 4, 64, 64), generator=torch.manual_seed(1))
    seen = []
    out = pipe(prompt='a <potter1> <potter2> in the park', height=512, width=512, num_inference_steps=50,
               guidance_scale=7.5, latents=latents.clone(), output_type='latent',
               callback=lambda i, t, l: seen.append(l.clone())).images
    emb = pipe._encode_prompt('a <potter1> <potter2> in the park', cfg, DEV, 1, True, None)
    rec = _denoise_loop(pipe, emb, latents.to(DEV))
    d0 = _absmax(seen[0].float() - rec[0][2].float())
    d = _absmax(out.float() - rec[-1][2].float())
    print(f'[parity] EDLoRAPipeline.__call__ vs written-out loop: after step 1 max|d| = {d0:.3e}, after 50 steps {d:.3e}')
    assert len(seen) == 50 and d0 <= 4 * TOL * max(1.0, _absmax(seen[0])) and d <= 0.1 * max(1.0, _absmax(out))   # d0: 1-2 fp16 ulps
    rp, emb, cak, lat = _regional_setup('small')
    from bench import regional_prompt
    prompt, neg = regional_prompt(512, 768)
    prompt[0][1].append(('a castle', neg, [100 / 512, 150 / 768, 400 / 512, 300 / 768]))
    seen = []
    out = rp(prompt=prompt, negative_prompt=[neg], height=512, width=768, num_inference_steps=50, guidance_scale=7.5,
             latents=lat.clone(), output_type='latent', callback=lambda i, t, l: seen.append(l.clone())).images
    rec = _denoise_loop(rp, emb, lat, cak=cak)
    d0 = _absmax(seen[0].float() - rec[0][2].float())
    d = _absmax(out.float() - rec[-1][2].float())
    print(f'[parity] RegionallyT2IAdapterPipeline.__call__ vs written-out loop: after step 1 max|d| = {d0:.3e}, after 50 '
          f'steps {d:.3e}')
    assert len(seen) == 50 and d0 <= 4 * TOL * max(1.0, _absmax(seen[0])) and d <= 0.1 * max(1.0, _absmax(out))   # d0: 1-2 fp16 ulps


def test_hipgraph_regional_sampling_equals_eager_sampling():
    """50-step regional latents with the UNet call replayed from a hipGraph (the default) vs launched eagerly; a UNet with
    forward hooks falls back to the eager loop."""
    from bench import regional_prompt
    from mixofshow.pipelines.pipeline_regionally_t2iadapter import RegionallyT2IAdapterPipeline
    H, W = 512, 768
    rp = RegionallyT2IAdapterPipeline.from_pretrained('synthetic://small?seed=0', torch_dtype=torch.float16).to(DEV)
    rp.set_new_concept_cfg(_concept_cfg(rp.tokenizer, rp.text_encoder,
                                        ['<potter1>', '<potter2>', '<hermione1>', '<hermione2>', '<thanos1>', '<thanos2>']))
    lat = torch.randn((1, 4, H // 8, W // 8), generator=torch.manual_seed(14))
    prompt, neg = regional_prompt(H, W)
    rkw = dict(prompt=prompt, negative_prompt=[neg], height=H, width=W, num_inference_steps=50, guidance_scale=7.5,
               output_type='latent')
    r_eager = rp(latents=lat.clone(), hipgraph=False, **rkw).images
    assert not rp.last_call_graphed
    r_eager2 = rp(latents=lat.clone(), hipgraph=False, **rkw).images
    r_graph = rp(latents=lat.clone(), **rkw).images              # the default (hipgraph=None) replays
    assert rp.last_call_graphed, 'the default call did not replay from a hipGraph (capture fell back to eager?)'
    h = rp.unet.conv_in.register_forward_hook(lambda m, i, o: None)
    rp(latents=lat.clone(), **dict(rkw, num_inference_steps=4))
    h.remove()
    assert not rp.last_call_graphed, 'a hooked UNet must be called eagerly'
    d = (r_eager.float() - r_graph.float()).abs().max().item()
    # a replayed graph launches the same kernels on the same data; yardstick = the eager loop's own run-to-run spread
    spread = (r_eager.float() - r_eager2.float()).abs().max().item()
    scale = max(1.0, r_eager.float().abs().max().item())
    print(f'[parity] hipgraph vs eager regional sampling: max|d|={d:.3e}, eager run-to-run max|d|={spread:.3e}, '
          f'latents absmax={scale:.2f}')
    assert d <= max(2.0 * spread, TOL * scale)          # spread: MIOpen split-K convolutions (atomics), amplified by CFG


def test_training_steps_match_reference_path_and_engine_runs():
    """3 optimisation steps: loss trajectory vs the oracle path (CPU fp32 twin updated with the same AdamW)."""
    from bench import TRAIN_OPT, build_trainer, synthetic_batch
    from mixofshow.pipelines.train_loop import TrainEngine
    from oracle import trainer_ref
    tr = build_trainer('small', torch.device(DEV))
    torch.manual_seed(1)
    with torch.no_grad():
        for l in list(tr.text_encoder_lora) + list(tr.unet_lora):
            l.lora_up.weight.normal_(0, 0.02)
    twin = trainer_ref.make_reference_twin(tr, device='cpu', dtype=torch.float32)
    engine = TrainEngine(tr, dict(TRAIN_OPT, optim_g=dict(TRAIN_OPT['optim_g'])), total_iter=100, mixed_precision='bf16')
    groups = [dict(params=[twin['concept']], lr=1e-3),
              dict(params=[p for l in twin['te_lora'] for p in (l.lora_down.weight, l.lora_up.weight)], lr=1e-5),
              dict(params=[p for l in twin['unet_lora'] for p in (l.lora_down.weight, l.lora_up.weight)], lr=1e-4)]
    ref_opt = torch.optim.AdamW(groups, lr=0.0, weight_decay=0.01, betas=(0.9, 0.999))
    g = torch.Generator().manual_seed(3)
    losses, ref_losses = [], []
    for step in range(3):
        B = 2
        b = synthetic_batch(B, 256, 'cpu', 50 + step)
        extra = dict(latents=torch.randn(B, 4, 32, 32, generator=g), noise=torch.randn(B, 4, 32, 32, generator=g),
                     timesteps=torch.randint(0, 1000, (B, ), generator=g))
        batch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in {**b, **extra}.items()}
        batch['images'] = None
        out = engine.step(batch)
        losses.append(out['loss'].item())
        for grp, base in zip(ref_opt.param_groups, (1e-3, 1e-5, 1e-4)):
            grp['lr'] = base * max(0.0, (100 - step) / 100)
        ref_opt.zero_grad()
        l = trainer_ref.reference_forward(twin, None, b['prompts'], b['masks'], b['img_masks'], **extra)
        l.backward()
        ref_opt.step()
        ref_losses.append(l.item())
    print(f'[parity] train losses hip={losses} oracle={ref_losses}')
    for a, r in zip(losses, ref_losses):
        assert abs(a - r) <= 3e-2 * abs(r) + 1e-4
    # parameters after 3 steps
    num = den = 0.0
    for a, r in zip(tr.trainable_parameters(), trainer_ref.twin_parameters(twin)):
        num += (a.detach().float().cpu() - r.detach()).pow(2).sum().item()
        den += r.detach().pow(2).sum().item()
    rel = (num / den)**0.5
    print(f'[parity] parameters after 3 AdamW steps: rel_l2_diff={rel:.3e}')
    assert rel < 3e-3        # observed 1.1e-3 .. 1.2e-3 (bf16 path vs fp32 twin; run-to-run spread 2e-4 .. 3e-4)
    assert engine.global_step == 3 and not bool(engine.stop_flag)   # synthetic CLIP rows have real-CLIP-like norms


def test_eager_steps_run_on_the_updated_lora_factors():
    """torch.optim.AdamW(fused=True) -- the engine's optimiser on the device -- updates the fp32 LoRA masters WITHOUT bumping their
    tensor version counters, which is what the packed-operand cache (functional.LoraPackRegistry) watched: found in round 6, an
    eager step (train.hipgraph false, gradient accumulation, a batch of another shape) ran its forward on the factors of the step
    before. Now a backward that forms LoRA gradients and the engine's optimiser step both invalidate the cache. Check: after two
    eager steps at a learning rate that matters, the loss of the state the engine left behind equals, bit for bit, the loss after
    an explicit invalidation; and a foreign loop (plain torch fused AdamW, no engine) sees its own update in the next forward."""
    from bench import TRAIN_OPT, build_trainer, synthetic_batch
    from mixofshow.hip import functional as F_hip
    from mixofshow.pipelines.train_loop import TrainEngine
    tr = build_trainer('small', torch.device(DEV))
    torch.manual_seed(2)
    with torch.no_grad():
        for l in list(tr.text_encoder_lora) + list(tr.unet_lora):
            l.lora_up.weight.normal_(0, 0.05)
    opt = dict(TRAIN_OPT, optim_g=dict(TRAIN_OPT['optim_g']))
    engine = TrainEngine(tr, opt, total_iter=100, mixed_precision='bf16')
    for grp in engine.optimizer.param_groups:
        grp['lr'] = 3e-2
    engine.base_lrs = [3e-2 for _ in engine.base_lrs]
    g = torch.Generator().manual_seed(4)
    B = 2
    b = synthetic_batch(B, 256, 'cpu', 60)
    extra = dict(latents=torch.randn(B, 4, 32, 32, generator=g), noise=torch.randn(B, 4, 32, 32, generator=g),
                 timesteps=torch.randint(0, 1000, (B, ), generator=g))
    batch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in {**b, **extra}.items()}
    batch['images'] = None

    def loss_now():
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
            return tr(None, batch['prompts'], batch['masks'], batch['img_masks'], latents=batch['latents'], noise=batch['noise'],
                      timesteps=batch['timesteps']).float().item()

    l0 = loss_now()
    for _ in range(2):
        engine.step(batch)
    l_engine = loss_now()                    # whatever operands the cache holds after the engine's steps
    F_hip.invalidate_lora_packs()
    l_fresh = loss_now()
    print(f'[parity] eager steps and the LoRA operand cache: loss before {l0:.6f}, after two steps {l_engine:.6f}, after an explicit repack {l_fresh:.6f}')
    assert l_engine == l_fresh and abs(l_engine - l0) > 1e-4 * abs(l0)
    # a foreign training loop: backward through the fused layers + torch's fused AdamW, no engine
    params = [p for l in tr.unet_lora for p in (l.lora_down.weight, l.lora_up.weight)]
    foreign = torch.optim.AdamW(params, lr=3e-2, fused=True)
    foreign.zero_grad(set_to_none=True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        loss = tr(None, batch['prompts'], batch['masks'], batch['img_masks'], latents=batch['latents'], noise=batch['noise'],
                  timesteps=batch['timesteps'])
    loss.backward()
    v0 = params[0]._version
    foreign.step()
    assert params[0]._version == v0          # (the premise: this optimiser does not tell; if torch ever changes that, fine)
    l_after = loss_now()
    F_hip.invalidate_lora_packs()
    assert l_after == loss_now() and l_after != l_fresh


def test_sd15_fp16_train_step_through_vae_vs_cpu_twin():
    """BASELINE configs[1] itself: SD-1.5 architecture, 512x512, batch 4, fp16 autocast + GradScaler, images THROUGH the
    VAE (posterior-sample noise injected), attention regulariser on — one forward+backward against the oracle twin
    (fp32 activations, plain torch attention with full probability maps, 3-GEMM LoRA; frozen weights rounded to half
    where autocast rounds them). Reference: trainer_edlora.py:202-261, train_edlora.py:113-121."""
    import time
    from bench import TRAIN_OPT, build_trainer, synthetic_batch
    from mixofshow.pipelines.train_loop import TrainEngine
    from oracle import trainer_ref
    B, size = 4, 512
    tr = build_trainer('sd15', torch.device(DEV))
    torch.manual_seed(1)
    with torch.no_grad():
        for l in list(tr.text_encoder_lora) + list(tr.unet_lora):
            l.lora_up.weight.normal_(0, 0.02)
    engine = TrainEngine(tr, dict(TRAIN_OPT, optim_g=dict(TRAIN_OPT['optim_g'])), total_iter=100, mixed_precision='fp16')
    g = torch.Generator().manual_seed(3)
    b = synthetic_batch(B, size, 'cpu', 50)
    extra = dict(noise=torch.randn(B, 4, size // 8, size // 8, generator=g),
                 timesteps=torch.randint(0, 1000, (B, ), generator=g),
                 latent_noise=torch.randn(B, 4, size // 8, size // 8, generator=g))
    dev = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in {**b, **extra}.items()}
    params = tr.trainable_parameters()
    scale = engine.scaler.get_scale()
    for _ in range(6):                               # what GradScaler does on overflow: halve the scale and retry
        engine.bucket.zero()
        with torch.autocast('cuda', dtype=torch.float16):
            loss = tr(dev['images'], dev['prompts'], dev['masks'], dev['img_masks'], noise=dev['noise'],
                      timesteps=dev['timesteps'], latent_noise=dev['latent_noise'])
        (loss * scale).backward()
        if all(torch.isfinite(p.grad).all() for p in params):
            break
        scale /= 2
    grads = [p.grad.detach().float().cpu() / scale for p in params]
    assert all(torch.isfinite(x).all() for x in grads), 'overflow at every loss scale'
    twin_dev = os.environ.get('MOS_TWIN_DEVICE', 'cpu')
    t0 = time.time()
    twin = trainer_ref.make_reference_twin(tr, device=twin_dev, dtype=torch.float32, round_frozen_to=torch.float16)
    tb = {k: (v.to(twin_dev) if torch.is_tensor(v) else v) for k, v in {**b, **extra}.items()}
    loss_ref = trainer_ref.reference_forward(twin, tb['images'], tb['prompts'], tb['masks'], tb['img_masks'],
                                             noise=tb['noise'], timesteps=tb['timesteps'], latent_noise=tb['latent_noise'])
    loss_ref.backward()
    num = den = 0.0
    per = []
    for a, r in zip(grads, trainer_ref.twin_parameters(twin)):
        rg = r.grad.detach().float().cpu()
        num += (a - rg).pow(2).sum().item()
        den += rg.pow(2).sum().item()
        per.append(((a - rg).norm() / rg.norm().clamp_min(1e-30)).item())
    rel = (num / den) ** 0.5
    lrel = abs(loss.item() - loss_ref.item()) / abs(loss_ref.item())
    print(f'[parity] sd15 512x512 B4 fp16 train step through the VAE: loss hip {loss.item():.6f} oracle {loss_ref.item():.6f} '
          f'(rel {lrel:.2e}); grad rel-L2 {rel:.3e} (concept rows {per[0]:.2e}, worst tensor {max(per):.2e}); loss scale '
          f'{scale:g}; twin on {twin_dev} took {time.time() - t0:.1f}s')
    assert lrel <= 1e-3
    assert rel <= 1e-2
    # and the engine's full step (GradScaler, all-reduce of the bucket, fused AdamW, norm rule) runs on this config
    out = engine.step({k: dev[k] for k in ('images', 'prompts', 'masks', 'img_masks', 'noise', 'timesteps', 'latent_noise')})
    assert torch.isfinite(out['loss']) and abs(out['loss'].item() - loss.item()) <= 1e-3 * abs(loss.item())


def test_hipgraph_step_equals_eager_step():
    """TrainEngine.enable_graph replays the same kernels. Main check, free of optimiser chaos: on identical parameters
    the FIRST step's loss and flat gradient bucket of the graphed engine equal the eager engine's. Then 3 steps: same
    losses, and parameters inside the eager engine's own run-to-run spread (measured with 3 eager engines; Adam's first
    steps turn the sign of a noise-level gradient element into a full +-lr move, MIOpen's split-K backward uses atomics)."""
    from bench import TRAIN_OPT, build_trainer, synthetic_batch
    from mixofshow.pipelines.train_loop import TrainEngine
    B = 2
    g = torch.Generator().manual_seed(5)
    batches = []
    for step in range(3):
        b = synthetic_batch(B, 256, 'cpu', 70 + step)
        b.update(latents=torch.randn(B, 4, 32, 32, generator=g), noise=torch.randn(B, 4, 32, 32, generator=g),
                 timesteps=torch.randint(0, 1000, (B, ), generator=g))
        b = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in b.items()}
        b['images'] = None
        batches.append(b)
    results = []
    for graphed in (False, False, False, True):
        tr = build_trainer('small', torch.device(DEV))
        torch.manual_seed(1)
        with torch.no_grad():
            for l in list(tr.text_encoder_lora) + list(tr.unet_lora):
                l.lora_up.weight.normal_(0, 0.02)
        engine = TrainEngine(tr, dict(TRAIN_OPT, optim_g=dict(TRAIN_OPT['optim_g'])), total_iter=100,
                             mixed_precision='fp16')
        if graphed:
            engine.enable_graph(batches[0])
            assert engine._graph is not None
        losses, first_grad = [], None
        for b in batches:
            losses.append(engine.step(b)['loss'].item())
            if first_grad is None:
                first_grad = engine.bucket.flat.detach().float().clone() / engine.scaler.get_scale()
        results.append((losses, [p.detach().float().clone() for p in tr.trainable_parameters()], first_grad))

    def rel(pa, pb):
        num = sum((a - b).pow(2).sum().item() for a, b in zip(pa, pb))
        return (num / sum(a.pow(2).sum().item() for a in pa)) ** 0.5

    eager, (lg, pg, gg) = results[:3], results[3]
    g_spread = max(rel([eager[i][2]], [eager[j][2]]) for i in range(3) for j in range(i))
    g_graph = max(rel([e[2]], [gg]) for e in eager)
    p_spread = max(rel(eager[i][1], eager[j][1]) for i in range(3) for j in range(i))
    p_graph = max(rel(e[1], pg) for e in eager)
    print(f'[parity] hipgraph step: first-step gradient rel diff graph-eager {g_graph:.3e} (eager-eager {g_spread:.3e}); '
          f'losses eager {eager[0][0]} graph {lg}; params after 3 steps graph-eager {p_graph:.3e} (eager-eager {p_spread:.3e})')
    assert torch.isfinite(gg).all() and gg.abs().sum() > 0
    assert g_graph <= max(3.0 * g_spread, 1e-4)
    # first step: identical parameters -> same loss. Later steps inherit Adam's sign chaos on noise-level gradient elements
    # AND MIOpen choosing other convolution algorithms under capture than in the eager warm-up (three eager engines of one
    # process agree to 5e-5 on the step-2 loss, eager engines of different processes only to 2e-3): a stale or skipped
    # replay is off in the first digit (the three batches' losses are 4.3 / 0.31 / 1.05).
    assert abs(lg[0] - eager[0][0][0]) <= 2e-4 * abs(lg[0])
    for k, b in enumerate(lg):
        assert abs(b - eager[0][0][k]) <= 1e-2 * abs(b), (k, [e[0][k] for e in eager], b)
    assert p_graph <= max(3.0 * p_spread, 1e-3)


def test_hipgraph_replay_after_an_eager_step_of_another_batch_size():
    """the captured graph replays ONE launch of the deferred LoRA gradient sums with a record table and
    workspaces baked in. An eager step of another batch size (the ragged-last-batch fallback of TrainEngine._graph_step) or a
    second engine stepping in the same process must not rewrite them: after such a step the replayed gradients still equal
    an eager engine's on the same batch. Also: a frozen store refuses changes instead of redirecting replays."""
    from bench import TRAIN_OPT, build_trainer, synthetic_batch
    from mixofshow.hip import functional as F_hip
    from mixofshow.pipelines.train_loop import TrainEngine

    def batch(B, seed):
        g = torch.Generator().manual_seed(seed)
        b = synthetic_batch(B, 256, 'cpu', seed)
        b.update(latents=torch.randn(B, 4, 32, 32, generator=g), noise=torch.randn(B, 4, 32, 32, generator=g),
                 timesteps=torch.randint(0, 1000, (B, ), generator=g))
        b = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in b.items()}
        b['images'] = None
        return b

    def engine_for(seed=0):
        tr = build_trainer('small', torch.device(DEV))
        torch.manual_seed(1)
        with torch.no_grad():
            for l in list(tr.text_encoder_lora) + list(tr.unet_lora):
                l.lora_up.weight.normal_(0, 0.02)
        # lr 0: parameters stay identical, so every step's gradient bucket is comparable across engines
        opt = dict(TRAIN_OPT, optim_g=dict(TRAIN_OPT['optim_g']))
        e = TrainEngine(tr, opt, total_iter=100, mixed_precision='fp16')
        e.base_lrs = [0.0 for _ in e.base_lrs]
        for g in e.optimizer.param_groups:
            g['weight_decay'] = 0.0
        return e

    def grad_of(e, b):
        e.step(b)
        return e.bucket.flat.detach().float().clone() / e.scaler.get_scale()

    b2, b1, b2b = batch(2, 70), batch(1, 71), batch(2, 72)
    ref = engine_for()
    g_ref = [grad_of(ref, b2), grad_of(ref, b2b)]
    eng = engine_for()
    eng.enable_graph(b2)
    assert eng._finals_graph.frozen
    g0 = grad_of(eng, b2)                     # replay
    g_odd = grad_of(eng, b1)                  # batch of another size: eager fallback, other M -> other workspaces / table
    assert torch.isfinite(g_odd).all() and g_odd.abs().sum() > 0
    other = engine_for()                      # a second eager engine in the same process
    grad_of(other, b1)
    g1 = grad_of(eng, b2b)                    # replay again: must not sum stale partial buffers
    g0b = grad_of(eng, b2)

    def rel(a, b):
        return ((a - b).norm() / b.norm()).item()
    print(f'[parity] graph replay around eager steps: first {rel(g0, g_ref[0]):.3e}, after eager B=1 + second engine '
          f'{rel(g1, g_ref[1]):.3e}, repeat of batch 0 {rel(g0b, g0):.3e}')
    assert rel(g0, g_ref[0]) <= 5e-3 and rel(g1, g_ref[1]) <= 5e-3 and rel(g0b, g0) <= 5e-3
    # the frozen store itself refuses to be rewritten
    with pytest.raises(RuntimeError):
        with F_hip.direct_grad_accumulation(defer_finals=True, store=eng._finals_graph):
            eng._finals_graph.workspace(('no', 'such', 'group'), 16, torch.device(DEV, torch.cuda.current_device()))
    eng.disable_graph()


def test_update_quasi_newton_vs_reference_golden(golden):
    """The Gram-form fp64 L-BFGS (HIP) against the iterates the REAL reference code produced (fp32, direct form)."""
    from mixofshow.utils.lsq import update_quasi_newton
    from oracle import fusion_ref
    for name, c in golden['lbfgs'].items():
        W = update_quasi_newton(c['X'], c['Y'], c['W0'].clone(), c['iters'], DEV)
        assert W.shape == c['W'].shape and W.dtype == torch.float32 and W.device.type == 'cpu'
        l_ref = fusion_ref.lsq_loss_ref(c['X'].double(), c['Y'].double(), c['W'].double()).item()
        l_got = fusion_ref.lsq_loss_ref(c['X'].double(), c['Y'].double(), W.double()).item()
        l_0 = fusion_ref.lsq_loss_ref(c['X'].double(), c['Y'].double(), c['W0'].double()).item()
        rel_w = ((W - c['W']).norm() / (c['W'] - c['W0']).norm()).item()
        print(f'[parity] lbfgs[{name}]: loss0={l_0:.4e} ref={l_ref:.6e} hip={l_got:.6e} rel_dW_err={rel_w:.3e}')
        # same optimiser on the same objective: the loss reached must match the reference's, and the update
        # direction W - W0 must agree (fp32-vs-fp64 line-search noise only)
        # (measured rounds 3-5: rel_dW_err <= 2.5e-5 on all four problems)
        assert l_got <= l_ref * (1 + 2e-2) + 1e-12
        assert rel_w < 1e-3


def test_gradient_fusion_end_to_end(tmp_path):
    """compose_concepts on two synthetic ED-LoRA checkpoints: every fused layer must reduce its LSQ loss, the fused
    model must be saved in the diffusers layout + new_concept_cfg.json and be loadable by the regional pipeline."""
    import gradient_fusion as gf
    from bench import build_trainer
    ckpts = []
    for i, (a, b) in enumerate([('<potter1>', '<potter2>'), ('<thanos1>', '<thanos2>')]):
        tr = build_trainer('small', torch.device('cpu'), seed=i)
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
    pipe, new_cfg = gf.compose_concepts(cfg, 60, 15, 'synthetic://small?seed=0', str(tmp_path), 'base', DEV)
    assert list(new_cfg) == ['<potter1>', '<potter2>', '<thanos1>', '<thanos2>']
    assert new_cfg['<thanos1>']['concept_token_names'][0] == '<new32>'      # numbering advances by 16 per word
    out = tmp_path / 'combined_model_base'
    assert (out / 'unet' / 'diffusion_pytorch_model.safetensors').exists() and (out / 'new_concept_cfg.json').exists()
    for p in pipe.unet.parameters():
        assert torch.isfinite(p).all()


def test_fusion_feature_collection_and_fused_weights_vs_oracle_gpu(tmp_path, monkeypatch):
    """F2 on the device ('small' preset, fp16, real kernels): the (X, Y) statistics the product's hooks and fused-projection
    feature taps stream into the Gram accumulators vs the features the reference procedure stores (oracle/fusion_ref.py
    run on the same modules with the oracle's processors), two concepts; fused weights vs the oracle's solver."""
    from tests.test_fusion_cpu import fusion_parity_report, make_fusion_fixture, run_product_and_oracle_fusion
    cfg = make_fusion_fixture(tmp_path, 'small', n_concepts=2)
    res = run_product_and_oracle_fusion(cfg, 'small', torch.device(DEV), 40, 10, monkeypatch)
    # fp16 activations: the two paths' features differ by half-precision rounding of the attention outputs upstream
    fusion_parity_report(res, 2e-3, solve_layers=2)


def test_fusion_three_sd15_layers_over_14_concepts_vs_oracle(tmp_path, monkeypatch):
    """configs[3] at ITS scale (reference gradient_fusion.py:627-747): THREE spatial layers of the
    SD-1.5 UNet -- self-attention output projections (their input is what the HIP attention kernel wrote) at level 0
    (320 -> 320, the last block), level 1 (640 -> 640) and level 2 (1280 -> 1280) -- accumulated
    over 14 synthetic concepts x 20 recorded DPM-Solver steps x 4096 / 1024 / 256 tokens = 1,146,880 / 286,720 / 71,680 rows.
    The product streams them into fp64 Gram statistics through the feature taps of the fused projections; the oracle runs the
    reference procedure (forward hooks on the nn.Linear modules, features STORED, oracle processors) on the same model and seeds.
    Compared per layer: n, G = X^T X, P = Y^T X, c = sum Y^2, and the FUSED WEIGHT -- the product's Gram-form fp64 L-BFGS on its
    streamed statistics against the reference's own solver (torch.optim.LBFGS on the chunked fp32 closure over the stored
    features, update_quasi_newton :38-96), 50 iterations as fuse.sh sets for the UNet: excess loss over the exact minimum and
    distance to the exact minimiser, both evaluated with the oracle's fp64 statistics."""
    import gradient_fusion as gf
    from bench import synthetic_edlora_checkpoints
    from mixofshow.utils.lsq import lbfgs_on_gram
    from oracle import edlora_ref as R
    from oracle import fusion_ref as FR
    # (one projection KIND: the reference hooks every module of the kinds that occur in the LoRA keys, :640-660, and the oracle
    #  stores their features on the host -- 16 modules for one kind, as in round 4)
    layers = ['up_blocks.3.attentions.2.transformer_blocks.0.attn1.to_out.0',
              'down_blocks.1.attentions.0.transformer_blocks.0.attn1.to_out.0',
              'down_blocks.2.attentions.1.transformer_blocks.0.attn1.to_out.0']
    rows = {layers[0]: 4096, layers[1]: 1024, layers[2]: 256}
    n_concepts = 14
    cfg = synthetic_edlora_checkpoints('sd15', n_concepts, str(tmp_path))
    pipe, _, sched = gf.init_stable_diffusion('synthetic://sd15?seed=0', DEV)
    for p in list(pipe.text_encoder.parameters()) + list(pipe.unet.parameters()):
        p.requires_grad = False
    emb, te, kv, sp, concepts = gf.parse_new_concepts(cfg)
    _, ncfg = gf.merge_new_concepts_(emb, concepts, pipe.tokenizer, pipe.text_encoder)
    sp = [{k: v for k, v in d.items() if any(k.startswith(l + '.lora_') for l in layers)} for d in sp]
    assert all(len(d) == 2 * len(layers) for d in sp)
    captured = {}
    monkeypatch.setattr(gf, '_solve_layers', lambda accs, sd, iters, tag: captured.setdefault(tag, accs) and {})
    u0 = {k: v.detach().clone() for k, v in pipe.unet.state_dict().items()}
    torch.manual_seed(77)                     # decode_to_latents draws from the global CPU generator (reference :601)
    gf.merge_spatial_attention(concepts, 1, ncfg, pipe.tokenizer, pipe.text_encoder, pipe.unet, sp, sched, DEV)
    accs = captured['spatial']
    assert all(torch.equal(v, u0[k]) for k, v in pipe.unet.state_dict().items()), 'original weights not restored'
    # the reference procedure on the same modules: every projection a real nn.Linear call, features kept on the host
    for m in pipe.unet.modules():
        if m.__class__.__name__ == 'Attention':
            m.set_processor(R.PlainAttnProcessorRef())
    torch.manual_seed(77)
    Xs, Ys, _ = FR.merge_spatial_attention_ref(concepts, 1, ncfg, pipe.tokenizer, pipe.text_encoder, pipe.unet, sp, sched, DEV,
                                               R.bind_concept_prompt_ref, return_features=True)
    for layer in layers:
        acc = accs[layer + '.weight']
        X, Y = Xs[layer + '.weight'], Ys[layer + '.weight']
        n = n_concepts * 20 * rows[layer]
        C = acc.cin
        assert acc.n == n == X.shape[0] and X.shape[1] == Y.shape[1] == C == acc.cout
        G = torch.zeros(C, C, dtype=torch.float64, device=DEV)
        P = torch.zeros(C, C, dtype=torch.float64, device=DEV)
        c = torch.zeros((), dtype=torch.float64, device=DEV)
        for s0 in range(0, n, 131072):
            x, y = X[s0:s0 + 131072].to(DEV).double(), Y[s0:s0 + 131072].to(DEV).double()
            G += x.T @ x
            P += y.T @ x
            c += (y * y).sum()
        eg = ((acc.G - G).norm() / G.norm()).item()
        ep = ((acc.P - P).norm() / P.norm()).item()
        ec = abs(acc.c.item() - c.item()) / c.item()
        # the fused weight: product solver on its streamed statistics vs the reference solver on the stored features. 50 L-BFGS
        # iterations do not converge a 320 x 320 .. 1280 x 1280 problem, and the two closures differ in arithmetic (fp64 Gram form
        # vs chunked fp32 mean of squares), so the ITERATES separate (first device run: 0.32 of the update at level 0, with the
        # HIP iterate at the LOWER loss). What is comparable is how close each gets to the solution: the exact full-data loss
        # L(W) = (tr(W G W^T) - 2 tr(W P^T) + c) / (n C) from the ORACLE's fp64 statistics, and its minimum L* at W* = P G^-1.
        W0 = u0[layer + '.weight'].float()
        W_hip, _ = lbfgs_on_gram(W0.cpu(), acc, 50)
        W_ref = FR.update_quasi_newton_ref(X.to(DEV).float(), Y.to(DEV).float(), W0.to(DEV), 50).cpu()

        def full_loss(W):
            Wd = W.to(DEV).double()
            return (((Wd @ G) * Wd).sum() - 2.0 * (Wd * P).sum() + c).item() / (float(n) * C)

        W_star = torch.linalg.solve(G + 1e-10 * G.diagonal().mean() * torch.eye(C, dtype=torch.float64, device=DEV), P.T).T
        l0, l_ref, l_hip, l_star = full_loss(W0), full_loss(W_ref), full_loss(W_hip), full_loss(W_star)
        upd = (W_ref - W0.cpu()).norm().item()
        ew = ((W_hip.cpu() - W_ref).norm() / max(upd, 1e-30)).item()
        d_ref = (W_ref.double() - W_star.cpu()).norm().item()
        d_hip = (W_hip.double() - W_star.cpu()).norm().item()
        print(f'[parity] fusion sd15 {layer} ({C} -> {C}), {n_concepts} concepts, n = {n}: streamed HIP Gram statistics vs the '
              f'reference procedure: rel err G {eg:.3e} P {ep:.3e} c {ec:.3e}; FUSED WEIGHT after 50 iterations: full-data loss '
              f'start {l0:.6e}, reference solver {l_ref:.6e}, HIP {l_hip:.6e}, minimum {l_star:.6e}; distance to the minimiser '
              f'reference {d_ref:.3e} / HIP {d_hip:.3e}; |W_hip - W_ref| / |W_ref - W0| = {ew:.3e}')
        # both paths sample 20 free-running fp16 steps per concept; they differ by the half-precision rounding of every attention
        # layer upstream (HIP flash kernel vs baddbmm/softmax/bmm in fp16)
        assert max(eg, ep, ec) <= 5e-3
        # the HIP iterate is at least as good a solution of the reference's problem as the reference's own iterate
        assert l_hip - l_star <= (l_ref - l_star) * 1.02 + 1e-3 * (l0 - l_star), (l0, l_ref, l_hip, l_star)
        assert d_hip <= 1.05 * d_ref + 1e-6
        del X, Y


def test_fusion_reduces_layer_loss_on_real_features():
    """One spatial layer: Gram accumulated from streamed fp16 features == direct loss; L-BFGS lowers it."""
    from mixofshow.utils.lsq import GramAccumulator, lbfgs_on_gram
    from oracle import fusion_ref
    g = torch.Generator().manual_seed(9)
    n, C = 40960, 320
    X = torch.randn(n, C, generator=g).half()
    W0 = torch.randn(C, C, generator=g) * 0.05
    Y = (X.float() @ (W0 + 0.01 * torch.randn(C, C, generator=g)).T).half()
    acc = GramAccumulator(C, C, DEV)
    for s in range(0, n, 8192):                      # streamed like the forward hooks do
        acc.add(X[s:s + 8192].to(DEV), Y[s:s + 8192].to(DEV))
    W, loss = lbfgs_on_gram(W0, acc, 50)
    direct = fusion_ref.lsq_loss_ref(X.double(), Y.double(), W.double()).item()
    l0 = fusion_ref.lsq_loss_ref(X.double(), Y.double(), W0.double()).item()
    Wr = fusion_ref.update_quasi_newton_ref(X.float(), Y.float(), W0.clone(), 50)
    lr = fusion_ref.lsq_loss_ref(X.double(), Y.double(), Wr.double()).item()
    print(f'[parity] spatial-layer LSQ: loss0={l0:.4e} hip(gram,fp64)={direct:.6e} gram_loss={loss:.6e} oracle(fp32 direct)={lr:.6e}')
    # `loss` is the Gram-form value at the fp64 iterate; `direct` re-evaluates the fp32-rounded W the API returns
    assert loss <= direct * (1 + 1e-6) and direct < 1e-3 * l0 and direct <= lr * (1 + 5e-2)


@pytest.mark.timeout(1500)
def test_cli_chain_on_the_device_train_test_fuse_regional_sample(tmp_path):
    """The reference's four command lines, run as a user runs them (subprocesses of the repo's scripts) on the device, chained
    through their on-disk formats: `train_edlora.py -opt` (two concepts; hipGraph step) -> `.pth` delta checkpoints ->
    `test_edlora.py -opt` (merge at alpha, sample the validation prompt) -> `gradient_fusion.py --concept_cfg ...` (fused
    model directory + new_concept_cfg.json) -> `regionally_controlable_sampling.py --pretrained_model <that directory>
    --prompt_rewrite ...` (PNG + run record). `small` preset (SD-1.5 head dims 40 / 80, 768-wide text tower), 256 px."""
    import glob
    import json
    import subprocess
    import sys
    import yaml
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)

    def run(argv, cwd):
        p = subprocess.run([sys.executable] + argv, cwd=cwd, env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, f'{argv[0]} failed:\n{p.stdout[-1500:]}\n{p.stderr[-3000:]}'
        return p

    with open(os.path.join(root, 'options', 'train', 'EDLoRA', 'synthetic', '8101_EDLoRA_potter_synthetic_B4.yml')) as f:
        base = yaml.safe_load(f)
    ckpts = []
    for name, toks in (('potter', '<potter1>+<potter2>'), ('thanos', '<thanos1>+<thanos2>')):
        opt = json.loads(json.dumps(base))
        opt['name'] = f'cli_gpu_{name}'
        opt['models'].update(pretrained_path='synthetic://small?seed=0', new_concept_token=toks)
        tr = opt['datasets']['train']
        tr.update(num_images=4, dataset_enlarge_ratio=2, batch_size_per_gpu=2, image_size=256,
                  replace_mapping={'<TOK>': toks.replace('+', ' ')})
        tr['instance_transform'][0]['size'] = 256
        opt['datasets']['val_vis'].update(latent_size=[4, 32, 32], num_samples_per_prompt=1, batch_size_per_gpu=1,
                                          replace_mapping={'<TOK>': toks.replace('+', ' ')})
        opt['val'].update(val_during_save=False, alpha_list=[0.7], sample=dict(num_inference_steps=4, guidance_scale=7.5))
        opt['logger'] = dict(print_freq=1, save_checkpoint_freq=1000)
        recipe = str(tmp_path / f'{name}.yml')
        with open(recipe, 'w') as f:
            yaml.safe_dump(opt, f)
        # train_edlora.py resolves experiments/ under the directory of the script: run a copy of the entry script from tmp_path
        run([os.path.join(root, 'train_edlora.py'), '-opt', recipe], cwd=root)
        found = glob.glob(os.path.join(root, 'experiments', f'cli_gpu_{name}', 'models', 'edlora_model-latest.pth'))
        assert found, 'no checkpoint written'
        sd = torch.load(found[0], weights_only=False)['params']
        assert set(sd) == {'new_concept_embedding', 'text_encoder', 'unet'} and len(sd['new_concept_embedding']) == 2
        ups = [v for k, v in sd['unet'].items() if k.endswith('lora_up.weight')]
        assert ups and all(torch.isfinite(u).all() for u in ups) and any(u.abs().max() > 0 for u in ups)
        ckpts.append((found[0], toks.replace('+', ' '), recipe, opt))
    try:
        # test_edlora.py on the first checkpoint
        ck, words, recipe, opt = ckpts[0]
        opt = json.loads(json.dumps(opt))
        opt['name'] = 'cli_gpu_test'
        opt['path'] = dict(lora_path=ck)
        opt['models']['alpha'] = 0.7
        tpath = str(tmp_path / 'test.yml')
        with open(tpath, 'w') as f:
            yaml.safe_dump(opt, f)
        run([os.path.join(root, 'test_edlora.py'), '-opt', tpath], cwd=root)
        pngs = glob.glob(os.path.join(root, 'results', 'cli_gpu_test', 'visualization', '**', '*.png'), recursive=True)
        assert len(pngs) == 1 and '<potter1>' in os.path.basename(pngs[0])
        from PIL import Image
        assert Image.open(pngs[0]).size == (256, 256)
        # gradient_fusion.py over both checkpoints
        cfg = str(tmp_path / 'fuse.json')
        with open(cfg, 'w') as f:
            json.dump([dict(lora_path=c[0], unet_alpha=1.0, text_encoder_alpha=1.0, concept_name=c[1]) for c in ckpts], f)
        fused_dir = str(tmp_path / 'fused')
        run([os.path.join(root, 'gradient_fusion.py'), '--concept_cfg', cfg, '--save_path', fused_dir, '--pretrained_models',
             'synthetic://small?seed=0', '--optimize_textenc_iters', '20', '--optimize_unet_iters', '5'], cwd=root)
        model = os.path.join(fused_dir, 'combined_model_base')
        with open(os.path.join(model, 'new_concept_cfg.json')) as f:
            new_cfg = json.load(f)
        assert list(new_cfg) == ['<potter1>', '<potter2>', '<thanos1>', '<thanos2>']
        assert os.path.exists(os.path.join(model, 'unet', 'diffusion_pytorch_model.safetensors'))
        # regionally_controlable_sampling.py on the fused directory: two regions, 256 x 384
        out = str(tmp_path / 'regional')
        rewrite = ('[a <potter1> <potter2>, in the park]-*-[blurry]-*-[8, 8, 250, 180]|'
                   '[a <thanos1> <thanos2>, purple armor]-*-[blurry]-*-[4, 200, 256, 376]')
        run([os.path.join(root, 'regionally_controlable_sampling.py'), '--pretrained_model', model, '--prompt',
             'two people in the park', '--prompt_rewrite', rewrite, '--negative_prompt', 'lowres', '--seed', '14', '--height', '256',
             '--width', '384', '--save_dir', out, '--suffix', 'cli'], cwd=root)
        imgs = glob.glob(os.path.join(out, 'seed_14', '*.png'))
        recs = glob.glob(os.path.join(out, 'seed_14', '*.txt'))
        assert len(imgs) == 1 and len(recs) == 1 and Image.open(imgs[0]).size == (384, 256)
        import numpy as np
        px = np.asarray(Image.open(imgs[0]), dtype=np.float32)
        assert np.isfinite(px).all() and px.std() > 1.0           # an image, not a constant / NaN frame
        print(f'[parity] CLI chain on the device: 2 x train_edlora.py -> test_edlora.py -> gradient_fusion.py -> '
              f'regionally_controlable_sampling.py ok ({os.path.basename(imgs[0])})')
    finally:
        import shutil
        for d in ('cli_gpu_potter', 'cli_gpu_thanos'):
            shutil.rmtree(os.path.join(root, 'experiments', d), ignore_errors=True)
        shutil.rmtree(os.path.join(root, 'results', 'cli_gpu_test'), ignore_errors=True)
