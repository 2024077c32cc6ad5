"""Host code on its GPU-box branches, run on the CPU (fixture `gpu_branches`: kernels emulated by the oracle, device /
autocast queries answering like the GPU box). Covers the autograd plumbing around the kernels that `-m gpu` tests can only
reach on hardware: the residual add fused into LayerNorm, the layer-major text states, the fused-QKV CLIP attention."""
import pytest
import torch

import mixofshow.hip.functional as F_hip


def _ln(C):
    torch.manual_seed(3)
    norm = torch.nn.LayerNorm(C)
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5)
        norm.bias.uniform_(-0.5, 0.5)
    return norm.requires_grad_(False)


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('with_r', [True, False])
def test_add_layer_norm_half_stream_is_bit_identical_to_add_then_layer_norm(gpu_branches, dtype, with_r):
    norm = _ln(320)
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(2, 24, 320, generator=g).to(dtype)
    r0 = torch.randn(2, 24, 320, generator=g).to(dtype) if with_r else None
    wy = torch.randn(2, 24, 320, generator=g).to(dtype)
    ws = torch.randn(2, 24, 320, generator=g).to(dtype)

    def run(fused):
        F_hip._fuse_add_ln = fused
        x = x0.clone().requires_grad_(True)
        r = r0.clone().requires_grad_(True) if with_r else None
        s, y = F_hip.add_layer_norm(norm, x, r)
        assert s.dtype == dtype and y.dtype == dtype
        ((y * wy).float().sum() + (s * ws).float().sum()).backward()
        return s.detach(), y.detach(), x.grad, (r.grad if with_r else None)

    try:
        fused, plain = run(True), run(False)
    finally:
        F_hip._fuse_add_ln = True
    for a, b in zip(fused, plain):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a, b)


def test_add_layer_norm_fp32_stream_matches_fp32_reference(gpu_branches):
    """CLIP tower under autocast: fp32 residual stream + half branch; statistics and gradient from the fp32 sum."""
    norm = _ln(768)
    g = torch.Generator().manual_seed(1)
    x0 = torch.randn(3, 77, 768, generator=g)
    r0 = torch.randn(3, 77, 768, generator=g).bfloat16()
    wy = torch.randn(3, 77, 768, generator=g)
    ws = torch.randn(3, 77, 768, generator=g)
    x = x0.clone().requires_grad_(True)
    r = r0.clone().requires_grad_(True)
    with torch.autocast('cpu', dtype=torch.bfloat16):
        s, y = F_hip.add_layer_norm(norm, x, r)
    assert s.dtype == torch.float32 and y.dtype == torch.bfloat16 and s.grad_fn is not None
    ((y.float() * wy).sum() + (s * ws).sum()).backward()
    xr = x0.clone().requires_grad_(True)
    rr = r0.float().requires_grad_(True)
    sr = xr + rr
    yr = torch.nn.functional.layer_norm(sr, (768, ), norm.weight, norm.bias, norm.eps)
    ((yr * wy).sum() + (sr * ws).sum()).backward()
    assert torch.equal(s.detach(), sr.detach())
    assert (y.float() - yr).abs().max() <= 2e-2 * yr.abs().max()           # one bf16 rounding of the output
    # the output gradient arrives rounded to bf16 (y is bf16): relative L2 at bf16 level, no systematic term
    assert ((x.grad - xr.grad).norm() / xr.grad.norm()).item() <= 6e-3
    assert r.grad.dtype == torch.bfloat16 and torch.equal(r.grad, x.grad.bfloat16())


def test_add_layer_norm_passthrough_routes_the_bypass_gradient(gpu_branches):
    """r None: s is x itself; a consumer of s alone (norm output unused) still gets its gradient back to x."""
    norm = _ln(64)
    x = torch.randn(4, 64).half().requires_grad_(True)
    s, y = F_hip.add_layer_norm(norm, x)
    (s.float() * 2).sum().backward()
    assert torch.equal(x.grad, torch.full_like(x, 2.0))


@pytest.mark.parametrize('silu', [True, False])
def test_group_norm_tap_is_bit_identical_to_autograd_accumulation(gpu_branches, silu):
    """x feeds a GroupNorm and a skip path: with the tap the skip's gradient is added inside the norm's backward kernel."""
    torch.manual_seed(5)
    norm = torch.nn.GroupNorm(4, 64).requires_grad_(False)
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5)
        norm.bias.uniform_(-0.5, 0.5)
    x0 = torch.randn(2, 64, 8, 8).half().contiguous(memory_format=torch.channels_last)
    wy = torch.randn(2, 64, 8, 8).half().contiguous(memory_format=torch.channels_last)
    ws = torch.randn(2, 64, 8, 8).half().contiguous(memory_format=torch.channels_last)

    def run(fused):
        F_hip._fuse_gn_res = fused
        x = x0.clone(memory_format=torch.preserve_format).requires_grad_(True)
        skip, y = F_hip.group_norm_act(norm, x, silu, tap=True)
        ((y * wy).float().sum() + (skip * ws).float().sum()).backward()
        return y.detach(), x.grad

    try:
        (yf, gf), (yu, gu) = run(True), run(False)
    finally:
        F_hip._fuse_gn_res = True
    assert torch.equal(yf, yu) and torch.equal(gf, gu)
    assert gf.is_contiguous(memory_format=torch.channels_last)
    # and against fp32 autograd of the same expression
    xr = x0.float().requires_grad_(True)
    yr = torch.nn.functional.group_norm(xr, 4, norm.weight, norm.bias, norm.eps)
    yr = torch.nn.functional.silu(yr) if silu else yr
    ((yr * wy.float()).sum() + (xr * ws.float()).sum()).backward()
    assert ((gf.float() - xr.grad).norm() / xr.grad.norm()).item() <= 2e-3


def test_group_norm_tap_reads_a_concatenation_gradient_slice_in_place(gpu_branches, monkeypatch):
    """The skip path of a tapped GroupNorm ends in a torch.cat along the channels (the UNet's up blocks): autograd hands the tap a
    CHANNEL SLICE of the concatenation's gradient. The backward passes it to the kernel as it is (pixel stride of the wide tensor,
    `ops.nhwc_pixel_stride`) -- no contiguous copy -- and the result is the dense one bit for bit."""
    import mixofshow.hip.ops as ops
    torch.manual_seed(6)
    norm = torch.nn.GroupNorm(4, 64).requires_grad_(False)
    cl = torch.channels_last
    x0 = torch.randn(2, 64, 8, 8).half().contiguous(memory_format=cl)
    other = torch.randn(2, 32, 8, 8).half().contiguous(memory_format=cl)
    wy = torch.randn(2, 64, 8, 8).half().contiguous(memory_format=cl)
    wc = torch.randn(2, 96, 8, 8).half().contiguous(memory_format=cl)
    seen = []
    real = ops.groupnorm_silu_bwd

    def spy(dy, x, gamma, beta, stats, groups, silu, ds=None, **kw):
        seen.append(None if ds is None else (tuple(ds.stride()), ops.nhwc_pixel_stride(ds)))
        return real(dy, x, gamma, beta, stats, groups, silu, ds=ds, **kw)

    monkeypatch.setattr(ops, 'groupnorm_silu_bwd', spy)

    def run(first):
        x = x0.clone(memory_format=torch.preserve_format).requires_grad_(True)
        skip, y = F_hip.group_norm_act(norm, x, True, tap=True)
        cat = torch.cat([skip, other] if first else [other, skip], dim=1)
        torch.autograd.backward([y, cat], [wy, wc])          # channels_last gradients, as the UNet's kernels produce them
        return x.grad

    for first in (True, False):
        seen.clear()
        g = run(first)
        assert seen == [((96 * 64, 1, 96 * 8, 96), 96)], seen            # the slice itself reached the kernel call
        lo = 0 if first else 32
        xr = x0.clone(memory_format=torch.preserve_format).requires_grad_(True)
        skip, y = F_hip.group_norm_act(norm, xr, True, tap=True)
        torch.autograd.backward([y, skip], [wy, wc[:, lo:lo + 64].contiguous(memory_format=cl)])
        assert torch.equal(g, xr.grad)
    # layouts the kernels do not take in place: not a channel slice of a channels_last tensor / misaligned channel offset
    assert ops.nhwc_pixel_stride(torch.zeros(2, 64, 8, 8).half()) is None
    assert ops.nhwc_pixel_stride(wc[:, 4:68]) is None                    # 8-byte offset: 16-byte loads need channel offsets % 8
    assert ops.nhwc_pixel_stride(wc[:, 32:]) == 96 and ops.nhwc_pixel_stride(wc) == 96
    assert ops.nhwc_pixel_stride(wc[:, :, ::2]) is None


def _tiny_trainer_loss_and_grads(fuse, autocast=torch.float16):
    from tests.test_host_cpu import _batch, _trainer
    F_hip._fuse_add_ln = F_hip._fuse_gn_res = fuse
    tr = _trainer()
    tr.unet.train(), tr.text_encoder.train()
    with torch.autocast('cpu', dtype=autocast, enabled=autocast is not None):
        loss = tr(**_batch())
    loss.backward()
    grads = torch.cat([p.grad.reshape(-1).float() for p in tr.trainable_parameters()])
    return loss.detach().float(), grads


def test_trainer_step_on_gpu_branches_fused_vs_unfused_and_vs_plain_cpu_path(emulated_hip, tmp_path):
    """Whole ED-LoRA step through the kernel-backed autograd Functions (emulated, fp16 autocast): add+LN fusion on/off must
    agree to rounding (the UNet half stream is bit-identical, the CLIP fp32 stream drops one half rounding of the LN input),
    and both must agree with the trainer on its plain CPU branches (fp32 torch ops around the same primitives). Measured:
    fused vs plain 4.5e-3, unfused vs plain 4.6e-3, fused vs unfused 5.7e-3 rel L2 of the gradient."""
    lp, gp = _tiny_trainer_loss_and_grads(True, None)          # plain branches: before the device queries are patched
    real_enabled, real_dtype = torch.is_autocast_enabled, torch.get_autocast_dtype
    torch.Tensor.is_cuda = property(lambda self: True)
    torch.is_autocast_enabled = lambda device_type=None: real_enabled('cpu')
    torch.get_autocast_dtype = lambda device_type=None: real_dtype('cpu')
    try:
        lf, gf = _tiny_trainer_loss_and_grads(True)
        lu, gu = _tiny_trainer_loss_and_grads(False)
    finally:
        F_hip._fuse_add_ln = F_hip._fuse_gn_res = True
        del torch.Tensor.is_cuda
        torch.is_autocast_enabled, torch.get_autocast_dtype = real_enabled, real_dtype

    def rel(a, b):
        return ((a - b).norm() / b.norm()).item()

    print(f'[parity] tiny trainer on GPU branches (emulated): loss fused {lf:.6f} unfused {lu:.6f} plain {lp:.6f}; '
          f'grad rel L2 fused/plain {rel(gf, gp):.2e} unfused/plain {rel(gu, gp):.2e} fused/unfused {rel(gf, gu):.2e}')
    assert abs(lf - lp) <= 2e-3 * abs(lp) and abs(lu - lp) <= 2e-3 * abs(lp)
    assert rel(gf, gp) <= 1.5e-2 and rel(gu, gp) <= 1.5e-2 and rel(gf, gu) <= 1.5e-2


# ---- host logic introduced with the launch reduction (plain CPU: no fixtures needed) ----------------------------------------
@pytest.mark.parametrize('identity', [False, True])
@pytest.mark.parametrize('mask_hw', [32, 24, 16])
def test_vectorised_attention_regulariser_equals_the_reference_loop(identity, mask_hw):
    """EDLoRATrainer.cal_attn_reg (both token columns and all resolutions through shared kernels, strided-view mask
    down-sampling, no division by the head count) against the oracle's restatement of the reference loop
    (trainer_edlora.py:263-313) on random maps: value and gradient w.r.t. every map. mask 24 px: 16-px maps take the
    F.interpolate branch (24 % 16 != 0), 8-px maps the strided view."""
    import types
    from mixofshow.pipelines.trainer_edlora import EDLoRATrainer
    from oracle import edlora_ref as R
    g = torch.Generator().manual_seed(17)
    B, H, pos = 2, 4, [3, 5]
    full, cols = {'down_cross': [], 'up_cross': []}, {'down_cross': [], 'up_cross': []}
    leaves = []
    for place, res in (('down_cross', 16), ('down_cross', 8), ('up_cross', 8), ('up_cross', 16), ('up_cross', 16)):
        m = torch.softmax(torch.randn(B * H, res * res, 77, generator=g) * 2, -1).requires_grad_(True)
        leaves.append(m)
        full[place].append(m)
        cols[place].append(m.reshape(B, H, res * res, 77)[..., pos])
    masks = (torch.rand(B, 1, mask_hw, mask_hw, generator=g) > 0.4).float()
    ids = torch.zeros(B * 16, 77, dtype=torch.long)
    ids[:, pos[0]], ids[:, pos[1]] = 11, 12
    ref = R.cal_attn_reg_ref(full, masks, ids, {11, 12}, 0.01, identity, strict_resolutions=False)
    g_ref = torch.autograd.grad(ref, leaves)
    me = types.SimpleNamespace(attn_reg_weight=0.01, reg_full_identity=identity, _mask_at=EDLoRATrainer._mask_at)
    got = EDLoRATrainer.cal_attn_reg(me, cols, masks)
    g_got = torch.autograd.grad(got, leaves)
    assert abs(got.item() - ref.item()) <= 1e-6 * abs(ref.item()) + 1e-9
    for a, b in zip(g_got, g_ref):
        assert (a - b).abs().max() <= 1e-5 * b.abs().max() + 1e-12
    # the (finite value, valid flag) form the training step uses
    val, ok = EDLoRATrainer.cal_attn_reg(me, cols, masks, return_valid=True)
    assert bool(ok) and abs(val.item() - ref.item()) <= 1e-6 * abs(ref.item()) + 1e-9


def test_layer_major_text_states_cache_and_gradient():
    """attach_layer_major_states: the per-layer slices equal states[:, k]; their gradient is the gradient of slicing; the
    attachment is refreshed when the tensor changes in place or the grad mode differs, and processors fall back to slicing
    when it is stale or absent."""
    from mixofshow.models import edlora
    torch.manual_seed(2)
    st = torch.randn(2, 16, 7, 8, requires_grad=True)
    w = torch.randn(16, 2, 7, 8)
    edlora.attach_layer_major_states(st)
    layers = [edlora._select_layer_states(st, k) for k in range(16)]
    assert all(l.is_contiguous() and torch.equal(l, st[:, k]) for k, l in enumerate(layers))
    sum((l * w[k]).sum() for k, l in enumerate(layers)).backward()
    assert torch.allclose(st.grad, w.transpose(0, 1))
    # detached tensor: cached across calls of a sampling loop, recomputed after an in-place update
    pe = torch.randn(2, 16, 7, 8)
    edlora.attach_layer_major_states(pe)
    first = pe._mos_layers
    edlora.attach_layer_major_states(pe)
    assert pe._mos_layers is first
    pe.mul_(2.0)
    assert torch.equal(edlora._select_layer_states(pe, 3), pe[:, 3])          # stale attachment: falls back to slicing
    edlora.attach_layer_major_states(pe)
    assert pe._mos_layers is not first and torch.equal(pe._mos_layers[1][3], pe[:, 3])
    # attached under no_grad, used with grad: re-attached so that gradients flow
    q = torch.randn(1, 16, 3, 8, requires_grad=True)
    with torch.no_grad():
        edlora.attach_layer_major_states(q)
    edlora.attach_layer_major_states(q)
    edlora._select_layer_states(q, 0).sum().backward()
    assert q.grad is not None and float(q.grad[:, 0].sum()) == 24.0 and float(q.grad[:, 1:].abs().sum()) == 0.0


def test_time_embedding_activation_is_computed_once_per_embedding():
    from mixofshow.models import unet_2d_condition as U
    act = torch.nn.SiLU()
    calls = []
    spy = lambda t: (calls.append(1), act(t))[1]
    e = torch.randn(2, 8)
    a, b = U._act_once(spy, e), U._act_once(spy, e)
    assert a is b and len(calls) == 1
    e.add_(1.0)                                             # in-place change: recomputed
    c = U._act_once(spy, e)
    assert len(calls) == 2 and torch.equal(c, act(e))
    e2 = e.clone().requires_grad_(True)                     # another tensor: recomputed, graph attached
    d = U._act_once(spy, e2)
    assert len(calls) == 3 and d.requires_grad
    with torch.no_grad():
        f = U._act_once(spy, e2)                            # same tensor, grad mode differs: not the cached graph-carrying one
    assert len(calls) == 4 and not f.requires_grad


def test_edlora_sampling_on_gpu_branches_fp16_vs_fp32_plain_path(emulated_hip):
    """The ED-LoRA sampling pipeline in fp16 through the kernel-backed branches (no grad: add+LayerNorm forward fusion,
    GroupNorm / conv Functions, layer-major text states under CFG) against the same pipeline in fp32 on the plain CPU branches.
    4 steps on the tiny model: the difference is the fp16 rounding of weights and activations."""
    from mixofshow.pipelines.pipeline_edlora import EDLoRAPipeline
    from tests.test_pipelines_cpu import _concept_cfg
    kw = dict(prompt=['a <potter1> <potter2> in the park', 'a photo of a dog'], negative_prompt=['blurry', ''],
              height=64, width=64, num_inference_steps=4, guidance_scale=7.5, output_type='latent')
    latents = torch.randn((2, 4, 8, 8), generator=torch.manual_seed(1))

    def build(dtype):
        pipe = EDLoRAPipeline.from_pretrained('synthetic://tiny?seed=0', torch_dtype=dtype)
        pipe.set_new_concept_cfg(_concept_cfg(pipe.tokenizer, pipe.text_encoder, ['<potter1>', '<potter2>']))
        return pipe

    ref = build(torch.float32)(latents=latents.clone(), **kw).images
    pipe = build(torch.float16)
    real_enabled, real_dtype = torch.is_autocast_enabled, torch.get_autocast_dtype
    torch.Tensor.is_cuda = property(lambda self: True)
    torch.is_autocast_enabled = lambda device_type=None: real_enabled('cpu')
    torch.get_autocast_dtype = lambda device_type=None: real_dtype('cpu')
    try:
        out = pipe(latents=latents.clone().half(), **kw).images
    finally:
        del torch.Tensor.is_cuda
        torch.is_autocast_enabled, torch.get_autocast_dtype = real_enabled, real_dtype
    assert out.dtype == torch.float16 and torch.isfinite(out).all()
    err = (out.float() - ref).abs().max().item()
    scale = max(1.0, ref.abs().max().item())
    print(f'[parity] tiny EDLoRA sampling, fp16 on GPU branches (emulated) vs fp32 plain path, 4 steps: max|d| = {err:.3e} on scale {scale:.2f}')
    assert err <= 3e-2 * scale


def test_batched_time_embedding_projections_equal_per_block_projections(emulated_hip, monkeypatch):
    """MOS_BATCH_TEMB (on by default since round 4, device tensors only): one baddbmm per output width for the time-embedding projections of all ResNet blocks
    == the per-block nn.Linear calls, in the outputs of the whole UNet; the stacks follow weight updates."""
    from mixofshow.models import unet_2d_condition as U
    from mixofshow.utils import pretrained
    torch.manual_seed(0)
    unet = pretrained.load_unet('synthetic://tiny?seed=0').eval().requires_grad_(False)
    x = torch.randn(2, 4, 8, 8)
    t = torch.tensor([10, 500])
    ehs = torch.randn(2, 77, unet.config.cross_attention_dim)

    def run():
        with torch.no_grad():
            return unet(x, t, ehs).sample

    monkeypatch.setattr(U, '_batch_time_proj', False)
    ref = run()
    monkeypatch.setattr(U, '_batch_time_proj', 'force')          # (the batched form is taken on device tensors only)
    got = run()
    assert unet._time_projections.usable() and len(unet._time_projections.blocks) >= 4
    assert (got - ref).abs().max() <= 1e-4 * ref.abs().max()          # (the emulated attention rounds its operands to half)
    with torch.no_grad():                                        # a changed projection is picked up (version-keyed stacks)
        unet.down_blocks[0].resnets[0].time_emb_proj.bias.add_(1.0)
    got2 = run()
    monkeypatch.setattr(U, '_batch_time_proj', False)
    ref2 = run()
    assert not torch.allclose(ref2, ref) and (got2 - ref2).abs().max() <= 1e-4 * ref2.abs().max()
    unet.down_blocks[0].resnets[0].time_emb_proj.weight.requires_grad_(True)      # trainable projection: not batched ...
    assert not unet._time_projections.usable()
    with torch.no_grad():                 # ... except under no_grad: the sampling pipelines never freeze their modules, and the
        assert unet._time_projections.usable()      # round-4 rocprofv3 trace of the regional sample showed 24 tiny GEMMs per call


# ---- round 4: GEMM residual epilogue on the feed-forward and the 1x1 proj_out --------------------------------------------
def test_residual_epilogue_emulation_equals_projection_then_add():
    """The emulated linear_fwd_ex(residual=...) == GEMM (rounded) followed by the half add."""
    from oracle import emu_ops
    g = torch.Generator().manual_seed(2)
    x = torch.randn(24, 64, generator=g).half()
    W = (torch.randn(256, 64, generator=g) * 0.1).half()
    b = torch.randn(256, generator=g) * 0.1
    h = emu_ops.linear_fwd(x, W, bias=b)
    r = torch.randn(24, 256, generator=g).half()
    got_r, _ = emu_ops.linear_fwd_ex(x, W, None, None, b, residual=r)
    assert torch.equal(got_r, (h.float() + r.float()).half())


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_feed_forward_with_fused_residual_epilogue(gpu_branches, dtype):
    """FeedForward(n, residual=x): GEMM + geglu kernel, FF2 with the add in its epilogue, is bit-identical to `ff(n) + x` with the
    separate add, gradients included; a residual stream of ANOTHER dtype keeps the unfused `linear(x) + residual`."""
    from mixofshow.models.unet_2d_condition import FeedForward
    torch.manual_seed(5)
    ff = FeedForward(64).requires_grad_(False)
    for p in ff.parameters():
        p.data = p.data.to(dtype)
    g = torch.Generator().manual_seed(6)
    n0 = torch.randn(2, 24, 64, generator=g).to(dtype)
    x0 = torch.randn(2, 24, 64, generator=g).to(dtype)
    wy = torch.randn(2, 24, 64, generator=g).to(dtype)

    def run(fused):
        F_hip._ff2_own = fused
        n = n0.clone().requires_grad_(True)
        x = x0.clone().requires_grad_(True)
        y = ff(n, residual=x) if fused else ff(n) + x
        (y * wy).float().sum().backward()
        return y.detach(), n.grad, x.grad

    try:
        fused, plain = run(True), run(False)
    finally:
        F_hip._ff2_own = True
    assert torch.equal(fused[2], plain[2]) and torch.equal(fused[2], wy)      # the residual's gradient IS dy
    # forward: own GEMM (fp32 accumulate, one rounding, then the half add) vs torch's CPU half linear + add
    assert (fused[0].float() - plain[0].float()).abs().max() <= 2 ** -6 * plain[0].float().abs().max()
    assert ((fused[1].float() - plain[1].float()).norm() / plain[1].float().norm()).item() <= 2e-2
    with torch.no_grad():
        y32 = ff(n0, residual=x0.float())             # fp32 residual stream: NOT folded into the half epilogue
        assert y32.dtype == torch.float32
        want = ff(n0).float() + x0.float()
        assert torch.equal(y32, want)


def test_conv1x1_residual_epilogue_matches_conv_plus_residual(gpu_branches):
    """Transformer2DModel.proj_out(x) + residual with the add in the GEMM's epilogue: same values as the separate add,
    gradient of the residual = dy, layout kept."""
    torch.manual_seed(7)
    conv = torch.nn.Conv2d(64, 32, 1).requires_grad_(False).half()
    g = torch.Generator().manual_seed(8)
    cl = torch.channels_last
    x0 = torch.randn(2, 64, 6, 8, generator=g).half().contiguous(memory_format=cl)
    r0 = torch.randn(2, 32, 6, 8, generator=g).half().contiguous(memory_format=cl)
    wy = torch.randn(2, 32, 6, 8, generator=g).half().contiguous(memory_format=cl)

    def run(fused):
        F_hip._gemm_residual = fused
        x = x0.clone(memory_format=torch.preserve_format).requires_grad_(True)
        r = r0.clone(memory_format=torch.preserve_format).requires_grad_(True)
        y = F_hip.conv1x1(conv, x, residual=r)
        (y * wy).float().sum().backward()
        return y.detach(), x.grad, r.grad

    try:
        fused, plain = run(True), run(False)
    finally:
        F_hip._gemm_residual = True
    for a, b in zip(fused, plain):
        assert a.shape == b.shape and torch.equal(a, b)
    assert fused[0].is_contiguous(memory_format=cl)


@pytest.mark.parametrize('cross_dim', [64])
def test_transformer_block_attention_residual_in_the_out_projection_epilogue(gpu_branches, monkeypatch, cross_dim):
    """Round 6: `attn(norm(h)) + h` of BasicTransformerBlock with the add in the epilogue of the
    out-projection GEMM (mos_lora_linear_fwd_ex) instead of in the LayerNorm kernel that follows. Same rounding points, so the
    block's output AND its gradients (input, LoRA factors) are bit-identical to the separate-add form; both attention layers of
    a block take the epilogue; a processor that does not run the fused layer leaves the block on the old path."""
    import mixofshow.hip.ops as ops
    import mixofshow.models.unet_2d_condition as U
    from mixofshow.models.edlora import EDLoRA_AttnProcessor, LoRALinearLayer
    torch.manual_seed(0)
    C, heads = 64, 8
    blk = U.BasicTransformerBlock(C, heads, C // heads, cross_dim).half().requires_grad_(False)
    blk.attn2.set_processor(EDLoRA_AttnProcessor(0))
    loras = []
    for name in ('to_q', 'to_k', 'to_v'):
        for attn in (blk.attn1, blk.attn2):
            loras.append(LoRALinearLayer(name, getattr(attn, name), rank=4, alpha=1.0))
    for attn in (blk.attn1, blk.attn2):
        loras.append(LoRALinearLayer('to_out.0', attn.to_out[0], rank=4, alpha=1.0))
    with torch.no_grad():
        for l in loras:
            l.lora_up.weight.normal_(0, 0.05)
    g = torch.Generator().manual_seed(1)
    x0 = torch.randn(2, 48, C, generator=g).half()
    ehs = torch.randn(2, 16, 77, cross_dim, generator=g).half()
    wy = torch.randn(2, 48, C, generator=g).half()
    calls = []
    real = ops.linear_fwd_ex

    def spy(x, W, A16, Bp16, bias, residual=None, **kw):
        calls.append(residual is not None)
        return real(x, W, A16, Bp16, bias, residual=residual, **kw)
    monkeypatch.setattr(ops, 'linear_fwd_ex', spy)

    def run(flag):
        monkeypatch.setattr(U, '_attn_out_residual', flag)
        calls.clear()
        for l in loras:
            l.lora_down.weight.grad = l.lora_up.weight.grad = None
        x = x0.clone().requires_grad_(True)
        y = blk(x, encoder_hidden_states=ehs)
        (y * wy).float().sum().backward()
        return [y.detach(), x.grad] + [None if p.grad is None else p.grad.clone() for l in loras for p in (l.lora_down.weight, l.lora_up.weight)], sum(calls)

    fused, n_fused = run(True)
    plain, n_plain = run(False)
    assert n_fused == n_plain + 2                          # attn1 and attn2 out-projections took the residual epilogue
    for a, b in zip(fused, plain):
        assert a is not None and torch.equal(a, b)
    assert '_mos_residual' not in blk.attn1.__dict__ and '_mos_residual' not in blk.attn2.__dict__

    class Foreign:                                           # a processor outside the fused path: never sees the side channel
        def __call__(self, attn, hidden_states, encoder_hidden_states=None, **kw):
            return hidden_states * 0.5
    blk.attn1.set_processor(Foreign())
    out, n = run(True)
    assert n == n_plain + 1 and '_mos_residual' not in blk.attn1.__dict__


def test_groupnorm_statistics_travel_with_the_convolution_output(gpu_branches, monkeypatch):
    """Round 6: on large maps the 3x3 convolution's epilogue leaves the GroupNorm statistics of its output
    with the output tensor and the norm that consumes it skips its statistics pass (mos_conv3x3_nhwc_gn ->
    mos_groupnorm_silu_fwd_nhwc_pre; kernels emulated here: the emulated norm really computes from the attached sums). The
    attachment is only honoured for the very tensor it was made for, unmodified since."""
    import mixofshow.hip.ops as ops
    import mixofshow.models.unet_2d_condition as U
    torch.manual_seed(0)
    blk = U.ResnetBlock2D(64, 64, 128).half().requires_grad_(False).to(memory_format=torch.channels_last)
    nxt = U.ResnetBlock2D(64, 64, 128).half().requires_grad_(False).to(memory_format=torch.channels_last)
    g = torch.Generator().manual_seed(2)
    x0 = torch.randn(2, 64, 48, 48, generator=g).half().contiguous(memory_format=torch.channels_last)     # HW = 2304: "large"
    temb = torch.randn(2, 128, generator=g).half()
    used = []
    real = ops.groupnorm_silu_fwd

    def spy(x, gamma, beta, groups, eps, silu, force_slices=False, chan_part=None):
        used.append(chan_part is not None)
        return real(x, gamma, beta, groups, eps, silu, force_slices=force_slices, chan_part=chan_part)
    monkeypatch.setattr(ops, 'groupnorm_silu_fwd', spy)

    monkeypatch.setattr(F_hip, '_gn_from_conv_always', False)       # first: statistics only where the norm would read twice

    def run(flag, grad):
        monkeypatch.setattr(F_hip, '_gn_from_conv', flag)
        used.clear()
        x = x0.clone().requires_grad_(grad)
        with torch.set_grad_enabled(grad):
            y = nxt(blk(x, temb), temb)
            if grad:
                y.float().square().mean().backward()
        return y.detach(), (x.grad if grad else None), list(used)

    for grad in (False, True):
        y1, g1, u1 = run(True, grad)
        y0, g0, u0 = run(False, grad)
        # blk.norm1 sees the block input (no producer), blk.norm2 conv1's output, nxt.norm1 conv2's output, nxt.norm2 conv1's
        assert u1 == [False, True, True, True] and u0 == [False] * 4
        torch.testing.assert_close(y1.float(), y0.float(), rtol=0, atol=4e-3)
        if grad:
            assert (g1.float() - g0.float()).norm() <= 2e-2 * g0.float().norm()
    # the attachment belongs to ONE tensor in ONE state
    monkeypatch.setattr(F_hip, '_gn_from_conv', True)
    with torch.no_grad():
        y = F_hip.conv3x3(blk.conv1, x0, gn_groups=32)
        assert F_hip._producer_gn_stats(y) is not None and y._mos_gn_part[0].shape == (2, 1, 64, 2)
        assert F_hip._producer_gn_stats(y + 0) is None and F_hip._producer_gn_stats(torch.cat([y, y], 1)) is None
        y.add_(1.0)
        assert F_hip._producer_gn_stats(y) is None
        xs = x0[:, :, :16, :16].contiguous(memory_format=torch.channels_last)
        small = F_hip.conv3x3(blk.conv1, xs, gn_groups=32)
        assert F_hip._producer_gn_stats(small) is None         # MOS_GN_FROM_CONV=1: small maps keep the one-launch column norm
        monkeypatch.setattr(F_hip, '_gn_from_conv_always', True)
        assert F_hip._producer_gn_stats(F_hip.conv3x3(blk.conv1, xs, gn_groups=32)) is not None      # the default: every map


def test_conv3x3_backward_reads_a_concatenation_gradient_slice_in_place(gpu_branches, monkeypatch):
    """A ResnetBlock2D whose output goes into a torch.cat (every up-block layer): conv2's backward receives a channel slice of the
    concatenation's gradient and hands it to the dX convolution -- and, as the residual's gradient, to the tapped norm1 -- as it is;
    MOS_GRADS_IN_PLACE=0 (the round-5 behaviour: a contiguous copy first) gives the same input gradient bit for bit."""
    import mixofshow.hip.ops as ops
    import mixofshow.models.unet_2d_condition as U
    torch.manual_seed(1)
    cl = torch.channels_last
    blk = U.ResnetBlock2D(64, 64, 128).half().requires_grad_(False).to(memory_format=cl)
    g = torch.Generator().manual_seed(3)
    x0 = torch.randn(2, 64, 16, 16, generator=g).half().contiguous(memory_format=cl)
    skip = torch.randn(2, 64, 16, 16, generator=g).half().contiguous(memory_format=cl)
    temb = torch.randn(2, 128, generator=g).half()
    gcat = torch.randn(2, 128, 16, 16, generator=g).half().contiguous(memory_format=cl)
    conv_in, tap_ds = [], []
    real_conv, real_gn = ops.conv3x3_nhwc, ops.groupnorm_silu_bwd

    def spy_conv(x, *a, **k):
        conv_in.append(ops.nhwc_pixel_stride(x))
        return real_conv(x, *a, **k)

    def spy_gn(dy, x, gamma, beta, stats, groups, silu, ds=None, **kw):
        tap_ds.append(None if ds is None else ops.nhwc_pixel_stride(ds))
        return real_gn(dy, x, gamma, beta, stats, groups, silu, ds=ds, **kw)

    monkeypatch.setattr(ops, 'conv3x3_nhwc', spy_conv)
    monkeypatch.setattr(ops, 'groupnorm_silu_bwd', spy_gn)

    def run(in_place, first):
        monkeypatch.setattr(F_hip, '_grads_in_place', in_place)
        x = x0.clone(memory_format=torch.preserve_format).requires_grad_(True)
        y = blk(x, temb)
        cat = torch.cat([y, skip] if first else [skip, y], dim=1)
        conv_in.clear(), tap_ds.clear()
        cat.backward(gcat)
        return x.grad, list(conv_in), list(tap_ds)

    for first in (True, False):
        g1, c1, t1 = run(True, first)
        g0, c0, t0 = run(False, first)
        assert c1 == [128, 64] and c0 == [64, 64]        # dX of conv2 reads the slice (pixel stride of the 128-channel tensor)
        assert t1 == [None, 128] and t0 == [None, 64]     # norm2: no bypass; norm1's tap: the residual's gradient, the same slice
        assert torch.equal(g1, g0)


def test_lora_operand_cache_repacks_after_a_backward_and_after_silent_updates(gpu_branches, monkeypatch):
    """functional.LoraPackRegistry (host logic; the pack launch replaced by a counter that really packs): one repack for any number
    of groups when something changed; NONE while nothing did; a repack after an update the version counters do not show
    (torch's fused AdamW writes parameters without bumping `_version`: emulated here with `.data` writes under no version bump)
    once a backward pass has formed LoRA gradients -- round 6, the eager training steps ran on stale operands before."""
    import mixofshow.hip.ops as ops
    dev = torch.device('cpu')
    reg = F_hip.LoraPackRegistry(dev, torch.float16)
    monkeypatch.setitem(F_hip._registries, ('cpu', None, torch.float16), reg)
    packs = []

    def fake_pack_all(desc_dev, n_groups, max_elems, dtype):
        packs.append(n_groups)
        for g in reg.groups.values():                       # what the kernel does: masters -> packed half operands
            downs, ups = g.params()
            A16, A16T, Bp16, BpT = ops.lora_pack(downs, ups, g.alphas, g.K, dtype, dev)
            for dst, src in zip(g.bufs, (A16, A16T, Bp16, BpT)):
                dst.copy_(src)

    monkeypatch.setattr(ops, 'lora_pack_all', fake_pack_all)
    from mixofshow.hip import lib as _lib
    monkeypatch.setattr(ops, 'lora_group_desc', lambda *a, **k: _lib.LoraGroup())      # (descriptors hold device pointers: unused here)
    torch.manual_seed(0)
    K, N, r = 64, 32, 4
    d1, u1 = torch.randn(r, K).requires_grad_(True), torch.randn(N, r).requires_grad_(True)
    d2, u2 = torch.randn(r, K).requires_grad_(True), torch.randn(N, r).requires_grad_(True)
    if True:
        reg.get((d1, ), (u1, ), (1.0, ), K)
        reg.get((d2, ), (u2, ), (0.5, ), K)
        n0 = len(packs)
        for _ in range(3):                                  # steady state: nothing changed, nothing launched
            reg.get((d1, ), (u1, ), (1.0, ), K)
            reg.get((d2, ), (u2, ), (0.5, ), K)
        assert len(packs) == n0
        with torch.no_grad():
            d1.mul_(2.0)                                    # a visible update (version counter): ONE repack for both groups
        a1 = reg.get((d1, ), (u1, ), (1.0, ), K)[0].clone()
        reg.get((d2, ), (u2, ), (0.5, ), K)
        assert len(packs) == n0 + 1 and packs[-1] == 2
        torch.testing.assert_close(a1[:r].float(), d1.detach().half().float())
        # a silent update: .data writes leave `_version` alone, like torch.optim.AdamW(fused=True)
        v = d1._version
        d1.data.mul_(0.5)
        assert d1._version == v
        stale = reg.get((d1, ), (u1, ), (1.0, ), K)[0]
        assert len(packs) == n0 + 1 and not torch.allclose(stale[:r].float(), d1.detach().half().float())     # the cache cannot see it ...
        # ... but a backward pass through a fused LoRA layer announces that an optimiser step follows
        x = torch.randn(8, K).half().requires_grad_(True)
        W = torch.randn(N, K).half()
        with torch.autocast('cpu', dtype=torch.float16):
            y = F_hip.lora_linear(x, W, W.t().contiguous(), None, [(d2, u2, 0.5)])
        e0 = reg.epoch
        y.float().sum().backward()
        assert reg.epoch > e0
        fresh = reg.get((d1, ), (u1, ), (1.0, ), K)[0]
        assert len(packs) == n0 + 2
        torch.testing.assert_close(fresh[:r].float(), d1.detach().half().float())
        # and the engine says so itself after its optimiser step
        e1 = reg.epoch
        F_hip.invalidate_lora_packs()
        assert reg.epoch == e1 + 1
