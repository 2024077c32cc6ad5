"""Generate golden vectors by executing the REAL reference code (read-only at /root/reference).

Runs only in the authoring container (the reference does not exist on the GPU box); the outputs are
committed as small fixtures next to this script and pin oracle/*.py (tests/test_oracle_golden.py).

The reference imports diffusers / omegaconf / torchvision / cv2 / IPython / xformers at module top and
none of them is installed. Their NAMES are satisfied with empty stub modules (a meta-path finder below);
no arithmetic comes from a stub. The one third-party object the reference's processors really compute
with — diffusers' `Attention` — is oracle/attention_shim.py (restated from diffusers 0.19.3, see
SURVEY.md App. A). Everything else that runs here is the reference's own code:
  mixofshow/models/edlora.py            LoRALinearLayer, EDLoRA_AttnProcessor, EDLoRA_Control_AttnProcessor
  mixofshow/utils/ptp_util.py           AttentionStore
  mixofshow/pipelines/trainer_edlora.py EDLoRATrainer.cal_attn_reg (unbound, fake self)
  mixofshow/pipelines/pipeline_edlora.py bind_concept_prompt
  mixofshow/pipelines/pipeline_regionally_t2iadapter.py RegionT2I_AttnProcessor
  mixofshow/utils/convert_edlora_to_diffusers.py merge_lora_into_weight
  gradient_fusion.py                    chunk_compute_mse, update_quasi_newton, merge_lora_into_weight
  regionally_controlable_sampling.py    prepare_text

Usage:  python tests/golden/make_golden.py [output.pt]            -> reference_golden.pt
        python tests/golden/make_golden.py fusion [output.pt]     -> reference_fusion_golden.pt (G8, see fusion_golden)
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.normpath(os.path.join(HERE, '..', '..'))
REF = '/root/reference'
STUB_ROOTS = ('diffusers', 'omegaconf', 'torchvision', 'cv2', 'IPython', 'xformers')


class _Anything:
    """Placeholder class for names imported from stubbed modules (never called for arithmetic)."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        return _Anything()


class _StubMeta(type):
    """Class attributes of stubbed names (e.g. `diffusers.utils.logging.get_logger`) resolve to no-op callables."""

    def __getattr__(cls, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return _Anything()


class _StubModule(types.ModuleType):

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        if name == 'is_xformers_available':
            return lambda: False
        if name == 'check_min_version':
            return lambda *_a, **_k: None
        cls = _StubMeta(name, (_Anything,), {})
        setattr(self, name, cls)
        return cls


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):

    def find_spec(self, fullname, path, target=None):
        if fullname.split('.')[0] in STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def _import_reference():
    # resolve the transformers / accelerate names the reference imports BEFORE the stub finder exists: their
    # lazy importers probe find_spec('torchvision') and would otherwise mistake the stub for the real thing
    import accelerate  # noqa: F401
    import accelerate.logging  # noqa: F401
    import accelerate.utils  # noqa: F401
    import transformers
    from transformers import CLIPTextModel, CLIPTokenizer  # noqa: F401
    sys.meta_path.insert(0, _StubFinder())
    np.Inf = np.inf  # gradient_fusion.py:59 uses np.Inf (removed in numpy 2)
    # only the NAME is needed (type annotation in RegionallyT2IAdapterPipeline.__init__); transformers 5 dropped it
    # REF must come BEFORE the repo root: the repo root holds the product's own gradient_fusion.py /
    # regionally_controlable_sampling.py (same script names as the reference); only `oracle` is taken from the repo
    sys.path.insert(0, REPO)
    sys.path.insert(0, REF)
    import gradient_fusion as ref_fusion
    assert ref_fusion.__file__.startswith(REF), ref_fusion.__file__
    # (the transformers module object in sys.modules is swapped during the imports above, so patch it late)
    sys.modules['transformers'].__dict__['CLIPFeatureExtractor'] = type('CLIPFeatureExtractor', (), {})
    import regionally_controlable_sampling as ref_region_cli
    from mixofshow.models import edlora as ref_edlora
    from mixofshow.pipelines import pipeline_edlora as ref_pipe
    from mixofshow.pipelines import pipeline_regionally_t2iadapter as ref_region
    from mixofshow.pipelines import trainer_edlora as ref_trainer
    from mixofshow.utils import convert_edlora_to_diffusers as ref_convert
    from mixofshow.utils import ptp_util as ref_ptp
    assert ref_edlora.__file__.startswith(REF), ref_edlora.__file__
    return dict(fusion=ref_fusion, region_cli=ref_region_cli, edlora=ref_edlora, pipe=ref_pipe, region=ref_region,
                trainer=ref_trainer, convert=ref_convert, ptp=ref_ptp)


def _seeded_attention(shim_cls, C, cross, heads, seed):
    torch.manual_seed(seed)
    attn = shim_cls(C, cross_attention_dim=cross, heads=heads, dim_head=C // heads)
    with torch.no_grad():
        for p in attn.parameters():
            p.copy_(torch.randn_like(p) * (0.5 / p.shape[-1]**0.5 if p.dim() > 1 else 0.02))
    return attn


def _attach_lora(ref_edlora, attn, rank, alpha, seed):
    torch.manual_seed(seed)
    loras = {}
    for name, lin in (('to_q', attn.to_q), ('to_k', attn.to_k), ('to_v', attn.to_v), ('to_out.0', attn.to_out[0])):
        l = ref_edlora.LoRALinearLayer(name, lin, rank=rank, alpha=alpha)
        with torch.no_grad():
            l.lora_up.weight.copy_(torch.randn_like(l.lora_up.weight) * 0.05)  # non-zero: zero would hide the branch
        loras[name] = l
    return loras


def main(out_path=None):
    ref = _import_reference()
    from oracle.attention_shim import Attention as Shim
    out = {}
    torch.set_grad_enabled(True)

    # ---- G1: LoRALinearLayer on Linear and on 1x1 Conv ----------------------------------------
    torch.manual_seed(1)
    lin = torch.nn.Linear(48, 24, bias=True)
    l1 = ref['edlora'].LoRALinearLayer('lin', lin, rank=4, alpha=0.7)
    with torch.no_grad():
        l1.lora_up.weight.copy_(torch.randn_like(l1.lora_up.weight) * 0.1)
    x = torch.randn(3, 5, 48)
    conv = torch.nn.Conv2d(16, 8, 1)
    l2 = ref['edlora'].LoRALinearLayer('conv', conv, rank=4, alpha=1.0)
    with torch.no_grad():
        l2.lora_up.weight.copy_(torch.randn_like(l2.lora_up.weight) * 0.1)
    xc = torch.randn(2, 16, 6, 6)
    out['lora'] = dict(
        lin_w=lin.weight.detach(), lin_b=lin.bias.detach(), lin_down=l1.lora_down.weight.detach(),
        lin_up=l1.lora_up.weight.detach(), lin_alpha=0.7, x=x, y=lin(x).detach(),
        conv_w=conv.weight.detach(), conv_b=conv.bias.detach(), conv_down=l2.lora_down.weight.detach(),
        conv_up=l2.lora_up.weight.detach(), xc=xc, yc=conv(xc).detach(),
        default_up_is_zero=bool((ref['edlora'].LoRALinearLayer('z', torch.nn.Linear(8, 8)).lora_up.weight == 0).all()))

    # ---- G2: EDLoRA_AttnProcessor (cross with layer-wise states, and self) -----------------------------
    C, H, N, B = 64, 8, 64, 2  # golden vectors pin the CPU oracle; real SD shapes are covered oracle-vs-HIP
    attn = _seeded_attention(Shim, C, 64, H, seed=2)
    loras = _attach_lora(ref['edlora'], attn, 4, 1.0, seed=3)
    torch.manual_seed(4)
    hs = torch.randn(B, N, C)
    ehs = torch.randn(B, 6, 77, 64)  # 6 layers x 64-dim text states keep the fixture small
    proc = ref['edlora'].EDLoRA_AttnProcessor(5)
    y_cross = proc(attn, hs, encoder_hidden_states=ehs)
    attn_self = _seeded_attention(Shim, C, None, H, seed=5)
    y_self = ref['edlora'].EDLoRA_AttnProcessor(0)(attn_self, hs)
    out['edlora_attn'] = dict(
        state={k: v.detach() for k, v in attn.state_dict().items()},
        lora={k: dict(down=l.lora_down.weight.detach(), up=l.lora_up.weight.detach()) for k, l in loras.items()},
        self_state={k: v.detach() for k, v in attn_self.state_dict().items()},
        hs=hs, ehs=ehs, idx=5, y_cross=y_cross.detach(), y_self=y_self.detach())

    # ---- G3: control processor + AttentionStore(training) + cal_attn_reg, with gradients -----------------
    torch.manual_seed(6)
    store = ref['ptp'].AttentionStore(training=True)
    store.num_att_layers = 4
    res_list = (64, 32, 16, 8)
    attns, places, hss = [], ('down', 'down', 'mid', 'up'), []
    ehs3 = torch.randn(B, 4, 77, 32)
    outs3 = []
    for i, res in enumerate(res_list):
        Cc = 8  # tiny width keeps the fixture small; heads=2 -> d=4 (oracle-only case)
        a = _seeded_attention(Shim, Cc, 32, 2, seed=10 + i)
        attns.append(a)
        h = torch.randn(B, res * res, Cc).requires_grad_(True)
        hss.append(h)
        p = ref['edlora'].EDLoRA_Control_AttnProcessor(i, places[i], store)
        outs3.append(p(a, h, encoder_hidden_states=ehs3))
    masks = torch.zeros(B, 1, 64, 64)
    masks[:, :, 16:48, 20:44] = 1
    ids = torch.full((B * 16, 77), 49407, dtype=torch.long)
    ids[:, 0] = 49406
    concept_ids = list(range(49408, 49408 + 32))
    for b in range(B):
        for layer in range(16):
            ids[b * 16 + layer, 1:4] = torch.tensor([320, 1125, 539])
            ids[b * 16 + layer, 4 + b] = 49408 + layer       # adjective token (position differs per sample)
            ids[b * 16 + layer, 5 + b] = 49424 + layer       # subject token
    fake_self = types.SimpleNamespace(attn_reg_weight=0.01, reg_full_identity=False,
                                      get_all_concept_token_ids=lambda: concept_ids)
    maps = store.get_average_attention()
    reg_false = ref['trainer'].EDLoRATrainer.cal_attn_reg(fake_self, maps, masks, ids)
    fake_self.reg_full_identity = True
    reg_true = ref['trainer'].EDLoRATrainer.cal_attn_reg(fake_self, maps, masks, ids)
    total = reg_false + sum(o.square().mean() for o in outs3)
    grads = torch.autograd.grad(total, hss)
    out['control'] = dict(
        states=[{k: v.detach() for k, v in a.state_dict().items()} for a in attns], places=places,
        hs=[h.detach() for h in hss], ehs=ehs3, masks=masks, ids=ids, concept_ids=concept_ids,
        outs=[o.detach() for o in outs3], reg_false=reg_false.detach(), reg_true=reg_true.detach(),
        grads=[g.detach() for g in grads],
        n_stored={k: len(v) for k, v in maps.items()})

    # ---- G4: RegionT2I_AttnProcessor --------------------------------------------------------------
    torch.manual_seed(7)
    attn_r = _seeded_attention(Shim, C, 64, H, seed=8)
    fh, fw = 8, 12
    hs_r = torch.randn(2, fh * fw, C)
    ctx = torch.randn(2, 4, 77, 64)
    height, width = 512, 768
    px_boxes = [[2, 2, 512, 184], [7, 184, 512, 345], [1, 488, 512, 747], [100, 150, 400, 300]]  # last overlaps 1 & 2
    region_list = []
    for bx in px_boxes:
        frac = [bx[0] / height, bx[1] / width, bx[2] / height, bx[3] / width]
        region_list.append((torch.randn(2, 4, 77, 64), frac))
    procr = ref['region'].RegionT2I_AttnProcessor(3)
    kw = dict(region_list=region_list, height=height, width=width)
    y_reg = procr(attn_r, hs_r, encoder_hidden_states=ctx, **kw)
    y_reg_none = procr(attn_r, hs_r, encoder_hidden_states=ctx, region_list=[], height=height, width=width)
    attn_rs = _seeded_attention(Shim, C, None, H, seed=9)
    y_reg_self = procr(attn_rs, hs_r, **kw)
    out['region'] = dict(
        state={k: v.detach() for k, v in attn_r.state_dict().items()},
        self_state={k: v.detach() for k, v in attn_rs.state_dict().items()},
        hs=hs_r, ctx=ctx, regions=[(r[0], r[1]) for r in region_list], height=height, width=width, idx=3,
        fh=fh, fw=fw, y=y_reg.detach(), y_none=y_reg_none.detach(), y_self=y_reg_self.detach())
    prompt_rewrite = ('[a <potter1> <potter2>, in Hogwarts uniform]-*-[lowres]-*-[4, 6, 1024, 490]|'
                      '[a <hermione1> <hermione2>, girl]-*-[bad hands]-*-[14, 490, 1024, 920]|[x]-*-[]-*-[]')
    out['prepare_text'] = dict(arg=prompt_rewrite, height=1024, width=2048,
                               result=ref['region_cli'].prepare_text('three people', prompt_rewrite, 1024, 2048))

    # ---- G5: update_quasi_newton ---------------------------------------------------------------------
    cases = {}
    for name, (n, cin, cout, iters, seed) in dict(under=(12, 48, 24, 60, 20), over=(300, 32, 16, 40, 21),
                                                  spatial=(6000, 24, 24, 15, 22)).items():
        torch.manual_seed(seed)
        X = torch.randn(n, cin)
        W0 = torch.randn(cout, cin) * 0.1
        Wt = W0 + torch.randn(cout, cin) * 0.02
        Y = X @ Wt.T + 0.01 * torch.randn(n, cout)
        Wn = ref['fusion'].update_quasi_newton(X, Y, W0.clone(), iters, 'cpu')
        cases[name] = dict(X=X, Y=Y, W0=W0, iters=iters, W=Wn.detach(),
                           loss0=ref['fusion'].chunk_compute_mse(X, Y, W0, 'cpu').item(),
                           loss=ref['fusion'].chunk_compute_mse(X, Y, Wn, 'cpu').item())
    torch.manual_seed(23)
    X4 = torch.randn(4, 16, 6, 6)
    W40 = torch.randn(8, 16, 1, 1) * 0.1
    Y4 = torch.nn.functional.conv2d(X4, W40 + 0.02 * torch.randn_like(W40))
    W4 = ref['fusion'].update_quasi_newton(X4, Y4, W40.clone(), 30, 'cpu')
    cases['conv'] = dict(X=X4, Y=Y4, W0=W40, iters=30, W=W4.detach())
    out['lbfgs'] = cases

    # ---- G6: small host-side pieces ---------------------------------------------------------------------
    cfg = {'<potter1>': {'concept_token_names': [f'<new{i}>' for i in range(16)]},
           '<potter2>': {'concept_token_names': [f'<new{16 + i}>' for i in range(16)]}}
    prompts = ['a <potter1> <potter2> in the park', 'photo of a cat']
    out['bind'] = dict(cfg=cfg, prompts=prompts, result=ref['pipe'].bind_concept_prompt(prompts, cfg))
    torch.manual_seed(30)
    sd = {'blk.attn1.to_q.weight': torch.randn(8, 8), 'blk.attn2.to_out.0.weight': torch.randn(8, 8),
          'blk.proj_in.weight': torch.randn(8, 8, 1, 1), 'blk.norm.weight': torch.randn(8)}
    lora_sd = {}
    for k in ('blk.attn1.to_q', 'blk.attn2.to_out.0'):
        lora_sd[k + '.lora_down.weight'] = torch.randn(4, 8)
        lora_sd[k + '.lora_up.weight'] = torch.randn(8, 4)
    lora_sd['blk.proj_in.lora_down.weight'] = torch.randn(4, 8, 1, 1)
    lora_sd['blk.proj_in.lora_up.weight'] = torch.randn(8, 4, 1, 1)
    merged = ref['convert'].merge_lora_into_weight(sd, lora_sd, model_type='unet', alpha=0.6)
    te_sd = {'text_model.encoder.layers.0.self_attn.q_proj.weight': torch.randn(8, 8),
             'text_model.encoder.layers.0.mlp.fc1.weight': torch.randn(16, 8)}
    te_lora = {'text_model.encoder.layers.0.self_attn.q_proj.lora_down.weight': torch.randn(4, 8),
               'text_model.encoder.layers.0.self_attn.q_proj.lora_up.weight': torch.randn(8, 4)}
    merged_te = ref['fusion'].merge_lora_into_weight(te_sd, te_lora, list(te_sd.keys()), model_type='text_encoder',
                                                     alpha=0.8, device='cpu')
    out['merge'] = dict(sd=sd, lora=lora_sd, alpha=0.6, merged=merged, te_sd=te_sd, te_lora=te_lora, te_alpha=0.8,
                        merged_te=merged_te)

    out['adapter'] = adapter_region_weight_golden()
    out['control_eval'] = control_eval_golden(ref, Shim)
    path = out_path or os.path.join(HERE, 'reference_golden.pt')
    torch.save(out, path)
    print(f'wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)')


def control_eval_golden(ref, Shim):
    """G12: the controller boundary with the FULL probability map, in eval mode (VERDICT r04 missing #2). The reference's own
    EDLoRA_Control_AttnProcessor (mixofshow/models/edlora.py:22-100) hands (B*H, N, 77) to the reference's own controllers
    (mixofshow/utils/ptp_util.py:22-108): `AttentionStore(training=False)` -- AttentionControl.__call__ forwards only the second
    half of the CFG batch (:45-46) and writes it back in place, AttentionStore.forward stores it (:79-82), between_steps /
    get_average_attention accumulate over two 'steps' -- and an EDITING controller built on the reference's AttentionControl base
    (a prompt-to-prompt style re-weighting of two token columns with renormalisation), whose edit must reach the layer output."""
    torch.manual_seed(40)
    C, H, N, B, L = 64, 8, 48, 4, 3
    attns = [_seeded_attention(Shim, C, 32, H, seed=41 + i) for i in range(L)]
    places = ('down', 'mid', 'up')
    ehs = torch.randn(B, L, 77, 32)
    hs = [[torch.randn(B, N, C) for _ in range(L)] for _ in range(2)]          # two sampling 'steps'

    store = ref['ptp'].AttentionStore(training=False)
    store.num_att_layers = L
    procs = [ref['edlora'].EDLoRA_Control_AttnProcessor(i, places[i], store) for i in range(L)]
    outs = [[procs[i](attns[i], hs[step][i], encoder_hidden_states=ehs).detach() for i in range(L)] for step in range(2)]
    avg = store.get_average_attention()

    class Reweight(ref['ptp'].AttentionControl):            # the reference's base class: its __call__ / bookkeeping run here
        def __init__(self, cols, gain):
            super().__init__(low_resource=False, training=False)
            self.cols, self.gain = cols, gain

        def forward(self, attn, is_cross, place_in_unet):
            attn = attn.clone()
            attn[:, :, self.cols] = attn[:, :, self.cols] * self.gain
            return attn / attn.sum(-1, keepdim=True)

    edit = Reweight([4, 5], 3.0)
    edit.num_att_layers = L
    procs_e = [ref['edlora'].EDLoRA_Control_AttnProcessor(i, places[i], edit) for i in range(L)]
    outs_e = [procs_e[i](attns[i], hs[0][i], encoder_hidden_states=ehs).detach() for i in range(L)]
    return dict(states=[{k: v.detach() for k, v in a.state_dict().items()} for a in attns], places=places, ehs=ehs, hs=hs,
                outs=outs, cur_step=store.cur_step, avg={k: [t.detach() for t in v] for k, v in avg.items()},
                edit_cols=[4, 5], edit_gain=3.0, outs_edit=outs_e, edit_cur_step=edit.cur_step)


def adapter_region_weight_golden():
    """G7: the region-weighted T2I-Adapter feature rule. The reference has it inline in the pipeline's __call__
    (mixofshow/pipelines/pipeline_regionally_t2iadapter.py:484-542: `num_states = ...` up to `adapter_state.append`), so
    the golden is produced by executing exactly those source lines, dedented, on seeded stand-ins for the adapter outputs."""
    import math
    import textwrap
    src = open(os.path.join(REF, 'mixofshow/pipelines/pipeline_regionally_t2iadapter.py')).read().splitlines()
    first = next(i for i, l in enumerate(src) if l.strip().startswith('num_states = len(keypose_adapter_state)'))
    last = next(i for i, l in enumerate(src) if i > first and l.strip() == 'adapter_state.append(feat_keypose + feat_sketch)')
    code = textwrap.dedent('\n'.join(src[first:last + 1]))
    cases = {}
    height, width = 512, 768
    g = torch.Generator().manual_seed(40)
    shapes = [(1, 8, 64, 96), (1, 8, 32, 48), (1, 8, 16, 24), (1, 8, 8, 12)]
    kp = [torch.randn(s, generator=g) for s in shapes]
    sk = [torch.randn(s, generator=g) for s in shapes]
    specs = dict(
        keypose_only=dict(kp=True, sk=False, kw=1.0, sw=1.0, rk='', rs=''),
        both=dict(kp=True, sk=True, kw=0.8, sw=0.5, rk='', rs=''),
        region_keypose=dict(kp=True, sk=False, kw=1.0, sw=1.0, rk='[2, 2, 512, 184]-0.0|[7, 184, 512, 345]-0.5', rs=''),
        region_both=dict(kp=True, sk=True, kw=1.0, sw=0.7, rk='[1, 488, 512, 747]-0.25', rs='[100, 150, 400, 300]-1.5|[0, 0, 511, 767]-0.1'),
        sketch_only=dict(kp=False, sk=True, kw=1.0, sw=0.9, rk='', rs='[33, 17, 301, 500]-0.0'),
    )
    for name, c in specs.items():
        ns = dict(math=math, torch=torch, height=height, width=width,
                  keypose_adapter_state=[t.clone() for t in kp] if c['kp'] else None,
                  sketch_adapter_state=[t.clone() for t in sk] if c['sk'] else None,
                  keypose_adaptor_weight=c['kw'], sketch_adaptor_weight=c['sw'],
                  region_keypose_adaptor_weight=c['rk'], region_sketch_adaptor_weight=c['rs'])
        exec(code, ns)                                      # the reference's own lines
        cases[name] = dict(spec=c, out=[t.clone() for t in ns['adapter_state']])
    return dict(height=height, width=width, keypose=kp, sketch=sk, cases=cases, source_lines=(first + 1, last + 1))


# =====================================================================================================================
# G8: gradient fusion feature collection — the reference's OWN merge_text_encoder / merge_kv_in_cross_attention /
# merge_spatial_attention (gradient_fusion.py:325-457, 460-576, 627-747, with get_hooker :150-167, get_text_feature
# :183-214, decode_to_latents :579-624) executed on duck-typed local modules: the product's 'tiny768' tokenizer / CLIP /
# UNet / DPM-Solver classes. Both packages are called `mixofshow`, so the product package is imported a second time under
# the alias `mosproduct` (source text rewritten on load: every `mixofshow` -> `mosproduct`); the reference keeps the real
# name. What the reference hands to update_quasi_newton (X, Y, W0 per layer) and what it returns are recorded per layer.
# =====================================================================================================================
ALIAS = 'mosproduct'
PRESET = 'tiny768'      # 768-wide text tower: the reference hard-codes reshape(-1, 768) (:202)


class _AliasLoader(importlib.machinery.SourceFileLoader):

    def get_code(self, fullname):
        path = self.get_filename(fullname)
        with open(path, 'rb') as f:
            src = f.read().decode().replace('mixofshow', ALIAS).replace('import mos_path', 'pass')
        return compile(src, path, 'exec', dont_inherit=True)


class _AliasFinder(importlib.abc.MetaPathFinder):
    """mosproduct[.a.b] -> mix-of-show_amd/mixofshow/a/b ; mosproduct_root.<script> -> <repo>/<script>.py"""

    def find_spec(self, fullname, path, target=None):
        import importlib.util
        parts = fullname.split('.')
        if parts[0] == ALIAS:
            base = os.path.join(REPO, 'mix-of-show_amd', 'mixofshow', *parts[1:])
        elif parts[0] == ALIAS + '_root':
            base = os.path.join(REPO, *parts[1:]) if len(parts) > 1 else REPO
        else:
            return None
        if os.path.isdir(base):
            init = os.path.join(base, '__init__.py')
            if os.path.exists(init):
                return importlib.util.spec_from_file_location(fullname, init, loader=_AliasLoader(fullname, init),
                                                              submodule_search_locations=[base])
            return importlib.machinery.ModuleSpec(fullname, None, is_package=True)     # namespace (repo root)
        if os.path.exists(base + '.py'):
            return importlib.util.spec_from_file_location(fullname, base + '.py', loader=_AliasLoader(fullname, base + '.py'))
        return None


def fusion_golden(out_path=None, iters_te=30, iters_unet=12):
    import importlib
    import logging
    import pathlib
    import re
    import tempfile
    ref = _import_reference()
    rf = ref['fusion']
    sys.dont_write_bytecode = True
    sys.meta_path.insert(0, _AliasFinder())
    os.environ['MOS_TEST_ALLOW_CPU'] = '1'
    from oracle import emu_ops
    pops = importlib.import_module(ALIAS + '.hip.ops')
    for name in emu_ops.EMULATED:                       # same CPU stand-ins for the kernel library as tests/conftest.py
        setattr(pops, name, getattr(emu_ops, name))
    pgf = importlib.import_module(ALIAS + '_root.gradient_fusion')
    pbench = importlib.import_module(ALIAS + '_root.bench')
    spec = importlib.util.spec_from_file_location('fusion_fixture', os.path.join(HERE, 'fusion_fixture.py'))
    fx = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fx)
    dev = torch.device('cpu')
    out = dict(preset=PRESET, n_concepts=2, iters_te=iters_te, iters_unet=iters_unet, seed_spatial=77, stages={})
    with tempfile.TemporaryDirectory() as td:
        cfg = fx.make_fusion_fixture(pathlib.Path(td), PRESET, 2, build_trainer=pbench.build_trainer)
        pipe, _, sched = pgf.init_stable_diffusion(f'synthetic://{PRESET}?seed=0', dev)
        for p_ in list(pipe.text_encoder.parameters()) + list(pipe.unet.parameters()):
            p_.requires_grad = False
        # the layers the reference leaves on diffusers' default AttnProcessor (every attn1: revise_edlora_unet_attention_forward
        # only replaces attn2, edlora.py:176-190) get the restatement of that default processor (oracle/edlora_ref.py), so
        # that to_q / to_k / to_v / to_out run as module calls and the reference's hooks fire as they do under diffusers
        from oracle.edlora_ref import PlainAttnProcessorRef
        for m_ in pipe.unet.modules():
            if m_.__class__.__name__ == 'Attention':
                m_.set_processor(PlainAttnProcessorRef())
        emb, te, kv, sp, concepts = rf.parse_new_concepts(cfg)                                   # reference :269-322
        _, new_cfg = rf.merge_new_concepts_(emb, concepts, pipe.tokenizer, pipe.text_encoder)    # reference :217-266
        out['new_concept_cfg'] = new_cfg
        current = {}
        calls = []
        real_uqn, real_info = rf.update_quasi_newton, logging.info

        def info(msg, *a, **k):
            m = re.search(r'optimizing (\S+)', str(msg))
            if m:
                current['name'] = m.group(1)

        def spy(K_target, V_target, W, iters, device):
            W0 = W.detach().clone()                 # (the reference's L-BFGS updates the tensor it is given in place)
            Wn = real_uqn(K_target, V_target, W, iters, device)
            calls.append(dict(name=current['name'], X=K_target.detach().clone(), Y=V_target.detach().clone(), W0=W0,
                              W=Wn.detach().clone(), loss0=rf.chunk_compute_mse(K_target, V_target, W0, device).item()
                              if W0.dim() == 2 else None,
                              loss=rf.chunk_compute_mse(K_target, V_target, Wn, device).item() if W0.dim() == 2 else None))
            return Wn

        rf.update_quasi_newton, logging.info = spy, info
        try:
            def compact(t):      # fp16 storage where it is lossless (features recorded from the fp16 models)
                h = t.half()
                return h if torch.equal(h.float(), t.float()) else t

            def record(stage):
                """Per layer: n and the Gram statistics G = X^T X, P = Y^T X, c = sum Y^2 (fp64) of what the reference handed to
                update_quasi_newton; the raw X / Y where they are small (768-wide layers: few rows) INSTEAD of G / P; the
                reference's loss before / after its L-BFGS; its fused weight where that is small."""
                layers, seen = {}, []
                for c in sorted(calls, key=lambda c_: c_['name']):     # (the reference iterates a set: order varies per run)
                    X, Y = c['X'], c['Y']
                    if X.dim() == 4:
                        X2, Y2 = X.permute(0, 2, 3, 1).reshape(-1, X.shape[1]).double(), Y.permute(0, 2, 3, 1).reshape(-1, Y.shape[1]).double()
                    else:
                        X2, Y2 = X.reshape(-1, X.shape[-1]).double(), Y.reshape(-1, Y.shape[-1]).double()
                    ent = dict(n=X2.shape[0], c=(Y2 * Y2).sum(), x_shape=tuple(X.shape), y_shape=tuple(Y.shape),
                               loss0=c['loss0'], loss=c['loss'], dW_norm=(c['W'] - c['W0']).norm().item(),
                               W0_norm=c['W0'].norm().item())
                    if X.numel() + Y.numel() <= 1_000_000:
                        same = next((nm for nm, t in seen if t.shape == X.shape and torch.equal(t, X)), None)
                        ent.update(X=('same_as', same) if same else compact(X), Y=compact(Y))
                        if same is None:
                            seen.append((c['name'], X))
                    else:
                        ent.update(G=X2.T @ X2, P=Y2.T @ X2)
                    if c['W'].numel() <= 65536:
                        ent.update(W=c['W'])
                    layers[c['name']] = ent
                out['stages'][stage] = layers
                calls.clear()

            # every stage starts from the PRETRAINED weights (compose_concepts chains them, :766-797; unchained, a stage's
            # features do not depend on the previous stage's solver, so each pins the feature collection on its own)
            te0 = {k: v.detach().clone() for k, v in pipe.text_encoder.state_dict().items()}
            u0 = {k: v.detach().clone() for k, v in pipe.unet.state_dict().items()}
            rf.merge_text_encoder(concepts, iters_te, new_cfg, pipe.tokenizer, pipe.text_encoder, te, dev)
            record('text_encoder')
            pipe.text_encoder.load_state_dict(te0)      # (the reference leaves the LAST concept's merged weights loaded)
            rf.merge_kv_in_cross_attention(concepts, iters_te, new_cfg, pipe.tokenizer, pipe.text_encoder, pipe.unet, kv, dev)
            record('cross_kv')
            pipe.unet.load_state_dict(u0)
            torch.manual_seed(77)                   # decode_to_latents draws the latents from the global generator (:601)
            rf.merge_spatial_attention(concepts, iters_unet, new_cfg, pipe.tokenizer, pipe.text_encoder, pipe.unet, sp,
                                       sched, dev)
            record('spatial')
        finally:
            rf.update_quasi_newton, logging.info = real_uqn, real_info
    path = out_path or os.path.join(HERE, 'reference_fusion_golden.pt')
    torch.save(out, path)
    n = {k: len(v) for k, v in out['stages'].items()}
    print(f'wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB); layers per stage: {n}')


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'fusion':
        fusion_golden(sys.argv[2] if len(sys.argv) > 2 else None)
    elif len(sys.argv) > 1 and sys.argv[1] == 'control_eval':  # add G12 to the existing fixture without touching the rest
        path = os.path.join(HERE, 'reference_golden.pt')
        out = torch.load(path, weights_only=False)
        ref = _import_reference()
        from oracle.attention_shim import Attention as Shim
        out['control_eval'] = control_eval_golden(ref, Shim)
        torch.save(out, path)
        print('added control_eval golden')
    elif len(sys.argv) > 1 and sys.argv[1] == 'adapter':      # add G7 to the existing fixture without touching the rest
        path = os.path.join(HERE, 'reference_golden.pt')
        out = torch.load(path, weights_only=False)
        out['adapter'] = adapter_region_weight_golden()
        torch.save(out, path)
        print('added adapter golden; reference source lines', out['adapter']['source_lines'])
    else:
        main(sys.argv[1] if len(sys.argv) > 1 else None)      # optional output path (tests regenerate into tmp_path)
