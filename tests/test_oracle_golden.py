"""The oracle (oracle/*.py) against golden vectors recorded from the REAL reference code
(tests/golden/make_golden.py). CPU only."""
import types

import torch

from oracle import edlora_ref as R
from oracle import fusion_ref, region_ref
from oracle.attention_shim import Attention as Shim


def _attn_from_state(state, C, cross, heads):
    a = Shim(C, cross_attention_dim=cross, heads=heads, dim_head=C // heads)
    a.load_state_dict(state)
    return a


def test_lora_linear_layer(golden):
    g = golden['lora']
    assert g['default_up_is_zero']
    lin = torch.nn.Linear(48, 24)
    lin.load_state_dict({'weight': g['lin_w'], 'bias': g['lin_b']})
    l = R.LoRALinearLayerRef('lin', lin, rank=4, alpha=g['lin_alpha'])
    l.lora_down.weight.data.copy_(g['lin_down']); l.lora_up.weight.data.copy_(g['lin_up'])
    torch.testing.assert_close(lin(g['x']), g['y'], rtol=1e-6, atol=1e-6)
    y2 = R.lora_linear_ref(g['x'], g['lin_w'], g['lin_b'], g['lin_down'], g['lin_up'], g['lin_alpha'])
    torch.testing.assert_close(y2, g['y'], rtol=1e-6, atol=1e-6)
    conv = torch.nn.Conv2d(16, 8, 1)
    conv.load_state_dict({'weight': g['conv_w'], 'bias': g['conv_b']})
    l2 = R.LoRALinearLayerRef('conv', conv, rank=4, alpha=1.0)
    l2.lora_down.weight.data.copy_(g['conv_down']); l2.lora_up.weight.data.copy_(g['conv_up'])
    torch.testing.assert_close(conv(g['xc']), g['yc'], rtol=1e-6, atol=1e-6)
    # known answer: default init (up = 0) leaves the wrapped layer unchanged (edlora.py:239)
    lin3 = torch.nn.Linear(8, 8)
    x3 = torch.randn(4, 8)
    y_before = lin3(x3)
    R.LoRALinearLayerRef('z', lin3)
    torch.testing.assert_close(lin3(x3), y_before)


def test_edlora_attn_processor(golden):
    g = golden['edlora_attn']
    C = g['hs'].shape[-1]
    attn = _attn_from_state(g['state'], C, g['ehs'].shape[-1], 8)
    for name, lin in (('to_q', attn.to_q), ('to_k', attn.to_k), ('to_v', attn.to_v), ('to_out.0', attn.to_out[0])):
        l = R.LoRALinearLayerRef(name, lin, rank=4, alpha=1.0)
        l.lora_down.weight.data.copy_(g['lora'][name]['down']); l.lora_up.weight.data.copy_(g['lora'][name]['up'])
    y = R.EDLoRA_AttnProcessorRef(g['idx'])(attn, g['hs'], encoder_hidden_states=g['ehs'])
    torch.testing.assert_close(y, g['y_cross'], rtol=1e-5, atol=1e-6)
    attn_s = _attn_from_state(g['self_state'], C, None, 8)
    y = R.EDLoRA_AttnProcessorRef(0)(attn_s, g['hs'])
    torch.testing.assert_close(y, g['y_self'], rtol=1e-5, atol=1e-6)


def test_control_processor_store_and_attn_reg(golden):
    g = golden['control']
    store = R.AttentionStoreRef(training=True)
    store.num_att_layers = 4
    hss = [h.clone().requires_grad_(True) for h in g['hs']]
    outs = []
    for i, (state, place) in enumerate(zip(g['states'], g['places'])):
        a = _attn_from_state(state, hss[i].shape[-1], g['ehs'].shape[-1], 2)
        outs.append(R.EDLoRA_Control_AttnProcessorRef(i, place, store)(a, hss[i], encoder_hidden_states=g['ehs']))
    for o, ref in zip(outs, g['outs']):
        torch.testing.assert_close(o, ref, rtol=1e-5, atol=1e-6)
    maps = store.get_average_attention()
    assert {k: len(v) for k, v in maps.items()} == g['n_stored']
    reg_f = R.cal_attn_reg_ref(maps, g['masks'], g['ids'], g['concept_ids'], 0.01, False)
    reg_t = R.cal_attn_reg_ref(maps, g['masks'], g['ids'], g['concept_ids'], 0.01, True)
    torch.testing.assert_close(reg_f, g['reg_false'], rtol=1e-5, atol=1e-8)
    torch.testing.assert_close(reg_t, g['reg_true'], rtol=1e-5, atol=1e-8)
    total = reg_f + sum(o.square().mean() for o in outs)
    grads = torch.autograd.grad(total, hss)
    for a, b in zip(grads, g['grads']):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-9)


class _ReweightCtl:
    """The editing controller of golden G12, duck-typed: the reference's AttentionControl.__call__ in eval mode edits the
    conditional half of the CFG batch in place (ptp_util.py:45-46)."""

    def __init__(self, cols, gain, n_layers):
        self.cols, self.gain, self.num_att_layers, self.cur_att_layer, self.cur_step = cols, gain, n_layers, 0, 0

    def __call__(self, attn, is_cross, place):
        h = attn.shape[0]
        e = attn[h // 2:].clone()
        e[:, :, self.cols] = e[:, :, self.cols] * self.gain
        attn[h // 2:] = e / e.sum(-1, keepdim=True)
        self.cur_att_layer += 1
        if self.cur_att_layer == self.num_att_layers:
            self.cur_att_layer, self.cur_step = 0, self.cur_step + 1
        return attn


def test_control_processor_eval_mode_full_maps(golden):
    """G12: the reference's control processor + its own AttentionStore(training=False) and an editing AttentionControl."""
    g = golden['control_eval']
    L = len(g['states'])
    C, cross = g['hs'][0][0].shape[-1], g['ehs'].shape[-1]
    attns = [_attn_from_state(st, C, cross, 8) for st in g['states']]
    store = R.AttentionStoreRef(training=False)
    store.num_att_layers = L
    procs = [R.EDLoRA_Control_AttnProcessorRef(i, g['places'][i], store) for i in range(L)]
    with torch.no_grad():
        for step in range(2):
            for i in range(L):
                y = procs[i](attns[i], g['hs'][step][i], encoder_hidden_states=g['ehs'])
                torch.testing.assert_close(y, g['outs'][step][i], rtol=1e-5, atol=1e-6)
        assert store.cur_step == g['cur_step'] == 2
        avg = store.get_average_attention()
        assert {k: len(v) for k, v in avg.items()} == {k: len(v) for k, v in g['avg'].items()}
        for k in avg:
            for a, b in zip(avg[k], g['avg'][k]):
                assert a.shape == b.shape                       # (B/2 * H, N, 77): the conditional half only
                torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-7)
        edit = _ReweightCtl(g['edit_cols'], g['edit_gain'], L)
        for i in range(L):
            y = R.EDLoRA_Control_AttnProcessorRef(i, g['places'][i], edit)(attns[i], g['hs'][0][i], encoder_hidden_states=g['ehs'])
            torch.testing.assert_close(y, g['outs_edit'][i], rtol=1e-5, atol=1e-6)
            assert not torch.allclose(y, g['outs'][0][i], atol=1e-4)         # the edit reaches the output
        assert edit.cur_step == g['edit_cur_step']


def test_region_processor(golden):
    g = golden['region']
    C = g['hs'].shape[-1]
    attn = _attn_from_state(g['state'], C, g['ctx'].shape[-1], 8)
    proc = region_ref.RegionT2I_AttnProcessorRef(g['idx'])
    kw = dict(region_list=g['regions'], height=g['height'], width=g['width'])
    torch.testing.assert_close(proc(attn, g['hs'], encoder_hidden_states=g['ctx'], **kw), g['y'], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(
        proc(attn, g['hs'], encoder_hidden_states=g['ctx'], region_list=[], height=g['height'], width=g['width']),
        g['y_none'], rtol=1e-5, atol=1e-6)
    attn_s = _attn_from_state(g['self_state'], C, None, 8)
    torch.testing.assert_close(proc(attn_s, g['hs'], **kw), g['y_self'], rtol=1e-5, atol=1e-6)
    # known answers: empty region list == plain cross attention; two identical regions == one region
    plain = R.EDLoRA_AttnProcessorRef(g['idx'])(attn, g['hs'], encoder_hidden_states=g['ctx'])
    torch.testing.assert_close(g['y_none'], plain, rtol=1e-5, atol=1e-6)
    r0 = g['regions'][0]
    one = proc(attn, g['hs'], encoder_hidden_states=g['ctx'], region_list=[r0], height=g['height'], width=g['width'])
    two = proc(attn, g['hs'], encoder_hidden_states=g['ctx'], region_list=[r0, r0], height=g['height'], width=g['width'])
    torch.testing.assert_close(one, two, rtol=1e-5, atol=1e-6)


def test_prepare_text(golden):
    g = golden['prepare_text']
    got = region_ref.prepare_text_ref('three people', g['arg'], g['height'], g['width'])
    assert got == g['result']


def test_update_quasi_newton(golden):
    for name, c in golden['lbfgs'].items():
        W = fusion_ref.update_quasi_newton_ref(c['X'], c['Y'], c['W0'].clone(), c['iters'])
        # same algorithm, same machine arithmetic -> same iterate
        torch.testing.assert_close(W, c['W'], rtol=1e-4, atol=1e-6, msg=lambda m: f'{name}: {m}')
        if 'loss' in c:
            l = fusion_ref.chunk_compute_mse_ref(c['X'], c['Y'], W).item()
            assert abs(l - c['loss']) <= 1e-6 * max(1.0, abs(c['loss'])) + 1e-9
            assert c['loss'] < c['loss0']


def test_bind_and_merge(golden):
    g = golden['bind']
    assert R.bind_concept_prompt_ref(g['prompts'], g['cfg']) == g['result']
    m = golden['merge']
    merged, n = R.merge_lora_into_weight_ref(m['sd'], m['lora'], 'unet', m['alpha'])
    assert n == 3
    for k, v in m['merged'].items():
        torch.testing.assert_close(merged[k], v, rtol=1e-6, atol=1e-6)
    merged_te, n = R.merge_lora_into_weight_ref(m['te_sd'], m['te_lora'], 'text_encoder', m['te_alpha'],
                                                layer_names=list(m['te_sd'].keys()))
    assert n == 1
    for k, v in m['merged_te'].items():
        torch.testing.assert_close(merged_te[k], v, rtol=1e-6, atol=1e-6)


def test_lsq_analytic_known_answers(emulated_hip):
    """SURVEY 8(c) known answers for the fusion solver, for the oracle restatement AND the product's Gram-form solver
    (kernels emulated): (1) n >= Cin, full rank => the normal-equation solution; (2) orthogonal sample rows, n < Cin =>
    W0 + sum_i (y_i - W0 x_i) x_i^T / |x_i|^2 (L-BFGS never leaves W0 + span of the samples)."""
    from mixofshow.utils import lsq
    torch.manual_seed(0)
    X, Y, W0 = torch.randn(200, 12), torch.randn(200, 6), torch.randn(6, 12) * 0.1
    W_ne = torch.linalg.solve(X.double().t() @ X.double(), X.double().t() @ Y.double()).t().float()
    for solver in (lambda: fusion_ref.update_quasi_newton_ref(X, Y, W0.clone(), 200),
                   lambda: lsq.update_quasi_newton(X, Y, W0.clone(), 200, torch.device('cpu'))):
        torch.testing.assert_close(solver(), W_ne, rtol=2e-3, atol=2e-4)
    Q, _ = torch.linalg.qr(torch.randn(16, 5))             # 5 orthonormal directions in R^16
    Xo = (Q.t() * torch.tensor([1.0, 2.0, 0.5, 3.0, 1.5])[:, None]).contiguous()     # rows x_i, mutually orthogonal
    Yo, W1 = torch.randn(5, 4), torch.randn(4, 16) * 0.1
    W_exp = W1 + sum(torch.outer(Yo[i] - W1 @ Xo[i], Xo[i]) / Xo[i].dot(Xo[i]) for i in range(5))
    for solver in (lambda: fusion_ref.update_quasi_newton_ref(Xo, Yo, W1.clone(), 200),
                   lambda: lsq.update_quasi_newton(Xo, Yo, W1.clone(), 200, torch.device('cpu'))):
        W = solver()
        torch.testing.assert_close(W, W_exp, rtol=2e-3, atol=2e-4)
        assert fusion_ref.lsq_loss_ref(Xo, Yo, W).item() < 1e-8
