"""The C-ABI boundary without a GPU: the shared library loads, exports every function include/mos_hip.h declares
(and the Python binding lists exactly those), argument validation fails loudly with a message, and the product
refuses CPU tensors / a missing library instead of falling back."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'mos_hip.h')


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(mos_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from mixofshow.hip import lib
    names = _declared_functions()
    assert len(names) >= 20 and 'mos_attn_fwd' in names and 'mos_region_cross_attn_fwd' in names
    assert sorted(lib.SIGNATURES.keys()) == names, 'python binding and header disagree'
    cdll = ctypes.CDLL(lib.LIB_PATH)
    for n in names:
        assert hasattr(cdll, n), f'{n} declared in mos_hip.h but not exported by libmos_hip.so'
    L = lib.load()
    assert L.mos_version() >= 100
    assert L.mos_last_error_string() is not None


def test_argument_validation_returns_status_and_message():
    from mixofshow.hip import lib
    L = lib.load()
    s = lib.AttnShape()
    s.B, s.H, s.Nq, s.Nkv, s.d = 1, 8, 64, 64, 48           # unsupported head dim
    s.q_bs = s.k_bs = s.v_bs = s.o_bs = 64 * 384
    s.q_rs = s.k_rs = s.v_rs = s.o_rs = 384
    s.scale = 1.0
    one = ctypes.c_void_p(16)                                 # never dereferenced: validation precedes launch
    rc = L.mos_attn_fwd(one, one, one, one, None, None, 0, None, ctypes.byref(s), lib.MOS_F16, None)
    assert rc == -2 and b'head dim' in L.mos_last_error_string()
    rc = L.mos_lora_linear_fwd(None, 0, None, 0, None, None, None, None, 0, 1, 8, 8, lib.MOS_F16, None)
    assert rc == -1 and b'NULL' in L.mos_last_error_string()
    rc = L.mos_lora_linear_fwd(one, 12, one, 8, None, None, None, one, 8, 4, 8, 8, lib.MOS_F16, None)   # ldx % 8
    assert rc == -1
    assert L.mos_attn_bwd_workspace_bytes(None) == 0
    assert L.mos_lora_bwd_workspace_bytes(16384, 960, 320) > 0
    assert L.mos_gram_workspace_bytes(100000, 320, 320) > 0
    assert L.mos_lsq_workspace_bytes(320, 320) == 25 * 8


def test_no_cpu_fallback():
    import mixofshow.hip.ops as ops
    from mixofshow.models.attention import Attention
    x = torch.randn(2, 64, 320)
    attn = Attention(320, heads=8, dim_head=40)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        attn(x.half())
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.gram_accumulate(x[0].half(), x[0].half(), torch.zeros(320, 320, dtype=torch.float64),
                            torch.zeros(320, 320, dtype=torch.float64), torch.zeros(1, dtype=torch.float64))


def test_missing_library_is_loud(monkeypatch, tmp_path):
    from mixofshow.hip import lib
    monkeypatch.setattr(lib, '_lib', None)
    monkeypatch.setattr(lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(lib.MosHipUnavailable, match='no CPU/PyTorch fallback'):
        lib.load()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'mix-of-show_amd')
    roots = [pkg] + [os.path.join(ROOT, f) for f in ('train_edlora.py', 'gradient_fusion.py',
                                                     'regionally_controlable_sampling.py', 'test_edlora.py')]
    bad = []
    for r in roots:
        files = [r] if os.path.isfile(r) else [os.path.join(d, f) for d, _, fs in os.walk(r) for f in fs if f.endswith('.py')]
        for f in files:
            if re.search(r'^\s*(from|import)\s+oracle\b', open(f).read(), flags=re.M):
                bad.append(f)
    assert not bad, f'product files import the oracle: {bad}'


def test_workspace_size_functions_host_side():
    """The `mos_*_workspace_bytes` entry points are pure host functions: callable without a GPU. They pin the launch
    plans (split counts) the kernels use: the caller allocates exactly this much."""
    import ctypes
    from mixofshow.hip import lib
    L = lib.load()

    def attn_ws(B, H, Nq, Nkv, d):
        s = lib.AttnShape(B, H, Nq, Nkv, d, 0, 0, 0, 0, 0, 0, 0, 0, 1.0)
        return L.mos_attn_bwd_workspace_bytes(ctypes.byref(s))

    def rows(B, H, Nq):
        return 4 * ((B * H * Nq + 3) // 4 * 4)

    # level-0 self attention of the bench: >= 512 key blocks already, no query splits -> only the D vector
    assert attn_ws(4, 8, 4096, 4096, 40) == rows(4, 8, 4096)
    # cross attention, 77 keys = one key block per (b, h): queries are split to fill the chip, fp32 partials for dK, dV
    ws = attn_ws(4, 8, 4096, 77, 40)
    extra = ws - rows(4, 8, 4096)
    per_split = 2 * 4 * 8 * 77 * 40 * 4
    assert extra > 0 and extra % per_split == 0 and 2 <= extra // per_split <= 64
    assert attn_ws(4, 8, 4096, 77, 40) <= attn_ws(4, 8, 8192, 77, 40)            # monotone in the number of queries
    assert L.mos_attn_bwd_workspace_bytes(None) == 0
    # LoRA backward: partial sums of dA / dB (at most 32 chunks... of 16 x max(N, K) floats) — positive, monotone
    w1, w2 = L.mos_lora_bwd_workspace_bytes(16384, 320, 320), L.mos_lora_bwd_workspace_bytes(16384, 960, 320)
    assert 0 < w1 <= w2 and w1 % 4 == 0
    # Gram accumulation: chunks x padded (Cout + Cin) x Cin floats; degenerate shapes -> 0
    g = L.mos_gram_workspace_bytes(20000, 320, 320)
    assert g > 0 and g % (2 * 384 * 320 * 4) == 0 or g % (320 * 4) == 0
    assert L.mos_gram_workspace_bytes(0, 320, 320) == 0
    assert L.mos_lsq_workspace_bytes(320, 768) == 5 * 12 * 8
    # GroupNorm: B x G x splits x 2 floats; invalid channel/group combination -> 0
    assert L.mos_groupnorm_workspace_bytes(4, 320, 4096, 32) > 0 and L.mos_groupnorm_workspace_bytes(4, 320, 4096, 33) == 0
