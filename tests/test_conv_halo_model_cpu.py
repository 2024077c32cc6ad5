"""The lane-level model of conv3x3_halo_kernel (tools/sim_conv_halo.py) on the CPU: the kernel's index arithmetic -- LDS-DMA
destinations, the chunk swizzle of both chunk widths, the pattern + immediate fragment reads, scalar-offset DMA sources, the
zero-record descriptor past the last chunk, ragged tiles, the upsampled halo fetch -- transcribed literally and compared with
conv2d. The device tests (tests/test_gpu_primitives.py::test_conv3x3_nhwc) check the kernel itself; this one keeps the model that
the kernel was designed against runnable without a GPU and tied to the source text it transcribes."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import sim_conv_halo  # noqa: E402


@pytest.mark.parametrize('B,H,W,C,N,TH,BN,CK,up', [
    (1, 9, 17, 64, 72, 8, 64, 64, False),        # default tile: ragged in both directions, Cout past the 64-wide tile
    (1, 17, 18, 64, 128, 16, 128, 32, False),    # the 16 x 16 x 128 tile on 32-channel chunks, ragged
    (1, 5, 9, 64, 64, 8, 64, 64, True),          # nearest-2x upsample folded into the halo fetch (10 x 18 output)
    (1, 8, 9, 32, 128, 16, 128, 32, True),       # ... on the 32-channel tile (16 x 18 output)
])
def test_halo_model_matches_conv2d(B, H, W, C, N, TH, BN, CK, up):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, H, W, C, generator=g, dtype=torch.float64)
    w = torch.randn(N, 3, 3, C, generator=g, dtype=torch.float64)
    xin = x.permute(0, 3, 1, 2)
    if up:
        xin = torch.nn.functional.interpolate(xin, scale_factor=2.0, mode='nearest')
    ref = torch.nn.functional.conv2d(xin, w.permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    got = torch.from_numpy(sim_conv_halo.run_halo(x.numpy(), w.numpy(), TH, BN, up=up, CK=CK))
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < 1e-9


def test_swizzles_are_conflict_free_for_ds_read_b128_lane_groups():
    """16 consecutive rows per k-group, any first row: every ds_read_b128 lane group (MI355X_MICROARCH.md, LDS) touches each of the
    sixteen 16-byte bank slots once -- for rows of 8 chunks with chunk ^ (row & 7) and rows of 4 chunks with chunk ^ ((row >> 1) & 3)."""
    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
              list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
    for CK in (64, 32):
        for r0 in range(0, 40):
            for kk in range(CK // 32):
                for grp in groups:
                    slots = set()
                    for lane in grp:
                        l15, lg = lane & 15, lane >> 4
                        row = r0 + l15
                        addr = row * CK * 2 + (((kk * 4 + lg) ^ sim_conv_halo.swz(CK, row)) * 16)
                        slots.add((addr // 16) % 16)
                    assert len(slots) == 16, (CK, r0, kk)


def test_model_transcribes_the_kernel_source():
    """The expressions the model copies are the ones in mos_conv.hip (a changed kernel must change the model with it)."""
    src = open(os.path.join(ROOT, 'mix-of-show_amd', 'csrc', 'mos_conv.hip')).read()
    for needle in ('return CK == 64 ? (row & 7) : ((row >> 1) & 3);',
                   'const int hr = (wave + 4 * i) * RPP + lane / CPR;',
                   'const int lc = (lane % CPR) ^ halo_swz<CK>(hr);',
                   'hpat[c][kk] = (wm * MI * HW18 + l15) * CK + (((kk * 4 + lg) ^ halo_swz<CK>(l15 + c)) * 8);',
                   'const int cst = (i + tap / 3) * HW18 + tap % 3;',
                   'bfrag[i] = as_v8<T>(ld16(hs + hpat[cst & 7][kk] + cst * CK));',
                   'constexpr int OOB = (int)0x80000000u;'):
        assert needle in src, needle
