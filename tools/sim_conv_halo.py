"""Lane-level model of the data movement of conv3x3_halo_kernel (csrc/mos_conv.hip; the default 3x3 path since round 5): LDS as an element array, dma16 = 16 B per lane at
(wave-uniform base) + lane * 16 from a global byte offset (zeros out of range), ld16 = 8 elements at an LDS element address, the
16x16x32 MFMA by its lane layout. The kernel's index arithmetic is transcribed literally; the sum is compared with conv2d.
The same model with the raster kernel's arithmetic (known to be right on the device) validates the model itself."""
import numpy as np
import torch

CBK = 64


def mfma16(acc, afrag, bfrag):
    """acc[lane][r] (D row = (lane >> 4) * 4 + r, col = lane & 15) += sum_k A[row][k] B[k][col];
    afrag[lane] = A[row = lane & 15][k = (lane >> 4) * 8 + e], bfrag[lane] = B[k = (lane >> 4) * 8 + e][col = lane & 15]."""
    A = np.zeros((16, 32)); Bm = np.zeros((32, 16))
    for lane in range(64):
        A[lane & 15, (lane >> 4) * 8:(lane >> 4) * 8 + 8] = afrag[lane]
        Bm[(lane >> 4) * 8:(lane >> 4) * 8 + 8, lane & 15] = bfrag[lane]
    D = A @ Bm
    for lane in range(64):
        for r in range(4):
            acc[lane][r] += D[(lane >> 4) * 4 + r, lane & 15]


def swz(CK, row):
    return (row & 7) if CK == 64 else ((row >> 1) & 3)


def run_halo(x, w, TH, BN, up=False, CK=64):
    """x (B, H, W, C) NHWC, w (Cout, 3, 3, C). Returns y (B, H, W, Cout) computed tile by tile like the kernel.
    up: x is (B, H/2, W/2, C) and is read through a nearest 2x upsample (template parameter UP of the kernel).
    CK: channels per chunk (template parameter CK: 64 -> rows of 8 chunks, 32 -> rows of 4 chunks, 16 rows per DMA piece)."""
    B, H, Wd, C = x.shape
    Hsrc, Wsrc = H, Wd
    if up:
        H, Wd = 2 * H, 2 * Wd
    N = w.shape[0]
    K = 9 * C
    MI, NJ, WCH = TH // 2, BN // 32, BN * CK // 2048
    RPP, CPR, KK = 512 // CK, CK // 8, CK // 32
    HW18 = 18
    HR = (TH + 2) * HW18
    NP = (HR + RPP - 1) // RPP
    NPW = (NP + 3) // 4
    HB = NPW * 4 * 512
    assert (MI * HW18) % 8 == 0 and WCH >= 1
    xg = x.reshape(-1).astype(np.float64)           # element-addressed global memory (offsets below are in BYTES, 2 per element)
    wg = w.reshape(-1).astype(np.float64)
    y = np.zeros((B, H, Wd, N))
    tx_n, ty_n = (Wd + 15) // 16, (H + TH - 1) // TH
    mt, nt = B * tx_n * ty_n, (N + BN - 1) // BN
    cpt = C // CK

    def gload(mem, lane_off, uni_off, limit_bytes):
        """buffer load of 16 B at lane_off (VGPR) + uni_off (scalar offset); out of range (either reading of the hardware rule:
        with or without the scalar part in the check) -> zeros. Asserts that both readings agree for every access issued."""
        a_ = lane_off >= limit_bytes                       # scalar offset not part of the check
        b_ = lane_off >= limit_bytes - uni_off             # ... or subtracted from the record count
        assert a_ == b_, (lane_off, uni_off, limit_bytes)
        if a_:
            return np.zeros(8)
        o = lane_off + uni_off
        assert 0 <= o and o + 16 <= limit_bytes
        return mem[o // 2: o // 2 + 8]

    for m_tile in range(mt):
        for n_tile in range(nt):
            b, tr = divmod(m_tile, tx_n * ty_n)
            y0, x0 = (tr // tx_n) * TH, (tr - (tr // tx_n) * tx_n) * 16
            n0 = n_tile * BN
            lds_h = np.zeros(2 * HB)
            lds_w = np.zeros(3 * BN * CK)
            acc = {(wave, j, i): [[0.0] * 4 for _ in range(64)] for wave in range(4) for j in range(NJ) for i in range(MI)}
            OOB = 0x80000000

            def hoff(wave, lane, i):
                hr = (wave + 4 * i) * RPP + lane // CPR
                hy, hx = divmod(hr, HW18)
                yy, xx = y0 + hy - 1, x0 + hx - 1
                lc = (lane % CPR) ^ swz(CK, hr)
                ok = hr < HR and 0 <= yy < H and 0 <= xx < Wd
                ys, xs = (yy >> 1, xx >> 1) if up else (yy, xx)
                return (((b * Hsrc + ys) * Wsrc + xs) * C + lc * 8) * 2 if ok else OOB

            def issue_halo(cch, hb):
                live = cch < cpt
                limit = xg.size * 2 if live else 0          # past the last chunk: the descriptor of zero records
                cb = cch * CK * 2 if live else 0
                for wave in range(4):
                    for i in range(NPW):
                        base = hb * HB + (wave + 4 * i) * 512
                        for lane in range(64):
                            lds_h[base + lane * 8: base + lane * 8 + 8] = gload(xg, hoff(wave, lane, i), cb, limit)

            def issue_w(cch, tap, buf):
                live = cch < cpt
                limit = ((N - 1) * K + K) * 2 if live else 0
                kb = (tap * C + cch * CK) * 2 if live else 0
                for wave in range(4):
                    for i in range(WCH):
                        base = buf * BN * CK + wave * 512 + i * 2048
                        for lane in range(64):
                            q = wave * 64 + lane + 256 * i
                            row = q // CPR
                            wo = ((n0 + row) * K + ((q % CPR) ^ swz(CK, row)) * 8) * 2
                            lds_w[base + lane * 8: base + lane * 8 + 8] = gload(wg, wo, kb, limit)

            issue_halo(0, 0)
            for cch in range(cpt):
                issue_halo(cch + 1, (cch + 1) & 1)         # (timing is not modelled: buffers are distinct, order is irrelevant here;
                hsb = (cch & 1) * HB                       #  the call past the last chunk exercises the zero-record descriptor)
                for tap in range(9):
                    issue_w(cch, tap, tap % 3)
                    if cch == cpt - 1 and tap == 8:
                        issue_w(cpt, 0, 0)                 # a weight tile past the end: zeros through the dead descriptor
                    for wave in range(4):
                        wm, wn = wave >> 1, wave & 1
                        for kk in range(KK):
                            bfr = {i: [None] * 64 for i in range(MI)}
                            afr = {j: [None] * 64 for j in range(NJ)}
                            for lane in range(64):
                                l15, lg = lane & 15, lane >> 4
                                for i in range(MI):
                                    cst = (i + tap // 3) * HW18 + tap % 3
                                    hpat = (wm * MI * HW18 + l15) * CK + (((kk * 4 + lg) ^ swz(CK, l15 + (cst & 7))) * 8)   # hpat[cst & 7][kk]
                                    a0 = hsb + hpat + cst * CK
                                    bfr[i][lane] = lds_h[a0:a0 + 8].copy()
                                wsw = swz(CK, l15)
                                for j in range(NJ):
                                    a0 = (tap % 3) * BN * CK + (wn * (BN // 2) + l15) * CK + j * 16 * CK + ((kk * 4 + lg) ^ wsw) * 8
                                    afr[j][lane] = lds_w[a0:a0 + 8].copy()
                            for j in range(NJ):
                                for i in range(MI):
                                    mfma16(acc[(wave, j, i)], afr[j], bfr[i])
            for wave in range(4):
                wm, wn = wave >> 1, wave & 1
                for j in range(NJ):
                    for i in range(MI):
                        for lane in range(64):
                            l15, lg = lane & 15, lane >> 4
                            nl = wn * (BN // 2) + j * 16 + lg * 4
                            ml = (wm * MI + i) * 16 + l15
                            yy, xx = y0 + (ml >> 4), x0 + (ml & 15)
                            for r in range(4):
                                if yy < H and xx < Wd and n0 + nl + r < N:
                                    y[b, yy, xx, n0 + nl + r] = acc[(wave, j, i)][lane][r]
    return y


if __name__ == '__main__':
    g = torch.Generator().manual_seed(0)
    for (B, H, Wd, C, N, TH, BN, CK) in ((1, 8, 16, 64, 64, 8, 64, 64), (2, 10, 20, 128, 64, 8, 64, 64), (1, 17, 33, 64, 128, 16, 128, 64),
                                         (1, 9, 16, 64, 96, 8, 64, 64), (1, 8, 16, 64, 64, 8, 64, 32), (2, 10, 20, 64, 128, 8, 128, 32),
                                         (1, 9, 17, 96, 96, 8, 128, 32), (1, 9, 17, 64, 40, 8, 64, 32), (1, 18, 21, 64, 128, 16, 128, 32)):
        x = torch.randn(B, H, Wd, C, generator=g, dtype=torch.float64)
        w = torch.randn(N, 3, 3, C, generator=g, dtype=torch.float64)
        ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
        got = torch.from_numpy(run_halo(x.numpy(), w.numpy(), TH, BN, CK=CK))
        err = (got - ref).abs().max().item()
        print(f'B{B} {H}x{Wd} C{C} N{N} TH{TH} BN{BN} CK{CK}: max|d| = {err:.3e}')
        assert err < 1e-9
    for (B, H, Wd, C, N, TH, BN, CK) in ((1, 8, 8, 64, 64, 8, 64, 64), (2, 5, 12, 64, 64, 8, 64, 64), (1, 8, 16, 64, 128, 16, 128, 64),
                                         (1, 8, 8, 64, 128, 8, 128, 32), (2, 5, 12, 32, 64, 8, 64, 32), (1, 9, 10, 32, 128, 16, 128, 32)):
        x = torch.randn(B, H, Wd, C, generator=g, dtype=torch.float64)
        w = torch.randn(N, 3, 3, C, generator=g, dtype=torch.float64)
        xu = torch.nn.functional.interpolate(x.permute(0, 3, 1, 2), scale_factor=2.0, mode='nearest')
        ref = torch.nn.functional.conv2d(xu, w.permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
        got = torch.from_numpy(run_halo(x.numpy(), w.numpy(), TH, BN, up=True, CK=CK))
        err = (got - ref).abs().max().item()
        print(f'upsampled read: B{B} {H}x{Wd} -> {2 * H}x{2 * Wd} C{C} N{N} TH{TH} BN{BN} CK{CK}: max|d| = {err:.3e}')
        assert err < 1e-9
