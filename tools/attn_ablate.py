"""Ablation builds of the attention kernels (cdna guide rule 17: ablate before optimising). Writes patched copies of
mos_attn.hip under _variants/, compiles each to an object and links it with the other translation units of the in-tree
build:   python tools/attn_ablate.py            -> _variants/libmos_abl_<name>.so (+ a list in _variants/abl_libs.txt)
Results of ablated kernels are WRONG by construction; only their timings mean something (tools/ab_kernels.sh).
  noexp    exp2 replaced by its argument                    (transcendental cost)
  novalu   no softmax arithmetic at all (accumulators converted as they are)
  nostage  tile prefetch + LDS staging stores skipped (tile 0 is reused; barrier kept)
  nosync   nostage + no per-tile barrier
  nomma2   the second MFMA group (P.V / dV,dK / dQ) skipped, operands kept alive
  mfma     nosync + novalu: LDS fragment reads + MFMAs only
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'mix-of-show_amd', 'csrc', 'mos_attn.hip')
VAR = os.path.join(ROOT, '_variants')
BUILD = os.path.join(ROOT, 'mix-of-show_amd', 'csrc', '_build')
FLAGS = ('--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable '
         '-ffp-contract=fast -mllvm -amdgpu-mfma-vgpr-form=1').split()


def sub(s, old, new, count=None):
    n = s.count(old)
    assert n >= 1 and (count is None or n == count), (old, n)
    return s.replace(old, new)


def patch(s, bits):
    NOEXP, NOSTAGE, NOSYNC, NOMMA2, NOVALU = (bits & 1, bits & 2, bits & 4, bits & 8, bits & 16)
    if NOEXP or NOVALU:
        s = re.sub(r'__builtin_amdgcn_exp2f\((s\[iq\]\[t\]\[r\] \* c - mc)\)', r'(\1)', s)          # fwd
        s = re.sub(r'__builtin_amdgcn_exp2f\((s\[r\] \* c - lse2)\)', r'(\1)', s)                   # dq
        s = re.sub(r'__builtin_amdgcn_exp2f\((s\[r\] \* c - l4\[r4\]\[rr\])\)', r'(\1)', s)         # dkdv
    if NOVALU:
        s = sub(s, 'const float p = (s[iq][t][r] * c - mc);', 'const float p = s[iq][t][r];')
        s = sub(s, '                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[iq][t][r]);', '                for (int r = 0; r < 1; ++r) {}')
        s = sub(s, 'const float p = (s[r] * c - lse2);', 'const float p = s[r];')
        s = sub(s, '                s[r] = p * g;', '                s[r] = g; asm volatile("" :: "v"(p));')
        s = sub(s, 'p = (s[r] * c - l4[r4][rr]);', 'p = s[r];')
        s = sub(s, '                    dp[r] = p * g;', '                    dp[r] = g;')
    if NOSTAGE:
        s = sub(s, '''        if (more) {
            kst.load(ksrc, (kv0 / KV_TILE + 1) * k_tile_bytes);''', '''        if (false) {
            kst.load(ksrc, (kv0 / KV_TILE + 1) * k_tile_bytes);''')
        s = sub(s, '''        if (more) {
            kst.store(Ks_ + (cur ^ 1) * HD<D>::ROW_TILE_ELEMS, tid);''', '''        if (false) {
            kst.store(Ks_ + (cur ^ 1) * HD<D>::ROW_TILE_ELEMS, tid);''')
        s = sub(s, '        if (more) {  // prefetch the next key tile into registers', '        if (false) {  // prefetch')
        s = sub(s, '''        if (more) {
            ktst.store_rows(Ks_ + (cur ^ 1) * RT, tid);''', '''        if (false) {
            ktst.store_rows(Ks_ + (cur ^ 1) * RT, tid);''')
        s = sub(s, '        if (more) load_tile(q0 + KV_TILE);', '')
        s = sub(s, '        if (more) store_tile(NB == 2 ? (cur ^ 1) : 0);', '')
        s = s.replace('        cur ^= 1;\n    };', '    };')                       # fwd / dq: keep reading buffer 0
        s = sub(s, '        if constexpr (NB == 2) cur ^= 1;', '')
    if NOSYNC:
        s = sub(s, '        __syncthreads();  // tile i+1 visible; every wave is done reading tile i before it is overwritten next round', '')
        s = sub(s, '''            ktst.store(Kt_ + (cur ^ 1) * TT, tid);
        }
        __syncthreads();''', '''            ktst.store(Kt_ + (cur ^ 1) * TT, tid);
        }''')
        s = sub(s, '''        if constexpr (NB == 1) __syncthreads();   // single buffer: every wave is done reading before the overwrite

        __syncthreads();''', '')
    if NOMMA2:
        s = sub(s, '                    for (int iq = 0; iq < NQ; ++iq) o[iq][dt] = MT<T>::mfma32(a, pf[iq][t][s2], o[iq][dt]);',
                '                    for (int iq = 0; iq < NQ; ++iq) asm volatile("" :: "v"(a), "v"(pf[iq][t][s2]));')
        s = sub(s, '                for (int s2 = 0; s2 < 2; ++s2) dq[dt] = MT<T>::mfma32(tk[dt][s2], dsf[s2], dq[dt]);',
                '                for (int s2 = 0; s2 < 2; ++s2) asm volatile("" :: "v"(tk[dt][s2]), "v"(dsf[s2]));')
        s = sub(s, '''                    dvT[dt] = MT<T>::mfma32(tdo[dt][s2], pf[s2], dvT[dt]);
                    dkT[dt] = MT<T>::mfma32(tq[dt][s2], dsf[s2], dkT[dt]);''',
                '''                    asm volatile("" :: "v"(tdo[dt][s2]), "v"(pf[s2]), "v"(tq[dt][s2]), "v"(dsf[s2]));''')
    return s


VARIANTS = dict(noexp=1, novalu=16, nostage=2, nosync=2 | 4, nomma2=8, mfma=2 | 4 | 16)


def main():
    os.makedirs(VAR, exist_ok=True)
    base = open(SRC).read()
    names = sys.argv[1:] or list(VARIANTS)
    procs = []
    for n in names:
        src = os.path.join(VAR, f'mos_attn_abl_{n}.hip')
        open(src, 'w').write(patch(base, VARIANTS[n]))
        obj = os.path.join(VAR, f'mos_attn_abl_{n}.o')
        procs.append((n, obj, subprocess.Popen(['/opt/rocm/bin/hipcc', *FLAGS, '-I' + os.path.dirname(SRC), '-c', src, '-o', obj])))
    libs = []
    for n, obj, p in procs:
        assert p.wait() == 0, n
        others = [os.path.join(BUILD, f + '.o') for f in 'mos_api mos_gemm mos_gram mos_norm mos_elem mos_conv'.split()]
        lib = os.path.join(VAR, f'libmos_abl_{n}.so')
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib, obj, *others], check=True)
        libs.append(f'{n}=_variants/libmos_abl_{n}.so')
        print('built', lib)
    open(os.path.join(VAR, 'abl_libs.txt'), 'w').write(' '.join(libs) + '\n')


if __name__ == '__main__':
    main()
