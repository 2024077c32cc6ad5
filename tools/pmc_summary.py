"""Summarise rocprofv3 --pmc output directories: per kernel name and grid size, the mean of every counter."""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    m = re.search(r'(attn_fwd_kernel|attn_bwd_dq_kernel|attn_bwd_dkdv_kernel|attn_bwd_dkdv_pipe_kernel|region_attn_kernel|'
                  r'conv3x3_halo_kernel|attn_probs_kernel|attn_pv_kernel|gn_col_kernel|'
                  r'gemm_lora_kernel|conv3x3_nhwc_kernel|gn_nhwc_reduce_kernel|gn_nhwc_apply_kernel|lora_grad_kernel|'
                  r'gram_kernel|lsq_grad_mfma_kernel)', name)
    if not m:
        return None
    dt = re.search(r'I(DF16_|DF16b)', name)
    extra = ('f16' if dt.group(1) == 'DF16_' else 'bf16') if dt else ''
    # every integral / bool template argument, in order (tile sizes, head dim, flags, pipeline stages)
    args = re.findall(r'L([ib])(\d+)E', name)
    extra += ' <' + ','.join(v for _, v in args) + '>'
    return f'{m.group(1)} {extra}'


def main(dirs):
    data = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            for row in csv.DictReader(open(f)):
                k = short(row.get('Kernel_Name', ''))
                if k is None:
                    continue
                key = (k, row.get('Grid_Size', ''), row.get('LDS_Block_Size', ''), row.get('VGPR_Count', ''),
                       row.get('Accum_VGPR_Count', ''))
                data[key][row['Counter_Name']].append(float(row['Counter_Value']))
    for key in sorted(data):
        print(f'== {key[0]} grid={key[1]} lds={key[2]} vgpr={key[3]} agpr={key[4]}')
        c = {n: sum(v) / len(v) for n, v in data[key].items()}
        for n in sorted(c):
            print(f'   {n:28s} {c[n]:16.1f}')
        if 'SQ_WAVE_CYCLES' in c and c['SQ_WAVE_CYCLES'] > 0:
            w = c['SQ_WAVE_CYCLES']
            parts = {n: c.get(n, 0) / w for n in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY',
                                                  'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS')}
            print('   fractions of wave cycles: ' + ' '.join(f'{n[3:]}={v:.3f}' for n, v in parts.items()))
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'SQ_BUSY_CYCLES' in c and c['SQ_BUSY_CYCLES'] > 0:
            print(f"   mfma_busy/busy_cycles = {c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_BUSY_CYCLES']:.3f}")
        if 'FETCH_SIZE' in c:
            print(f"   HBM read bytes (FETCH_SIZE KB x1024 x2 gfx950 correction) = {c['FETCH_SIZE'] * 1024 * 2:.3e}")
        if 'WRITE_SIZE' in c:
            print(f"   HBM write bytes (WRITE_SIZE KB x1024, uncalibrated)        = {c['WRITE_SIZE'] * 1024:.3e}")


if __name__ == '__main__':
    main(sys.argv[1:])
