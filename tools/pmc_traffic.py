"""HBM traffic per launch of the library kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes), written with the fingerprint of the kernel sources it was measured on:

  bash tools/pmc_collect.sh attn                     # -> /tmp/pmc_fetch, /tmp/pmc_write (+ the SQ passes)
  python tools/pmc_traffic.py /tmp/pmc_fetch /tmp/pmc_write > gpurun_out/pmc_traffic.json

bench.py reports `roofline.traffic` from profiles/pmc_traffic.json only while the kernel sources still hash to
`source_sha16` (a number profiled on other code is stale -> null). Keys are the profiler names bench.py uses
("attn_bwd_dkdv f16 d40 B4 H8 Nq4096 Nkv4096"): kernel families are matched by symbol + template head dim + grid size.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def collect(d, counter):
    out = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get('Counter_Name') != counter:
                continue
            m = re.search(r'(attn_fwd_kernel|attn_bwd_dq_kernel|attn_bwd_dkdv_kernel|attn_bwd_dkdv_pipe_kernel|'
                          r'region_attn_kernel|gemm_lora_kernel|lora_grad_kernel|gram_kernel)I(DF16_|DF16b)Li(\d+)',
                          row.get('Kernel_Name', ''))
            if not m:
                continue
            # the slot-interleaved forms report under the profiler name of the kernel they replace (same launch site, same grid)
            sym = m.group(1).replace('_pipe_kernel', '_kernel')
            out[(sym, 'f16' if m.group(2) == 'DF16_' else 'bf16', int(m.group(3)), row.get('Grid_Size', ''))].append(
                float(row['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in out.items()}


# tools/bench_kernels.py --only attn launches these shapes; grid sizes identify them (threads = blocks * 256)
NAMES = {
    ('attn_bwd_dkdv_kernel', 'f16', 40): [('attn_bwd_dkdv f16 d40 B4 H8 Nq4096 Nkv4096', 4 * 8 * (4096 // 128) * 256)],
    ('attn_bwd_dq_kernel', 'f16', 40): [('attn_bwd_dq f16 d40 B4 H8 Nq4096 Nkv4096', 4 * 8 * (4096 // 128) * 256)],
    ('attn_fwd_kernel', 'f16', 40): [('attn_fwd f16 d40 B4 H8 Nq4096 Nkv4096', 4 * 8 * (4096 // 256) * 256),
                                     # regional sampling, CFG pair at 512x768: < 512 workgroups of 256 queries -> 128-query blocks
                                     ('attn_fwd f16 d40 B2 H8 Nq6144 Nkv6144', 2 * 8 * (6144 // 128) * 256)],
    ('region_attn_kernel', 'f16', 40): [('region_attn f16 d40 B2 H8 Nq6144 Nkv77 R3', 2 * 8 * (6144 // 128) * 256)],
    ('region_attn_kernel', 'f16', 80): [('region_attn f16 d80 B2 H8 Nq1536 Nkv77 R3', 2 * 8 * (1536 // 128) * 256)],
    ('region_attn_kernel', 'f16', 160): [('region_attn f16 d160 B2 H8 Nq384 Nkv77 R3', 2 * 8 * (384 // 128) * 256)],
}


def main(fetch_dir, write_dir):
    import bench
    fetch, write = collect(fetch_dir, 'FETCH_SIZE'), collect(write_dir, 'WRITE_SIZE')
    kernels = {}
    for (sym, dt, d, grid), fv in fetch.items():
        ent = next((e for e in NAMES.get((sym, dt, d), []) if str(e[1]) == str(grid)), None)
        if ent is None:
            continue
        wv = write.get((sym, dt, d, grid))
        rd = fv * 1024 * 2                 # FETCH_SIZE is in KB; x2 = the gfx950 correction of the guide
        wr = wv * 1024 if wv is not None else None
        kernels[ent[0]] = dict(read=rd, write=wr, total=rd + (wr or 0.0), grid=grid)
    print(json.dumps(dict(
        _comment='HBM bytes per launch from rocprofv3 --pmc passes over tools/bench_kernels.py --only attn,region (FETCH_SIZE and '
                 'WRITE_SIZE in separate passes; FETCH_SIZE KB x1024 x2 gfx950 correction, WRITE_SIZE KB x1024). '
                 'Valid only for the attention kernel sources with this fingerprint (bench.PMC_SOURCE_FILES; bench.py checks it).',
        source_sha16=bench.kernel_source_fingerprint(), kernels=kernels), indent=1))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
