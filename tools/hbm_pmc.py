"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; collected separately, as MI355X_MICROARCH.md prescribes) of
    rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -d <dir> -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline
into profiles/<name>.json: HBM bytes of the gather-convolution family (conv_kernel / conv32_kernel / wgrad_kernel and the pointwise pw_* kernels that serve the 1x1x1 single-channel layers) per train
step and per launch.  usage: python tools/hbm_pmc.py <fetch_dir> <write_dir> <out.json>"""
import csv
import glob
import json
import sys


def sums(d, counter):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    kb = {'conv': 0.0, 'conv_aux': 0.0, 'other': 0.0}
    n = {'conv': 0, 'conv_aux': 0, 'other': 0}
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        name = r['Kernel_Name']
        fam = 'conv' if any(k in name for k in ('conv_kernel', 'conv32_kernel', 'conv_thin_kernel', 'conv_dma_kernel', 'wgrad_kernel', 'wgrad_dma_kernel', 'wgrad_pw_dma_kernel', 'wgrad_thin_kernel', 'wgrad_thin_reduce', 'pw_cto', 'pw_1toc', 'pw_wgrad', 'pw_gemm', 'c1k3_', 'c1m_')) else 'other'
        if name.startswith('materialize_kernel') or name.startswith('reduce_partials_kernel'):
            fam = 'conv_aux'           # passes that belong to a vg_conv3d_wgrad / LDS-DMA vg_conv3d call: their bytes count for the family, not as launches
        kb[fam] += float(r['Counter_Value'])
        n[fam] += 1
    return kb, n


def main():
    fd, wd, out = sys.argv[1:4]
    steps = 2                                   # --steps 1 --warmup 1
    fk, fn = sums(fd, 'FETCH_SIZE')
    wk, wn = sums(wd, 'WRITE_SIZE')
    launches = fn['conv'] / steps
    rd = (fk['conv'] + fk['conv_aux']) * 1024 * 2 / steps          # gfx950: 128-B read requests are counted as 64 B (guide: double it)
    wr = (wk['conv'] + wk['conv_aux']) * 1024 / steps
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'van_gan_amd'))
    import build as _b                        # source hash of the kernels this run measured (bench.py drops a stale summary)
    json.dump({
        'csrc_hash': _b._src_hash(),
        'command': 'rocprofv3 --pmc FETCH_SIZE (and, in a separate pass, WRITE_SIZE) --kernel-trace --output-format csv -- '
                   'python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline   (tools/hbm_pmc.py)',
        'note': 'sums over every conv_kernel/conv32_kernel/conv_thin_kernel/conv_dma_kernel/wgrad_dma_kernel (+ its materialize / reduce_partials passes)/wgrad_pw_dma_kernel/wgrad_kernel/pw_* dispatch of 2 train steps (128^3, batch 1); FETCH_SIZE in '
                'KB doubled per MI355X_MICROARCH.md (gfx950 counts 128-B read requests at 64 B); WRITE_SIZE in KB taken as is '
                '(calibrated for 16-B/lane stores; the epilogue stores are 8 B/lane, so it is approximate)',
        'raw': {'FETCH_SIZE': {'sum_kb_2_steps': fk, 'dispatches_2_steps': fn},
                'WRITE_SIZE': {'sum_kb_2_steps': wk, 'dispatches_2_steps': wn}},
        'conv_family_launches_per_step': launches,
        'hbm_read_bytes_per_step': rd, 'hbm_write_bytes_per_step': wr,
        'other_kernels_hbm_bytes_per_step': (fk['other'] * 1024 * 2 + wk['other'] * 1024) / steps,
        'hbm_bytes_per_launch': (rd + wr) / launches,
    }, open(out, 'w'), indent=1)
    print(open(out).read())


if __name__ == '__main__':
    main()
