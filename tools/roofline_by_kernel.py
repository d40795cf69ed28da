"""Roofline of every kernel template of the gather-convolution family, joined from three measurements of the same workload
(128^3, batch 1, serial schedule VG_LANES=0 VG_SIDE_STREAM=0 so that durations are attributable):

  1. bench.py --dump-kernels <json>       per template: launches, HIP-event ms, algorithmic FLOPs and bytes (ops.KernelProfile)
  2. rocprofv3 --kernel-trace --stats     per kernel name: calls, total duration                     (<dir>/**/*kernel_stats.csv)
  3. rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes)  per dispatch HBM bytes      (counter_collection.csv)

into profiles/<name>.json: kernel template, launches per step, ms per step (rocprof and HIP events, which must agree), GFLOP,
TFLOP/s, fraction of the 2.5 PFLOP/s dense bf16 peak, algorithmic bytes, measured HBM bytes (FETCH_SIZE doubled on gfx950 per
MI355X_MICROARCH.md, WRITE_SIZE as is), their ratio, and the fraction of the 8 TB/s HBM peak -- i.e. which bound each template
sits closest to.

usage: python tools/roofline_by_kernel.py <bench_kernels.json> <stats_dir> <steps_in_stats_run> <fetch_dir> <write_dir> <steps_in_pmc_runs> <out.json>
(the PMC directories may be '-' to leave the byte columns empty)"""
import csv
import glob
import json
import os
import re
import sys

PEAK_TF, PEAK_HBM = 2500.0, 8.0e12


def tname(t):
    return {'unsigned short': 'bf16', 'float': 'f32'}.get(t.strip(), t.strip())


def b(x):
    return '1' if x.strip() == 'true' else '0'


def variant_of(kernel_name: str):
    """rocprof kernel name -> the variant string of vg_conv3d_variant / vg_conv3d_wgrad_variant (without |walk|ch flags)."""
    if 'pw_wgrad_cc_kernel' in kernel_name:
        return 'pw_wgrad_cc'
    if kernel_name.startswith('materialize_kernel'):          # the operand pass and the partial-slab sum of the DMA weight gradients:
        return 'wgrad_dma:materialize'                        # (also the operand pass of conv_dma) rows of their own (time and HBM bytes; their FLOPs are the wgrad_dma rows')
    if kernel_name.startswith('reduce_partials_kernel'):
        return 'wgrad:reduce_partials'
    m = re.search(r'(\w+)_kernel<([^>]*)>', kernel_name)
    if not m:
        return None
    k, args = m.group(1), [a.strip() for a in m.group(2).split(',')]
    if k == 'conv' and len(args) == 8:
        return 'conv<%s,%s,%s,n%s,wl%s,dma%s,mc%s,c1%s>' % (tname(args[0]), args[1], args[2], b(args[3]), b(args[4]), b(args[5]), args[6], b(args[7]))
    if k == 'conv_thin' and len(args) >= 4:
        return '%s<m%s,b%s,r%s,s%s%s%s>' % ('conv_thin2' if len(args) > 5 and args[5] == '2' else 'conv_thin', args[0], b(args[1]), b(args[2]), b(args[3]),
                                            ',bs' if len(args) > 4 and b(args[4]) == '1' else '', ',pl' if len(args) > 6 and args[6] != '1' else '')       # pl: output panels looped over one staged halo
    if k == 'pw_gemm' and len(args) == 6 and args[5] == 'true':
        return 'pw_gemm_split<%s,%s>' % (args[0], args[1])
    if k == 'pw_gemm' and len(args) in (5, 6):
        return 'pw_gemm<%s,%s,g%s,a%s>' % (args[0], args[1], b(args[2]), b(args[3]))
    if k == 'conv_dma' and len(args) in (3, 4):               # (4th: BSTAT -- the statistics epilogue; same row)                 # <NW, MW, GT>: the variant string names the tile (BN = 64 NW, BM = 128 MW); the GT instances share a row
        return 'conv_dma<%d,%d>' % (64 * int(args[0]), 128 * int(args[1]))
    if k == 'conv32' and len(args) == 4:
        return 'conv32<%s,%s,n%s,cp%s>' % (args[0], args[1], b(args[2]), b(args[3]))
    if k == 'wgrad_dma' and len(args) == 3:
        return 'wgrad_dma<%s,%s,d%s>' % (args[0], args[1], b(args[2]))
    if k == 'wgrad_thin' and len(args) == 1:
        return 'wgrad_thin<m%s>' % args[0]
    if k == 'wgrad_pw_dma' and len(args) == 2:
        return 'wgrad_pw_dma<%s,%s>' % (args[0], args[1])
    if k == 'wgrad' and len(args) == 4:
        return 'wgrad<%s,%s,%s,n%s>' % (tname(args[0]), args[1], args[2], b(args[3]))
    if k in ('pw_cto1', 'pw_ctoc') and len(args) == 2:
        return '%s<%s,%s>' % (k, tname(args[0]), args[1])
    if k == 'pw_1toc':
        return 'pw_1toc<%s>' % tname(args[0])
    if k == 'pw_wgrad' and len(args) == 2:
        return 'pw_wgrad<%s,%s>' % (tname(args[0]), 'cto1' if args[1] == 'true' else '1toc')
    if k in ('c1k3_fwd', 'c1k3_wgrad') and len(args) == 2:
        return '%s<%s,%s>' % (k, tname(args[0]), tname(args[1]))
    if k == 'c1m_fwd' and len(args) == 5:                   # <S, KS, ST, MT, NOISE>
        return 'c1m_fwd<%s,%s,%s,%s,n%s>' % (tname(args[0]), args[1], args[2], args[3], b(args[4]))
    if k == 'c1m_wgrad' and len(args) == 6:                 # <S, KS, ST, TH, NT, NOISE>
        return 'c1m_wgrad<%s,%s,%s,%s,n%s>' % (tname(args[0]), args[1], args[2], args[4], b(args[5]))
    return None


def read_stats(d):
    f = glob.glob(d + '/**/*kernel_stats.csv', recursive=True)
    out = {}
    for r in csv.DictReader(open(f[0])):
        v = variant_of(r['Name'])
        if v:
            e = out.setdefault(v, {'calls': 0, 'ns': 0.0, 'name': r['Name'].split('(')[0]})
            e['calls'] += int(r['Calls']); e['ns'] += float(r['TotalDurationNs'])
    return out


def read_pmc(d, counter):
    out = {}
    if d == '-':
        return out
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        v = variant_of(r['Kernel_Name'])
        if v:
            out[v] = out.get(v, 0.0) + float(r['Counter_Value'])
    return out


def main():
    kjson, sdir, ssteps, fdir, wdir, psteps, outp = sys.argv[1:8]
    ssteps, psteps = int(ssteps), int(psteps)
    bench = json.load(open(kjson))
    stats = read_stats(sdir)
    fetch, write = read_pmc(fdir, 'FETCH_SIZE'), read_pmc(wdir, 'WRITE_SIZE')
    # one template may serve several kinds (forward and data gradient): merge the bench rows per template
    merged = {}
    for r in bench:
        r['kernel'] = re.sub(r',bs[12]>', ',bs>', re.sub(r'pw_wgrad_cc<r\d+>', 'pw_wgrad_cc', r['kernel']))
        e = merged.setdefault(r['kernel'], {'kinds': [], 'launches': 0, 'ms_events': 0.0, 'gflop': 0.0, 'alg_bytes': 0.0})
        e['kinds'].append(r['kind']); e['launches'] += r['launches']; e['ms_events'] += r['ms']; e['gflop'] += r['gflop']
        e['alg_bytes'] += r['algorithmic_bytes']
    for aux in ('wgrad_dma:materialize', 'wgrad:reduce_partials'):
        if aux in stats:
            merged[aux] = {'kinds': ['conv_wgrad (auxiliary pass)'], 'launches': int(stats[aux]['calls'] / ssteps), 'ms_events': 0.0, 'gflop': 0.0, 'alg_bytes': 0.0}
    rows = []
    for k, e in merged.items():
        st = stats.get(k)
        ms_prof = st['ns'] / 1e6 / ssteps if st else None
        ms = ms_prof if ms_prof else e['ms_events']
        hbm = None
        if k in fetch or k in write:
            hbm = (fetch.get(k, 0.0) * 1024 * 2 + write.get(k, 0.0) * 1024) / psteps
        tf = e['gflop'] / ms if ms else None                  # GFLOP / ms = TFLOP/s
        rows.append({'kernel': k, 'rocprof_name': st['name'] if st else None, 'kinds': sorted(set(e['kinds'])),
                     'launches_per_step': e['launches'], 'rocprof_calls_per_step': st['calls'] / ssteps if st else None,
                     'ms_per_step_rocprof': ms_prof, 'ms_per_step_hip_events': e['ms_events'], 'gflop_per_step': e['gflop'],
                     'tflops': tf, 'frac_mfma_peak': tf / PEAK_TF if tf else None,
                     'algorithmic_bytes_per_step': e['alg_bytes'], 'hbm_bytes_per_step': hbm,
                     'hbm_over_algorithmic': hbm / e['alg_bytes'] if hbm and e['alg_bytes'] else None,
                     'frac_hbm_peak': (hbm / (ms * 1e-3)) / PEAK_HBM if hbm and ms else None})
    rows.sort(key=lambda r: -(r['ms_per_step_rocprof'] or r['ms_per_step_hip_events']))
    tot_ms = sum(r['ms_per_step_rocprof'] or r['ms_per_step_hip_events'] for r in rows)
    tot_gf = sum(r['gflop_per_step'] for r in rows)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'van_gan_amd'))
    import build as _b
    json.dump({'csrc_hash': _b._src_hash(), 'workload': '128^3, batch 1, one train step, serial schedule',
               'family': {'ms_per_step': tot_ms, 'gflop_per_step': tot_gf, 'tflops': tot_gf / tot_ms, 'frac_mfma_peak': tot_gf / tot_ms / PEAK_TF},
               'peaks': {'mfma_bf16_tflops': PEAK_TF, 'hbm_bytes_per_s': PEAK_HBM}, 'kernels': rows}, open(outp, 'w'), indent=1)
    print('%-46s %5s %8s %8s %7s %7s %7s' % ('kernel', 'n', 'ms', 'TF/s', '%mfma', 'hbm/alg', '%hbm'))
    for r in rows[:25]:
        ms = r['ms_per_step_rocprof'] or r['ms_per_step_hip_events']
        print('%-46s %5d %8.3f %8.1f %7.1f %7s %7s' % (r['kernel'], r['launches_per_step'], ms, r['tflops'] or 0, 100 * (r['frac_mfma_peak'] or 0),
                                                   '-' if not r['hbm_over_algorithmic'] else '%.2f' % r['hbm_over_algorithmic'],
                                                   '-' if not r['frac_hbm_peak'] else '%.1f' % (100 * r['frac_hbm_peak'])))
    print('family: %.2f ms, %.0f TFLOP/s = %.1f %% of the dense bf16 peak' % (tot_ms, tot_gf / tot_ms, 100 * tot_gf / tot_ms / PEAK_TF))


if __name__ == '__main__':
    main()
