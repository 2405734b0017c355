#!/usr/bin/env python3
"""Counter summary of a round: reads the per-kernel PMC averages and kernel durations tools/gpu_round6_pmc.sh (round 5: gpu_round5_pmc.sh) left in a directory and writes
profiles/rNN_pmc_kernels.json (one entry per kernel: counters, derived fractions, a sentence for bench.py's notes) and a text table.
usage: pmc_summary.py gpurun_out/r6pmc profiles/r06_pmc_kernels.json [r06] > profiles/r06_pmc_kernels.txt"""
import json, re, sys

d, out = sys.argv[1], sys.argv[2]
RND = sys.argv[3] if len(sys.argv) > 3 else 'r06'
SIMDS, XCDS = 1024, 8


def counters(path):
    res, cur = {}, None
    for ln in open(path):
        if ln.startswith('== '):
            cur = res.setdefault(ln[3:].strip(), {})
        elif cur is not None and 'launches' in ln:
            p = ln.split()
            cur[p[0]] = float(p[3])
            cur['launches'] = int(p[1])
    return res


def durations(path):
    res = {}
    for ln in open(path):
        if ln.startswith('#') or not ln.strip():
            continue
        m = re.match(r'(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$', ln)
        if m:
            res[m.group(1).strip()] = dict(calls=int(m.group(2)), avg_us=float(m.group(4)))
    return res


def dur_of(table, sub):
    hit = [(k, v) for k, v in table.items() if sub in k]
    calls = sum(v['calls'] for _, v in hit)
    return sum(v['avg_us'] * v['calls'] for _, v in hit) / calls if calls else None


cal = counters(d + '/fetch_calibration.txt')
factor = {k: 1048576.0 / v['FETCH_SIZE'] for k, v in cal.items()}
bench, bdur = counters(d + '/pmc_bench_kernels.txt'), durations(d + '/bench_kernel_stats.txt')
vec, vdur = counters(d + '/pmc_vecchia_kernels.txt'), durations(d + '/vecchia_kernel_stats.txt')
kst = counters(d + '/pmc_kmatrix_standalone.txt')
res = {'_calibration': {'what': 'FETCH_SIZE (KB) reported for 1 GiB streamed once, by load form (tools/ubench/fetch_calib.hip)',
                        'true_over_reported': factor,
                        'note': 'every load form the library uses -- 16-byte plain, 16-byte buffer sc1, 8-byte sc1 -- reads exactly 1/2: FETCH_SIZE x 2 is the '
                                'correction for each of them, not a blanket guess'}}
print('# ' + RND + ' counter evidence (tools/gpu_round%s_pmc.sh;' % RND[-1] + ' rocprofv3 --kernel-trace --pmc, one counter set per pass, the program itself after "--").')
print('# FETCH_SIZE calibration (1 GiB streamed once): true / reported = ' + ', '.join('%s %.3f' % (k, v) for k, v in factor.items()))
print('# Derived columns: clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; VALU issue = SQ_INSTS_VALU x 4 cycles / 1024 SIMDs / elapsed cycles (an f64 wave-instruction')
print('#   issues in 4 cycles; 32-bit ones in this count issue faster, so this is an upper bound of the f64 share); MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 / elapsed cycles;')
print('#   TFLOP/s = SQ_INSTS_MFMA x 2048 / duration; HBM-side bytes = FETCH_SIZE x 2 (calibrated) + WRITE_SIZE.')


def entry(name, c, us, extra=None):
    e = dict(launches=c.get('launches'), avg_us=us, counters={k: v for k, v in c.items() if k != 'launches'})
    cyc = c.get('GRBM_GUI_ACTIVE', 0.0) / XCDS
    if cyc and us:
        e['clock_GHz'] = cyc / us / 1e3
        e['valu_issue_frac'] = c.get('SQ_INSTS_VALU', 0.0) * 4.0 / SIMDS / cyc
        e['mfma_busy_frac'] = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / SIMDS / cyc
        e['mfma_tflops'] = c.get('SQ_INSTS_MFMA', 0.0) * 2048.0 / us / 1e6
    if 'FETCH_SIZE' in c:
        e['hbm_side_bytes_per_launch'] = (2.0 * c['FETCH_SIZE'] + c.get('WRITE_SIZE', 0.0)) * 1024.0
        if us:
            e['hbm_side_GBs'] = e['hbm_side_bytes_per_launch'] / us / 1e3
    if c.get('SQ_WAVES'):
        e['valu_per_wave'] = c.get('SQ_INSTS_VALU', 0.0) / c['SQ_WAVES']
    e.update(extra or {})
    res[name] = e
    print('\n== %s: %s launches, %.1f us on average' % (name, e['launches'], us or float('nan')))
    for k, v in sorted(e['counters'].items()):
        print('   %-30s %18.1f per launch' % (k, v))
    print('   -> ' + ', '.join('%s %.3g' % (k, v) for k, v in e.items() if isinstance(v, float) and k != 'avg_us'))
    return e


e = entry('potrf_mega_kernel', bench['potrf_mega_kernel'], dur_of(bdur, 'potrf_mega_kernel'))
e['note'] = ('counters over all %d launches of the bench command (no-op speculative batches included): %.2f M v_mfma_f64_16x16x4 per launch = %.1f GFLOP, '
             'f64 MFMA pipes busy %.2f of the elapsed cycles, %.1f M VALU wave-instructions, HBM-side traffic %.2f GB per launch = FETCH_SIZE x 2 + WRITE_SIZE with the x 2 '
             'calibrated for every load form of the kernel' % (e['launches'], e['counters']['SQ_INSTS_MFMA'] / 1e6, e['counters']['SQ_INSTS_MFMA'] * 2048 / 1e9,
                                                              e['mfma_busy_frac'], e['counters']['SQ_INSTS_VALU'] / 1e6, e['hbm_side_bytes_per_launch'] / 1e9))
for nm, key in (('kmatrix_kernel', 'kmatrix_kernel'), ('kmatrix_multi_kernel', 'kmatrix_multi_kernel')):
    e = entry(nm, bench[key], dur_of(bdur, key + ('<' if nm == 'kmatrix_kernel' else '(')))
    per_entry = e['valu_per_wave'] / 16.0
    e['valu_per_entry'] = per_entry
    e['note'] = ('%s in the bench command: %.0f VALU wave-instructions per matrix entry (16 entries per lane), VALU issue %.2f of the peak, %.1f MB written per launch = %.2f of 8 TB/s over '
                 'its %.0f us: a launch of one to three rounds of workgroups, bound by neither -- by a workgroup\'s own latency (inputs, arithmetic, store drain)'
                 % (nm, per_entry, e['valu_issue_frac'], e['counters']['WRITE_SIZE'] / 1024.0, e['counters']['WRITE_SIZE'] * 1024 / e['avg_us'] / 1e3 / 8000.0, e['avg_us']))
e = entry('grad_reduce_multi_kernel', bench['grad_reduce'], dur_of(bdur, 'grad_reduce_multi_kernel'))
e = entry('linkgp_Jsep_kernel', bench['linkgp_Jsep_kernel'], dur_of(bdur, 'linkgp_Jsep_kernel'))
e['note'] = ('%.1f M f64 MFMA per launch = %.1f TFLOP/s executed, MFMA pipes busy %.2f and VALU issue %.2f of the elapsed cycles (the two share the double-precision units: %.2f together), '
             'HBM-side traffic %.1f GB per launch = %.1f TB/s (record re-reads that miss the XCD\'s L2)'
             % (e['counters']['SQ_INSTS_MFMA'] / 1e6, e['mfma_tflops'], e['mfma_busy_frac'], e['valu_issue_frac'], e['mfma_busy_frac'] + e['valu_issue_frac'],
                e['hbm_side_bytes_per_launch'] / 1e9, e['hbm_side_GBs'] / 1e3))
entry('matern_records_kernel', bench['matern_records_kernel'], dur_of(bdur, 'matern_records_kernel'))
entry('gp_quad_kernel', bench['gp_quad_kernel'], dur_of(bdur, 'gp_quad_kernel'))
e = entry('vecchia_row4_kernel', vec['vecchia_row4_kernel'], dur_of(vdur, 'vecchia_row4_kernel'))
e['note'] = ('f64 VALU issue: %.0f VALU wave-instructions per wave of four rows, %.2f of the issue peak over the launches of tools/gpu_vecchia_rowbench.py (likelihood, batched likelihood, '
             'objective + gradient, sparse-factor rows), HBM-side traffic %.0f MB per launch = %.2f of 8 TB/s' % (e['valu_per_wave'], e['valu_issue_frac'], e['hbm_side_bytes_per_launch'] / 1e6, e['hbm_side_GBs'] / 8000.0))
for nm in kst:
    c = kst[nm]
    cyc = c['GRBM_GUI_ACTIVE'] / XCDS
    us = cyc / 2.4e3   # (no duration table for this command: elapsed cycles at 2.4 GHz)
    e = entry('standalone ' + nm + ' n=8192 D=10 full', c, us, dict(duration_from='GRBM_GUI_ACTIVE / 8 at 2.4 GHz'))
    e['valu_per_entry'] = e['valu_per_wave'] / 16.0
# round 6: the training form of K assembly (lower tiles, ten matrices, n = 5000) and the SExp pair kernel at cfg3's shape, if their passes were taken
import os as _os
if _os.path.exists(d + '/pmc_kmatrix_lower.txt'):
    for nm, c in counters(d + '/pmc_kmatrix_lower.txt').items():
        cyc = c['GRBM_GUI_ACTIVE'] / XCDS
        e = entry('standalone ' + nm + ' n=5000 D=10 lower tiles, 10 matrices', c, cyc / 2.4e3, dict(duration_from='GRBM_GUI_ACTIVE / 8 at 2.4 GHz'))
        e['valu_per_entry'] = e['valu_per_wave'] / 16.0
        e['note'] = ('training form (lower tiles of ten n = 5000 matrices): %.0f VALU wave-instructions per matrix entry, VALU issue %.2f of the elapsed cycles (an f64 instruction '
                     'issues in 4 cycles): the kernel is bound by its double-precision arithmetic, not by its stores' % (e['valu_per_entry'], e['valu_issue_frac']))
if _os.path.exists(d + '/pmc_sexp_pair.txt'):
    for nm, c in counters(d + '/pmc_sexp_pair.txt').items():
        cyc = c['GRBM_GUI_ACTIVE'] / XCDS
        e = entry(nm, c, cyc / 2.4e3, dict(duration_from='GRBM_GUI_ACTIVE / 8 at 2.4 GHz'))
        issue = (c.get('SQ_INSTS_VALU', 0.0) * 4.0 + c.get('SQ_INSTS_MFMA', 0.0) * 64.0) / SIMDS / cyc
        e['issue_cycles_frac'] = issue
        e['note'] = ('SExp pair kernel at cfg3\'s shape: per launch %.1f M VALU wave-instructions (4 issue cycles each) + %.2f M f64 MFMAs (64 each) = %.2f of the SIMDs\' elapsed cycles: the kernel '
                     'is bound by instruction issue (MFMA busy %.2f)' % (c.get('SQ_INSTS_VALU', 0.0) / 1e6, c.get('SQ_INSTS_MFMA', 0.0) / 1e6, issue, e.get('mfma_busy_frac', float('nan'))))
json.dump(res, open(out, 'w'), indent=1)
# ... and the potrf entry in the form bench.py reads for roofline.traffic (profiles/rNN_pmc_bench_potrf_kernel.json)
import os
pe = res['potrf_mega_kernel']
cmd = next((ln[1:].strip() for ln in open(d + '/pmc_bench_kernels.txt') if ln.startswith('#')), '')
json.dump({'kernel': 'potrf_mega_kernel', 'launches': pe['launches'], 'fetch_size_kb_per_launch_raw': pe['counters']['FETCH_SIZE'],
           'write_size_kb_per_launch': pe['counters']['WRITE_SIZE'], 'hbm_bytes_per_launch': pe['hbm_side_bytes_per_launch'],
           'how': 'FETCH_SIZE x 2 + WRITE_SIZE per launch, the x 2 calibrated for each load form of the kernel (16-byte plain, 16-byte buffer sc1, 8-byte sc1 all report '
                  'exactly 1/2: profiles/' + RND + '_pmc_fetch_calibration.txt); fabric-side traffic, Infinity-Cache hits included',
           'command': cmd.split('(one pass per set')[0].strip().replace('<set>', 'FETCH_SIZE|WRITE_SIZE (separate passes)')},
          open(os.path.join(os.path.dirname(out), RND + '_pmc_bench_potrf_kernel.json'), 'w'))
