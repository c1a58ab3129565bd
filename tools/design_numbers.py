"""Fills the @@name@@ fields of a DESIGN.md template with the numbers of a profile collection (tools/collect_profiles.sh):
    python tools/design_numbers.py tools/DESIGN.template.md profiles > DESIGN.md
Every field is read from the file DESIGN.md names next to it; a field the files do not yield stays visible as @@name@@."""
import csv
import json
import os
import re
import sys


def stats(path):
    rows = {}
    with open(path) as f:
        for r in csv.DictReader(l for l in f if not l.startswith('#')):
            rows[r['kernel']] = r
    return rows


def main(template, d):
    v = {}
    P = lambda n: os.path.join(d, n)
    # ---- parity / gradient stats
    t = open(P('r06_parity_stats.txt')).read()
    m = re.search(r'tol 0\.0001: (\d+) tensors, worst (\S+) of the limit', t)
    v['parity_n'], v['parity_worst'] = m.group(1), m.group(2)
    t = open(P('r06_grad_stats.txt')).read()
    tight = re.findall(r'tight, \S+: (\d+) tensors, worst element (\S+) x scale', t)
    v['grad_tight_n'] = str(sum(int(a) for a, _ in tight)); v['grad_tight_worst'] = '%.1e' % max(float(b) for _, b in tight)
    v['grad_loose_worst'] = '%.1e' % max(float(b) for b in re.findall(r'loose, \S+: \d+ parameter tensors, worst element (\S+) x scale', t))
    # ---- forward kernels of the frame
    h = stats(P('r06_kernel_stats_headline.csv'))
    for name in ('k_forward_head', 'k_sample', 'k_tsdf', 'k_decode_lc16', 'k_decode_high_g', 'k_attention_g', 'k_fallback_points', 'k_composite'):
        r = next((r for k, r in h.items() if name + '(' in k or name + '<' in k), None)
        if r:
            v[name] = '%.0f' % float(r['avg_us']) if float(r['avg_us']) >= 20 else '%.1f' % float(r['avg_us'])
    # ---- the fused iteration
    tr = stats(P('r06_kernel_stats_train.csv'))
    iters = int(next(r for k, r in tr.items() if 'k_mapper_loss' in k)['calls'])
    per = lambda *names: sum(float(r['total_ms']) * 1e3 / iters for k, r in tr.items() if any(n in k for n in names))
    zero = per('k_zero_multi')
    head = per('k_backward_head', 'k_max_reduce')
    v['train_head'] = '%.0f' % (per('k_prefilter_mask', 'k_pack_multi', 'k_forward_head', 'k_sample', 'k_tsdf(') + (zero if head else zero / 2))
    v['k_decode_lc16_train'] = '%.0f' % per('k_decode_lc16_train')
    v['train_inband_fwd'] = '%.0f' % per('k_decode_h<64', 'k_attention_h<1')
    v['train_mid'] = '%.0f' % per('k_fallback_points', 'k_composite(', 'k_mapper_loss')
    v['train_bwd_head'] = '%.0f' % (head if head else per('k_composite_bwd', 'k_bin_keys') + zero / 2)
    v['train_sort'] = '%.0f' % per('k_rs_')
    v['train_att_bwd'] = '%.0f' % per('k_attention_bwd_h', 'k_outer_h', 'k_reduce_partials_scaled')
    v['train_hl_bwd'] = '%.0f' % per('k_decode_bwd_h<')
    v['k_decode_bwd_roles'] = '%.0f' % per('k_decode_bwd_roles')
    v['k_reduce_roles'] = '%.0f' % per('k_reduce_partials_roles')
    v['k_scatter_sorted'] = '%.0f' % per('k_scatter_sorted')
    v['train_adam'] = '%.0f' % per('k_masked_adam_multi', 'k_adam_cl_multi', 'k_adam_step')
    v['n_launches'] = '%.0f' % sum(int(r['calls']) / iters for k, r in tr.items() if (k.startswith('k_') or k.startswith('void k_')) and int(r['calls']) >= iters)
    # ---- bench line
    b = json.loads(open(P('r06_bench_f16x3.json')).read().strip().split('\n')[-1])
    v['headline_value'] = '%.1f' % (b['value'] / 1e6); v['headline_ms'] = '%.2f' % b['ms_per_step']
    cfg = b['config']
    v['value_f32'] = '%.1f' % (cfg['exact_f32_value'] / 1e6)
    v['x20_value'] = '%.1f' % (cfg['value_at_x20_grids'] / 1e6)
    v['x20_parity'] = 'max-rel depth %.1e / colour %.1e' % (cfg['parity_max_rel_depth_at_x20'], cfg['parity_max_rel_color_at_x20'])
    lim = b['roofline']['limiter']
    v['loop_mfma'] = str(lim['per_tile_budget']['mfma_instructions']); v['loop_valu'] = '{:,}'.format(lim['per_tile_budget']['valu_instructions']).replace(',', ' ')
    v['limiter_cycles'] = '{:,.0f}'.format(lim['simd_cycles_per_tile']).replace(',', ' ')
    v['limiter_ceiling'] = '{:,.0f}'.format(lim['per_tile_budget']['mfma_pipe_cycles'] + 2.2 * lim['per_tile_budget']['valu_instructions']).replace(',', ' ')
    v['limiter_frac'] = '%.2f' % lim['frac_of_that_ceiling']; v['limiter_cpi'] = '%.2f' % lim['valu_issue_cycles_per_instruction']
    sm = cfg['shard_model']
    v['k8_shard_ms'] = '%.3f' % sm['k8']['ms_slowest_shard']; v['k8_ideal_ms'] = '%.3f' % (b['ms_per_step'] / 8)
    v['k8_bound_incl'] = '%.2f' % sm['k8']['speedup_bound_incl_gather']
    ro = b['roofline']
    v['avg_launch_ms'] = '%.2f' % ro['avg_launch_ms']
    v['roofline_achieved'] = '%.0f' % ro['achieved']; v['roofline_frac'] = '%.3f' % ro['frac']; v['roofline_exec_frac'] = '%.3f' % ro['frac_executed']
    pmc = ro['limiter'].get('pmc', {})
    v['mfma_busy'] = '%.3f' % pmc['mfma_busy_frac'] if 'mfma_busy_frac' in pmc else '@@mfma_busy@@'
    v['clock'] = '%.2f' % pmc['clock_ghz'] if 'clock_ghz' in pmc else '@@clock@@'
    v['traffic_bps'] = '%.1f' % (ro['traffic'] / ro['points_per_launch']) if ro.get('traffic') else '@@traffic_bps@@'
    v['tsdf_gbps'] = '%.0f' % b['roofline_tsdf']['achieved']; v['tsdf_frac'] = '%.2f' % b['roofline_tsdf']['frac']
    v['cpu_value'] = '%.0f' % b['cpu_baseline']['value']; v['cpu_cores'] = str(b['cpu_baseline']['cores'])
    rn = b.get('replica_native_frame') or b['config'].get('replica_native_frame')
    v['replica_ms'] = '%.2f' % rn['ms_per_frame']; v['replica_value'] = '%.1f' % (rn['rays_per_s'] / 1e6)
    tb = b['tracker_iteration']['by_batch']
    v['tracker_200'] = '%.3f' % tb['200']['ms_per_iteration']; v['tracker_1000'] = '%.3f' % tb['1000']['ms_per_iteration']
    c1 = b['config1']
    v['config1'] = '%.3f ms forward (%.1f M rays/s), %.2f ms forward + loss + backward through autograd' % (c1['forward']['ms'], c1['forward']['value'] / 1e6, c1['forward_backward']['ms'])
    v['config3_ms'] = '%.3f' % b['config3']['ms_per_iteration']
    v['torch_speedup'] = '%.0f' % b['torch_gpu_baseline']['speedup']
    v['shard_bound'] = '%.2f' % b['config']['shard_model']['k8']['speedup_bound']
    c5 = b['config5']; rr = c5['random_ray_order']
    v['c5_given_gbps'] = '%.0f' % rr['as_given']['tsdf_algorithmic_gbps']; v['c5_sorted_gbps'] = '%.0f' % rr['sorted']['tsdf_algorithmic_gbps']
    fb = rr['as_given'].get('tsdf_counter_bytes_per_sample')
    v['c5_fetch_b'] = '%.0f' % fb if fb else '@@c5_fetch_b@@'
    v['c5_given_ms'] = '%.2f' % rr['as_given']['ms_per_batch']; v['c5_sorted_ms'] = '%.2f' % rr['sorted']['ms_per_batch']
    v['c5_pixel_value'] = '%.1f' % (c5['value'] / 1e6)
    # ---- training
    t = open(P('r06_fused_iteration.txt')).read()
    avg = lambda xs: sum(xs) / len(xs)
    v['iter_5000'] = '%.3f' % avg([float(x) for x in re.findall(r'fused iteration, graph replay, 5000 x 64: ms per iteration (\S+)', t)])
    v['iter_5000_one_stream'] = '%.3f' % avg([float(x) for x in re.findall(r'one stream \(ADFP_SIDE_LANE=0\), graph replay, 5000 x 64: ms per iteration (\S+)', t)])
    v['iter_1000'] = '%.3f' % avg([float(x) for x in re.findall(r'1000 x 48: ms per iteration (\S+)', t)])
    for line in open(P('r06_bench_train.json')):
        if line.strip():
            r = json.loads(line)
            if r['rays'] == 5000 and r['samples_per_ray'] == 64:
                v['unchanged_5000'] = '%.2f' % r['ms_per_iter']; v['torch_floor'] = '%.2f' % r['ms_per_iter_torch_floor']
    lines = [json.loads(l) for l in open(P('r06_mapping_loop.json')) if l.strip()]
    v['loop_fused'] = '%.3f' % next(l for l in lines if l['fused'])['ms_per_iteration']
    t = open(P('r06_ab_train_forward.txt')).read()
    f = lambda pat: '%.3f' % avg([float(x) for x in re.findall(pat, t)])
    v['fwd_intree'] = f(r': (\S+) ms per call \(in-tree\)'); v['fwd_nox'] = f(r': (\S+) ms per call \(\S*NOX\.so\)')
    v['fwd_noc'] = f(r': (\S+) ms per call \(\S*NOC\.so\)')
    for row in csv.reader(l for l in open(P('r06_pmc_hbm_train.csv')) if not l.startswith('#')):
        if row and 'k_decode_lc16_train' in row[0]:
            v['lc16_train_write_B'] = '%.0f' % float(row[4]); v['lc16_train_write_MB'] = '%.0f' % (float(row[2]) / 1000.0)
    # ---- host A/B (tools/host_ab.sh): the summary lines at the end of the file
    hp = P('r06_host_ab.txt')
    if os.path.exists(hp):
        rows = re.findall(r'(round \d) (render_batch_ray \(forward\)|loss\.backward\(\))\s+host cost min\s+([\d.]+) us\s+floor min\s+([\d.]+) us\s+our share\s+([\d.]+) us', open(hp).read())
        d = {(a, sec): (float(h), float(f), float(o)) for a, sec, h, f, o in rows}
        if len(d) == 4:
            f5, f6 = d[('round 5', 'render_batch_ray (forward)')], d[('round 6', 'render_batch_ray (forward)')]
            b5, b6 = d[('round 5', 'loss.backward()')], d[('round 6', 'loss.backward()')]
            v['host_ab'] = ('`render_batch_ray` forward %.0f → %.0f µs of host time (its share above the allocation-only floor %.0f → %.0f), `loss.backward()` %.0f → %.0f (share %.0f → %.0f)'
                            % (f5[0], f6[0], f5[2], f6[2], b5[0], b6[0], b5[2], b6[2]))
    s = open(template).read()
    out = re.sub(r'@@(\w+)@@', lambda m: v.get(m.group(1), m.group(0)), s)
    sys.stdout.write(out)
    left = sorted(set(re.findall(r'@@(\w+)@@', out)))
    if left:
        sys.stderr.write('unfilled: ' + ', '.join(left) + '\n')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
