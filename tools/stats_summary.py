"""Condense the comparison logs the GPU suite writes when ADFP_GRAD_STATS / ADFP_PARITY_STATS name a file
(tests/conftest.py: assert_grad_tight, assert_param_grad_close, assert_close) into the summaries kept under profiles/.

    python tools/stats_summary.py grad   <raw log> > profiles/r05_grad_stats.txt
    python tools/stats_summary.py parity <raw log> > profiles/r05_parity_stats.txt"""
import re
import statistics
import sys


def grad(path):
    tight, loose = {}, {}
    for line in open(path):
        t = line.split()
        if not t:
            continue
        if ' swapped ' in line:                               # the negative control of tests/test_gpu_grad.py (0.89 x scale by design)
            continue
        if t[0] == 'tight':
            tight.setdefault(t[1], []).append((float(t[-1]), line.strip()))
        else:
            m = re.search(r'max (\S+) fro (\S+) rows_bad (\d+)/(\d+)', line)
            if m:
                loose.setdefault(t[0], []).append((float(m.group(1)), float(m.group(2)), int(m.group(3)), line.strip()))
    print('# tests/ on the MI355X with ADFP_GRAD_STATS=<file>: every gradient comparison of the GPU suite, summarised by tools/stats_summary.py')
    print('# "tight" = conftest.assert_grad_tight: kernels vs the oracle\'s autograd along the ReLU decisions the kernels\' backward took (forced in the oracle);')
    print('# "loose" = conftest.assert_param_grad_close: kernels vs the reference\'s own (unforced) autograd gradients (tests/golden/mini_<stage>.npz / the unforced oracle)')
    for mode, v in sorted(tight.items()):
        e = [x[0] for x in v]
        print(f'tight, {mode}: {len(v)} tensors, worst element {max(e):.2e} x scale, median of the per-tensor worst {statistics.median(e):.2e}; limit 5e-05 on every element')
    for mode, v in sorted(loose.items()):
        print(f'loose, {mode}: {len(v)} parameter tensors, worst element {max(x[0] for x in v):.2e} x scale, worst relative Frobenius error '
              f'{max(x[1] for x in v):.2e}, tensors with any row beyond 2e-4 x scale: {sum(1 for x in v if x[2])}; limits 1e-3 / 1e-3')
    print('\nten worst tight comparisons:')
    for e, line in sorted((x for v in tight.values() for x in v), reverse=True)[:10]:
        print('  ' + line)
    print('\nten worst loose comparisons:')
    for x in sorted((x for v in loose.values() for x in v), reverse=True)[:10]:
        print('  ' + x[3])


def parity(path):
    rows = []
    for line in open(path):
        m = re.search(r'\| tol (\S+) worst (\S+) of the limit', line)
        if m:
            rows.append((float(m.group(2)), float(m.group(1)), line.strip()))
    print('# tests/ on the MI355X with ADFP_PARITY_STATS=<file>: every conftest.assert_close comparison of the GPU suite (kernels vs the CPU oracle / the')
    print('# golden vectors), summarised by tools/stats_summary.py.  Criterion, element by element: |a - b| <= tol * max(|b|, 1e-2 * max|b|);')
    print('# "worst" is the largest |a - b| / limit of the tensor (1.0 = at the limit).')
    by_tol = {}
    for w, tol, _ in rows:
        by_tol.setdefault(tol, []).append(w)
    for tol, v in sorted(by_tol.items()):
        print(f'tol {tol:g}: {len(v)} tensors, worst {max(v):.3f} of the limit, median {statistics.median(v):.4f}, tensors beyond half the limit: {sum(1 for x in v if x > 0.5)}')
    print('\ntwenty comparisons closest to their limit:')
    for w, tol, line in sorted(rows, reverse=True)[:20]:
        print('  ' + line)


if __name__ == '__main__':
    (grad if sys.argv[1] == 'grad' else parity)(sys.argv[2])
