"""VALU instructions between consecutive MFMAs of one kernel (from `hipcc -S`): how well the compiler filled the gaps.
  python tools/isa_gaps.py /tmp/isa/adfp.s k_decode_p ILi32ELi4ELi2E"""
import re
import sys

path, *pats = sys.argv[1:]
name, body, out = None, [], {}
for line in open(path):
    m = re.match(r'^(_Z\w+):', line)
    if m:
        name, body = m.group(1), []
        out[name] = body
    elif name and line.strip().startswith('.end_amdhsa_kernel'):
        name = None
    elif name:
        body.append(line.strip())
for name, body in out.items():
    if not all(p in name for p in pats):
        continue
    gaps, cur, seen, tr = [], 0, False, 0
    for l in body:
        t = l.split()
        if not t:
            continue
        op = t[0]
        if op.startswith('v_mfma'):
            if seen:
                gaps.append(cur)
            seen, cur = True, 0
        elif op.startswith('v_') and not op.startswith('v_accvgpr'):
            cur += 2 if op in ('v_sin_f32_e32', 'v_cos_f32_e32', 'v_exp_f32_e32', 'v_rcp_f32_e32', 'v_log_f32_e32') else 1
        elif op.startswith(('s_cbranch', 's_branch')) and seen:
            gaps.append(-1)
    print(name, 'MFMAs', len([g for g in gaps if g >= 0]) + 1)
    print(' '.join('|' if g < 0 else str(g) for g in gaps))
    vals = [g for g in gaps if g >= 0]
    print('mean %.1f  <=5: %d  6-12: %d  13-24: %d  >24: %d' % (sum(vals) / max(1, len(vals)), sum(g <= 5 for g in vals),
                                                              sum(5 < g <= 12 for g in vals), sum(12 < g <= 24 for g in vals), sum(g > 24 for g in vals)))
