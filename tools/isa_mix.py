"""Instruction mix of one kernel of libadfp (compile with `hipcc -S --cuda-device-only`, then count per class).

  python tools/isa_mix.py /tmp/isa/adfp.s k_decode_h 'ILi32ELi4ELi2E'
  python tools/isa_mix.py --loop /tmp/isa/role0.s k_decode_bwd_roles ILi4ELi2E      # the tile loop only
"""
import collections
import re
import sys


def kernels(path):
    out, name, body = {}, None, []
    for line in open(path):
        m = re.match(r'^(_Z\w+):', line)
        if m:
            name, body = m.group(1), []
            out[name] = body
        elif name and line.strip().startswith('.end_amdhsa_kernel'):
            name = None
        elif name:
            body.append(line)
    return out


def mfma_loop(body):
    """The smallest loop (label ... backward branch to it) that holds every MFMA of the kernel: the tile loop of the decoder kernels."""
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = i
    mf = [i for i, l in enumerate(body) if l.strip().startswith('v_mfma')]
    best = None
    for i, l in enumerate(body):
        m = re.search(r's_(?:cbranch_\w+|branch) (\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i and mf and labels[m.group(1)] <= mf[0] and i >= mf[-1]:
            if best is None or i - labels[m.group(1)] < best[1] - best[0]:
                best = (labels[m.group(1)], i)
    return body[best[0]:best[1] + 1] if best else body


def main():
    args = sys.argv[1:]
    loop_only = '--loop' in args
    args = [a for a in args if a != '--loop']
    path, *pats = args
    for name, body in kernels(path).items():
        if not all(p in name for p in pats):
            continue
        if loop_only:
            body = mfma_loop(body)
            name += '  [tile loop only]'
        c = collections.Counter()
        for l in body:
            t = l.strip().split()
            if not t or t[0].startswith(('.', ';', '/')) or t[0].endswith(':'):
                continue
            c[t[0]] += 1
        tot = sum(c.values())
        valu = sum(v for k, v in c.items() if k.startswith('v_') and not k.startswith('v_mfma') and not k.startswith('v_accvgpr'))
        mfma = sum(v for k, v in c.items() if k.startswith('v_mfma'))
        pk = sum(v for k, v in c.items() if k.startswith('v_pk_'))
        print(f'{name}: {tot} instr, VALU {valu} (packed {pk}), MFMA {mfma}, '
              f'LDS {sum(v for k, v in c.items() if k.startswith("ds_"))}, '
              f'VMEM {sum(v for k, v in c.items() if k.startswith(("global_", "buffer_", "flat_", "scratch_")))}, '
              f'SALU {sum(v for k, v in c.items() if k.startswith("s_"))}')
        for k, v in c.most_common(40):
            print(f'   {v:5d} {k}')


if __name__ == "__main__":
    main()
