#!/usr/bin/env python3
"""What the compiler left between two MFMAs of a kernel's hot loop ("commit the disassembly check").

    python scripts/isa_gaps.py file.s [kernel-name-substring] [--max-valu 4] [--max-ds 2] [--list]

Reads the gfx950 assembly of a kernel (`hipcc -S` / `--save-temps` output, or `llvm-objdump -d` text), finds the innermost loop with the
most MFMAs (a backward branch target .. the branch), and for every MFMA-to-MFMA gap of that loop counts the vector-ALU, LDS, vector-memory,
scalar and wait instructions.  Prints the histogram, every `s_waitcnt` inside the loop with the gap it sits in, and the gaps that
exceed the budget; exit code 1 if any gap does (so it can run as a check)."""
import re
import sys


def classify(op):
    if op.startswith('v_mfma') or op.startswith('v_smfmac'):
        return 'mfma'
    if op.startswith('ds_'):
        return 'ds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem'
    if op.startswith('s_waitcnt'):
        return 'wait'
    if op.startswith('s_barrier'):
        return 'barrier'
    if op.startswith('s_nop'):
        return 'nop'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


def kernels(text):
    """-> {name: [(label or None, opcode, operands)]}"""
    out, cur, name = {}, None, None
    for line in text.splitlines():
        line = line.split(';')[0].rstrip()
        if not line.strip():
            continue
        m = re.match(r'^([A-Za-z_.$][\w.$]*):', line)
        if m:
            lab = m.group(1)
            if not lab.startswith('.L') and not lab.startswith('.Lfunc'):
                name, cur = lab, []
                out[name] = cur
            elif cur is not None:
                cur.append((lab, None, None))
            continue
        s = line.strip()
        if s.startswith('.') or cur is None:
            continue
        parts = s.split(None, 1)
        cur.append((None, parts[0], parts[1] if len(parts) > 1 else ''))
    return out


def hot_loop(ins):
    labels = {lab: i for i, (lab, op, _) in enumerate(ins) if lab}
    best = None
    for i, (lab, op, args) in enumerate(ins):
        if op and op.startswith(('s_cbranch', 's_branch')):
            tgt = args.strip()
            if tgt in labels and labels[tgt] < i:
                body = ins[labels[tgt]:i + 1]
                n = sum(1 for (_, o, _) in body if o and classify(o) == 'mfma')
                inner = sum(1 for (_, o, a) in body[:-1] if o and o.startswith(('s_cbranch', 's_branch')) and a.strip() in labels
                            and labels[a.strip()] < labels[tgt] + 0)
                if best is None or n > best[0]:
                    best = (n, labels[tgt], i)
    return best


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    opts = sys.argv[1:]
    def opt(name, default):
        return int(opts[opts.index(name) + 1]) if name in opts else default
    max_valu, max_ds, max_vmem = opt('--max-valu', 4), opt('--max-ds', 2), opt('--max-vmem', 2)
    text = open(args[0]).read()
    ks = kernels(text)
    want = args[1] if len(args) > 1 else ''
    rc = 0
    for name, ins in ks.items():
        if want not in name:
            continue
        loop = hot_loop(ins)
        if not loop or loop[0] < 8:
            continue
        n, lo, hi = loop
        body = [(op, a) for (lab, op, a) in ins[lo:hi + 1] if op]
        gaps, cur = [], None
        for op, a in body:
            c = classify(op)
            if c == 'mfma':
                if cur is not None:
                    gaps.append(cur)
                cur = dict(valu=0, ds=0, vmem=0, salu=0, wait=[], nop=0, barrier=0, other=0, ops=[])
            elif cur is not None:
                if c == 'wait':
                    cur['wait'].append(a)
                else:
                    cur[c] += 1
                cur['ops'].append(op + ' ' + a)
        tot = {k: sum(g[k] for g in gaps) for k in ('valu', 'ds', 'vmem', 'salu', 'nop', 'barrier')}
        waits = [(i, w) for i, g in enumerate(gaps) for w in g['wait']]
        print('%s\n  hot loop: %d instructions, %d MFMAs, %d gaps; between MFMAs: %d VALU, %d DS, %d VMEM, %d SALU, %d s_nop, %d barriers, %d s_waitcnt'
              % (name[:110], len(body), n, len(gaps), tot['valu'], tot['ds'], tot['vmem'], tot['salu'], tot['nop'], tot['barrier'], len(waits)))
        hist = {}
        for g in gaps:
            key = (g['valu'], g['ds'], g['vmem'])
            hist[key] = hist.get(key, 0) + 1
        print('  gaps by (VALU, DS, VMEM): ' + ', '.join('%s x%d' % (k, v) for k, v in sorted(hist.items())))
        for i, w in waits:
            print('  s_waitcnt %-28s in gap %d' % (w, i))
        over = [(i, g) for i, g in enumerate(gaps) if g['valu'] > max_valu or g['ds'] > max_ds or g['vmem'] > max_vmem]
        for i, g in over:
            print('  OVER BUDGET gap %d: %d VALU, %d DS, %d VMEM' % (i, g['valu'], g['ds'], g['vmem']))
        if '--list' in opts:
            for i, g in enumerate(gaps):
                print('  gap %3d: %s' % (i, ' | '.join(g['ops'])))
        if over:
            rc = 1
    sys.exit(rc)


if __name__ == '__main__':
    main()
