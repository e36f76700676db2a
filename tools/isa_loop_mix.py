#!/usr/bin/env python3
"""Instruction mix of the hottest loop of a kernel in a hipcc -S dump (tools: budget kernels by instruction count).
Usage: isa_loop_mix.py file.s substring [substring...]   (substring of the mangled kernel name)"""
import re, sys, collections

def classify(op):
    if op.startswith('v_mfma') or op.startswith('v_smfma'): return 'MFMA'
    if op.startswith('ds_'): return 'LDS'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'VMEM'
    if op.startswith('s_waitcnt'): return 'WAIT'
    if op.startswith(('s_load', 's_buffer_load')): return 'SMEM'
    if op.startswith(('s_nop', 's_sleep')): return 'NOP'
    if op.startswith(('s_cbranch', 's_branch', 's_barrier', 's_setprio', 's_endpgm')): return 'CTRL'
    if op.startswith('s_'): return 'SALU'
    if op.startswith('v_accvgpr'): return 'ACCMOV'
    if op.startswith('v_'): return 'VALU'
    return 'OTHER'

def kernels(path):
    cur, body, out = None, [], {}
    for ln in open(path):
        m = re.match(r'^(_Z[\w]+):', ln)
        if m and cur is None:
            cur, body = m.group(1), []
            continue
        if cur is not None:
            body.append(ln.rstrip('\n'))
            if 's_endpgm' in ln:
                out[cur] = body; cur = None
    return out

def loops(body):
    labels = {}
    for i, ln in enumerate(body):
        m = re.match(r'^(\.LBB[\w]+):', ln)
        if m: labels[m.group(1)] = i
    res = []
    for i, ln in enumerate(body):
        m = re.match(r'\s+s_cbranch_\w+\s+(\.LBB[\w]+)', ln) or re.match(r'\s+s_branch\s+(\.LBB[\w]+)', ln)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            res.append((labels[m.group(1)], i))
    return res

def mix(lines):
    c = collections.Counter(); ops = collections.Counter()
    for ln in lines:
        m = re.match(r'\s+([a-z_0-9]+)', ln)
        if not m or ln.lstrip().startswith(('.', ';')): continue
        op = m.group(1); c[classify(op)] += 1; ops[op] += 1
    return c, ops

if __name__ == '__main__':
    ks = kernels(sys.argv[1])
    for name, body in ks.items():
        if not all(s in name for s in sys.argv[2:]): continue
        best = None
        for a, b in loops(body):
            c, ops = mix(body[a:b + 1])
            if best is None or c['MFMA'] > best[0]['MFMA'] or (c['MFMA'] == best[0]['MFMA'] and b - a < best[3] - best[2]): best = (c, ops, a, b)
        if best is None: print(name[:110], 'no loop'); continue
        c, ops, a, b = best
        print(name[:140]); print('   loop lines %d..%d  ' % (a, b) + '  '.join('%s %d' % kv for kv in sorted(c.items())))
        valu = [(o, n) for o, n in ops.most_common() if classify(o) in ('VALU', 'ACCMOV')]
        print('   VALU: ' + '  '.join('%s %d' % kv for kv in valu[:14]))
