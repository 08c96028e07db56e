#!/usr/bin/env python3
"""Instruction-class schedule of one kernel of a hipcc -S listing, run-length encoded:
M mfma, E v_exp, C v_cvt_pk, v other VALU, L ds_read, T ds_read_b64_tr, W ds_write, G global load, S global store,
w s_waitcnt, | s_barrier, s other SALU, B branch.   tools/isa_schedule.py listing.s kernel_name_substring"""
import re
import sys


def main():
    s = open(sys.argv[1]).read()
    m = re.search(r'^(_Z\w*%s\w*):[^\n]*\n' % re.escape(sys.argv[2]), s, re.M)
    body = s[m.end():]
    body = body[:body.index('s_endpgm')]
    lines = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith((';', '.'))]

    def cls(l):
        op = l.split()[0]
        if op.endswith(':'):
            return '\n' + op + ' '
        for p, c in (('v_mfma', 'M'), ('v_exp', 'E'), ('v_cvt_pk', 'C'), ('ds_read_b64_tr', 'T'), ('ds_read', 'L'), ('ds_', 'W'),
                     ('global_load', 'G'), ('global_store', 'S'), ('s_waitcnt', 'w'), ('s_barrier', '|'), ('s_cbranch', 'B'),
                     ('s_branch', 'B'), ('s_', 's'), ('v_', 'v')):
            if op.startswith(p):
                return c
        return '?'
    out, prev, n = [], None, 0
    for c in (cls(l) for l in lines):
        if c == prev and len(c) == 1:
            n += 1
        else:
            if prev:
                out.append(prev + (str(n) if n > 1 else ''))
            prev, n = c, 1
    out.append(prev + (str(n) if n > 1 else ''))
    print(len(lines), 'instructions')
    print(' '.join(out))


main()
