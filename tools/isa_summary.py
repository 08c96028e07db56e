#!/usr/bin/env python3
"""Per-kernel instruction census of a hipcc -save-temps .s file: MFMAs, scratch accesses, vmcnt(0) waits, barriers,
LDS reads, LDS-DMA -- and, with --loops, the same for every loop body (label .. backward branch).
Usage: python tools/isa_summary.py file.s [--match substr] [--loops]"""
import re
import sys


def census(lines):
    c = dict(mfma=0, scratch=0, vm0=0, vmN=0, barrier=0, ds_read=0, glds=0, valu=0)
    for l in lines:
        t = l.strip()
        if t.startswith('v_mfma'): c['mfma'] += 1
        elif t.startswith('scratch_'): c['scratch'] += 1
        elif t.startswith('s_waitcnt') and 'vmcnt(0)' in t: c['vm0'] += 1
        elif t.startswith('s_waitcnt') and 'vmcnt(' in t: c['vmN'] += 1
        elif t.startswith('s_barrier'): c['barrier'] += 1
        elif t.startswith('ds_read'): c['ds_read'] += 1
        elif 'global_load_lds' in t or ('buffer_load' in t and ' lds' in t): c['glds'] += 1
        elif t.startswith('v_'): c['valu'] += 1
    return c


def main():
    path = sys.argv[1]
    match = sys.argv[sys.argv.index('--match') + 1] if '--match' in sys.argv else ''
    loops = '--loops' in sys.argv
    text = open(path).read().split('\n')
    starts = [(i, l.split(':')[0]) for i, l in enumerate(text) if re.match(r'^_Z\w+:', l)]
    for k, (i, name) in enumerate(starts):
        if match not in name:
            continue
        end = next((j for j in range(i, len(text)) if text[j].strip().startswith('s_endpgm')), len(text))
        end2 = next((j for j in range(i, len(text)) if text[j].startswith('.Lfunc_end')), len(text))
        body = text[i:end2]
        print(name, len(body), census(body))
        if loops:
            labels = {l.split(':')[0]: j for j, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l)}
            found = []
            for j, l in enumerate(body):
                m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l)
                if m and m.group(1) in labels and labels[m.group(1)] < j:
                    lo = labels[m.group(1)]
                    c = census(body[lo:j])
                    if c['mfma']:
                        found.append((j - lo, m.group(1), lo, j, c))
            for span, lab, lo, j, c in sorted(found)[:3]:   # the innermost loops that hold MFMAs
                print('   loop', lab, 'lines', lo, '..', j, c)


if __name__ == '__main__':
    main()
