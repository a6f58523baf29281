"""Condensed instruction-class trace of a kernel's ISA (tuning aid, build container only).

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only igemm16.hip -o /tmp/k.s
  python tools/isa_trace.py /tmp/k.s <kernel-substring> [--min-mfma N]

M mfma, r ds_read, W ds_write, G buffer_load, g global_load, S store, v VALU, s SALU, [..] s_waitcnt, |B| barrier.
Prints the basic blocks that contain MFMAs (and their neighbours) so the load/convert/store/MFMA order is visible.
"""
import re
import sys


def cls(l):
    l = l.strip()
    if not l or l.startswith(';') or l.startswith('.'):
        return None
    op = l.split()[0]
    if op.startswith('v_mfma'): return 'M'
    if op.startswith('ds_read') or op.startswith('ds_load'): return 'r'
    if op.startswith('ds_write') or op.startswith('ds_store'): return 'W'
    if op.startswith('buffer_load'): return 'G'
    if op.startswith('global_load'): return 'g'
    if op.startswith('global_store') or op.startswith('buffer_store'): return 'S'
    if op.startswith('global_atomic'): return 'A'
    if op == 's_waitcnt': return '[' + l.split(None, 1)[1].replace(' ', '') + ']'
    if op == 's_barrier': return '|B|'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return '<' + op[2:] + ' ' + l.split()[1] + '>'
    if op.startswith('v_'): return 'v'
    if op.startswith('s_'): return 's'
    return '?'


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split('\n')
    start = [i for i, l in enumerate(lines) if re.match(r'^_Z\w*:', l) and key in l]
    if not start:
        sys.exit('kernel not found')
    i0 = start[0]
    i1 = next(i for i in range(i0, len(lines)) if lines[i].strip().startswith('s_endpgm'))
    body = lines[i0:i1]
    blocks, cur = [], ['entry', []]
    for l in body:
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            blocks.append(cur)
            cur = [m.group(1), []]
        else:
            c = cls(l)
            if c:
                cur[1].append(c)
    blocks.append(cur)
    has = [i for i, b in enumerate(blocks) if 'M' in b[1]]
    show = set()
    for i in has:
        show.update(range(max(0, i - 3), min(len(blocks), i + 3)))
    for i in sorted(show):
        print(blocks[i][0] + ': ' + ''.join(blocks[i][1]))
    for l in lines[i1:i1 + 400]:
        if re.search(r'(NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize|SGPRBlocks|NumSgprs)', l):
            print(l.strip())
        if re.match(r'^_Z\w*:', l):
            break


if __name__ == '__main__':
    main()
