"""Order of memory operations, waits and arithmetic in one kernel of an assembly listing (hipcc -S --cuda-device-only):
runs of L(oad) S(tore) r/w (scratch load / store) f (f64 arithmetic) a (accvgpr moves) and every s_waitcnt vmcnt.
usage: asm_sequence.py <file.s> <mangled-name-substring> [first-line [last-line]]"""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith('_Z') and sys.argv[2] in l.split(':')[0] and l.rstrip().split(';')[0].strip().endswith(':')][0]
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
body = lines[start:end]
seq = []
for l in body:
    l = l.strip()
    if l.startswith('global_load'): k = 'L'
    elif l.startswith('global_store'): k = 'S'
    elif l.startswith('scratch_load'): k = 'r'
    elif l.startswith('scratch_store'): k = 'w'
    elif l.startswith('s_waitcnt') and 'vmcnt' in l: k = 'W(' + re.search(r'vmcnt\((\d+)\)', l).group(1) + ')'
    elif l.startswith(('v_fma_f64', 'v_mul_f64', 'v_add_f64')): k = 'f'
    elif l.startswith('v_accvgpr'): k = 'a'
    elif l.startswith('; sched_barrier') or 'sched_barrier' in l: k = '|'
    else: continue
    if seq and seq[-1][0] == k: seq[-1][1] += 1
    else: seq.append([k, 1])
print(len(body), 'lines')
print(' '.join('%s%d' % (k, n) if n > 1 else k for k, n in seq))
