"""Compressed load / MFMA / waitcnt schedule of one kernel in a hipcc -S listing:  python tools/isa_sched.py file.s mangled_name_substring"""
import re, sys
s = open(sys.argv[1]).read()
m = re.search(r'^(\S*' + re.escape(sys.argv[2]) + r'\S*):(.*?)s_endpgm', s, re.S | re.M)
print(m.group(1))
seq = []
for l in m.group(2).splitlines():
    l = l.strip()
    if l.startswith(('global_load', 'buffer_load')): seq.append('L')
    elif l.startswith('v_mfma'): seq.append('M')
    elif l.startswith('ds_read') or l.startswith('ds_load'): seq.append('r')
    elif l.startswith('ds_write') or l.startswith('ds_store'): seq.append('s')
    elif l.startswith('global_store') or l.startswith('buffer_store'): seq.append('S')
    elif l.startswith('s_barrier'): seq.append('B')
    elif l.startswith('s_waitcnt') and 'vmcnt' in l: seq.append('w' + re.search(r'vmcnt\((\d+)\)', l).group(1))
    elif l.startswith(('s_cbranch', 's_branch')): seq.append('|')
    elif l.startswith('.LBB'): seq.append('#')
out, prev, cnt = [], None, 0
for x in seq + [None]:
    if x == prev: cnt += 1
    else:
        if prev: out.append(f"{prev}x{cnt}" if cnt > 1 else prev)
        prev, cnt = x, 1
print(' '.join(out))
