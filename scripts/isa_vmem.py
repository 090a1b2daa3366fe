#!/usr/bin/env python3
"""Experiments only: VMEM instructions, waitcnts on vmcnt and labels of one kernel in a hipcc -S dump, in program order.
usage: isa_vmem.py render.s <kernel name substring>"""
import re, sys
s = open(sys.argv[1]).read().split('\n')
start = next(i for i, l in enumerate(s) if sys.argv[2] in l and l.startswith('_Z'))
end = next(i for i in range(start, len(s)) if 's_endpgm' in s[i])
for i in range(start, end):
    l = s[i].strip()
    if re.match(r'^(global_|buffer_|flat_)', l) or ('s_waitcnt' in l and 'vmcnt' in l):
        print(i - start, l[:100])
