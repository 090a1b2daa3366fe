#!/usr/bin/env python3
"""Experiments only: per-basic-block instruction mix of one kernel in a hipcc -S dump (VALU / SALU / LDS / VMEM / other).
usage: isa_blocks.py render.s k_render_bwd"""
import re, sys
path, kern = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z.*%s.*:" % kern, l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
blocks, cur = [], ["entry", 0, 0, 0, 0, 0, start, []]
for i in range(start + 1, end + 1):
    l = lines[i].strip()
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur); cur = [m.group(1), 0, 0, 0, 0, 0, i, []]; continue
    if not l or l.startswith(";") or l.startswith("."): continue
    op = l.split()[0]
    if op.startswith("v_"): cur[1] += 1
    elif op.startswith("s_"):
        cur[2] += 1
        if op.startswith("s_cbranch") or op.startswith("s_branch"): cur[7].append(l.split()[-1])
    elif op.startswith("ds_"): cur[3] += 1
    elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): cur[4] += 1
    else: cur[5] += 1
blocks.append(cur)
tot = [sum(b[k] for b in blocks) for k in range(1, 6)]
print("total valu %d salu %d lds %d vmem %d other %d blocks %d" % (*tot, len(blocks)))
for b in blocks:
    if b[1] + b[2] + b[3] + b[4] >= int(sys.argv[3]) if len(sys.argv) > 3 else 12:
        print("%-10s line %5d valu %4d salu %4d lds %3d vmem %3d -> %s" % (b[0], b[6], b[1], b[2], b[3], b[4], ",".join(b[7])))
