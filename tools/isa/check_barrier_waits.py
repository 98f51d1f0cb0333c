"""Scan gfx9 assembly for s_barrier instructions that are reached (in straight-line order) with LDS / vector-memory operations
issued since the last matching s_waitcnt ...cnt(0). Approximate (no control-flow analysis): a report is a place to look at."""
import re, sys
def main(path):
    pend_lds = pend_vm = None
    n = 0
    for ln, raw in enumerate(open(path), 1):
        line = raw.split(';')[0].strip()
        if not line or line.startswith('.'): continue
        if line.endswith(':'): continue
        op = line.split()[0]
        if op.startswith('ds_'): pend_lds = ln
        elif op.startswith(('global_', 'flat_', 'buffer_', 'scratch_')):
            pend_vm = ln
            if op.startswith('flat_'): pend_lds = ln
        elif op == 's_waitcnt':
            if 'lgkmcnt(0)' in line: pend_lds = None
            if 'vmcnt(0)' in line: pend_vm = None
        elif op == 's_barrier':
            n += 1
            if pend_lds or pend_vm:
                print(f"{path}:{ln}: s_barrier with pending " + (f"LDS op from line {pend_lds} " if pend_lds else "") + (f"VMEM op from line {pend_vm}" if pend_vm else ""))
    print(path, "barriers:", n)
for p in sys.argv[1:]: main(p)
