"""Control-flow-aware check of gfx9 assembly (hipcc --cuda-device-only -S): which s_barrier can be REACHED -- along any path, loop
back edges included -- with an LDS store / atomic (or a vector-memory store) issued and no s_waitcnt lgkmcnt(0) (vmcnt(0)) since?
hipcc 7.2 drops the workgroup-release `s_waitcnt lgkmcnt(0)` in front of a barrier at a loop header when the pending LDS writes arrive
over the back edge only (found in the simulator's bitonic pair sort); on MI355X such a barrier does not order the LDS
between waves. Usage: python tools/isa/check_barrier_cfg.py file.s [--vm] (--vm also reports pending global stores, which LLVM leaves
pending by design at workgroup scope)."""
import re
import sys


def parse(path):
    funcs, cur, name = {}, None, None
    for raw in open(path):
        line = raw.split(';')[0].rstrip()
        m = re.match(r'^([A-Za-z_.$][\w.$]*):', line)
        if m and not m.group(1).startswith('.L'):
            name = m.group(1)
            cur = funcs.setdefault(name, [])
            continue
        if cur is None:
            continue
        t = line.strip()
        if not t or (t.startswith('.') and not t.endswith(':')):
            continue
        cur.append(t)
    return funcs


def analyse(name, ins, want_vm):
    # basic blocks
    leaders = {0}
    labels = {}
    for i, t in enumerate(ins):
        if t.endswith(':'):
            labels[t[:-1]] = i
            leaders.add(i)
        op = t.split()[0]
        if op.startswith(('s_branch', 's_cbranch', 's_endpgm', 's_setpc', 's_swappc')):
            leaders.add(i + 1)
    leaders = sorted(l for l in leaders if l < len(ins))
    block_of = {}
    blocks = []
    for bi, l in enumerate(leaders):
        end = leaders[bi + 1] if bi + 1 < len(leaders) else len(ins)
        blocks.append((l, end))
        for i in range(l, end):
            block_of[i] = bi
    succ = [[] for _ in blocks]
    for bi, (l, e) in enumerate(blocks):
        last = ins[e - 1].split()
        op = last[0]
        if op == 's_branch':
            tgt = labels.get(last[1])
            if tgt is not None:
                succ[bi].append(block_of[tgt])
        elif op.startswith('s_cbranch'):
            tgt = labels.get(last[1])
            if tgt is not None:
                succ[bi].append(block_of[tgt])
            if bi + 1 < len(blocks):
                succ[bi].append(bi + 1)
        elif op in ('s_endpgm', 's_setpc_b64'):
            pass
        elif bi + 1 < len(blocks):
            succ[bi].append(bi + 1)

    def transfer(state, l, e, report):
        lds, vm = state
        for i in range(l, e):
            t = ins[i]
            op = t.split()[0]
            if op.startswith('ds_') and not op.startswith(('ds_read', 'ds_bpermute', 'ds_permute', 'ds_swizzle', 'ds_nop')):
                lds = i
            elif op.startswith(('global_store', 'flat_store', 'buffer_store', 'global_atomic', 'flat_atomic')):
                vm = i
                if op.startswith('flat_'):
                    lds = i
            elif op == 's_waitcnt':
                if 'lgkmcnt(0)' in t:
                    lds = None
                if 'vmcnt(0)' in t:
                    vm = None
            elif op == 's_barrier' and report is not None:
                if lds is not None:
                    report.append((i, 'LDS', lds))
                if want_vm and vm is not None:
                    report.append((i, 'VMEM', vm))
        return (lds, vm)

    inn = [(None, None)] * len(blocks)
    out = [(None, None)] * len(blocks)
    work = list(range(len(blocks)))
    merge = lambda a, b: (a[0] if a[0] is not None else b[0], a[1] if a[1] is not None else b[1])
    it = 0
    while work and it < 200000:
        it += 1
        bi = work.pop(0)
        l, e = blocks[bi]
        o = transfer(inn[bi], l, e, None)
        if o != out[bi] or it <= len(blocks):
            out[bi] = o
            for s in succ[bi]:
                m = merge(inn[s], o)
                if m != inn[s]:
                    inn[s] = m
                    if s not in work:
                        work.append(s)
    rep = []
    for bi, (l, e) in enumerate(blocks):
        transfer(inn[bi], l, e, rep)
    return rep, ins


def main():
    want_vm = '--vm' in sys.argv
    total = 0
    for path in [a for a in sys.argv[1:] if not a.startswith('--')]:
        for name, ins in parse(path).items():
            if not any(t.split()[0] == 's_barrier' for t in ins):
                continue
            rep, ins = analyse(name, ins, want_vm)
            seen = set()
            for (i, kind, src) in rep:
                if (i, kind) in seen:
                    continue
                seen.add((i, kind))
                total += 1
                print(f"{path}: {name[:60]}: s_barrier (instr {i}) reachable with pending {kind} op: '{ins[src]}'")
    print("barriers reachable with pending operations:", total)


if __name__ == '__main__':
    main()
