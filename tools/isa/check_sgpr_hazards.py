"""Scan gfx950 assembly (hipcc -save-temps *.s) for two manually-managed hazards of the CDNA ISA:
  (a) VALU writes an SGPR  ->  VMEM / FLAT / scratch instruction reads that SGPR  : 5 wait states
  (b) VALU writes an SGPR  ->  v_readlane / v_writelane uses it as lane select     : 4 wait states
Wait states are counted as instructions issued in between (s_nop N counts N + 1). Labels reset nothing: a
fall-through path keeps its history, which is the conservative reading. Prints every violation found."""
import re, sys

def regs(tok):
    tok = tok.strip()
    m = re.fullmatch(r's\[(\d+):(\d+)\]', tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r's(\d+)', tok)
    if m: return {int(m.group(1))}
    if tok == 'vcc': return {'vcc'}
    return set()

def main(path):
    hist = []   # (line_no, text, sgprs_written_by_valu)
    viol = 0
    for ln, raw in enumerate(open(path), 1):
        line = raw.split(';')[0].strip()
        if not line or line.startswith('.') or line.endswith(':'):
            continue
        parts = line.split(None, 1)
        op = parts[0]
        ops = [o.strip() for o in parts[1].split(',')] if len(parts) > 1 else []
        ws = 1
        if op == 's_nop':
            ws = int(ops[0], 0) + 1
        written = set()
        if op.startswith('v_'):
            if op.startswith(('v_readlane', 'v_readfirstlane')):
                written |= regs(ops[0])
            elif op.startswith('v_cmp') and '_e64' in op:
                written |= regs(ops[0])
            elif op.startswith('v_cmp'):
                written |= {'vcc'}
            elif len(ops) > 1 and re.match(r'v_(add|sub|subrev|addc|subb|subbrev)_co', op):
                written |= regs(ops[1])
            elif op.startswith(('v_div_scale', 'v_mad_u64_u32', 'v_mad_i64_i32')) and len(ops) > 1:
                written |= regs(ops[1])
        is_vmem = op.startswith(('global_', 'flat_', 'scratch_', 'buffer_'))
        is_lane = op.startswith(('v_readlane', 'v_writelane'))
        need = None
        used = set()
        if is_vmem:
            need = 5
            for o in ops:
                for t in o.split():
                    used |= regs(t)
        elif is_lane and len(ops) >= 3:
            need = 4
            used |= regs(ops[2])
        if need:
            dist = 0
            for (hl, ht, hw, hws) in reversed(hist):
                if dist >= need: break
                if hw & used:
                    print(f"{path}:{ln}: '{line}' reads {sorted(hw & used, key=str)} written by VALU at line {hl} '{ht}' only {dist} wait states earlier (needs {need})")
                    viol += 1
                    break
                dist += hws
        hist.append((ln, line, written, ws))
        if len(hist) > 16: hist.pop(0)
    print(path, "violations:", viol)

if __name__ == '__main__':
    for p in sys.argv[1:]: main(p)
