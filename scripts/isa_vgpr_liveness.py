"""VGPR liveness of one kernel in a hipcc `-S` listing: how many vector registers are live at every instruction.

    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only x.hip -o x.s
    python scripts/isa_vgpr_liveness.py x.s <mangled-kernel-name-substring> [--at LINE ...] [--top N]

A development aid for the register budgets of the matrix kernels: hipcc reports only the total and
the spill count; this shows WHERE the pressure is and which registers are live there.  Conventions: first operand = def
for VALU / loads, every operand = use for stores / compares; a def under a divergent branch is treated as a kill (slight
under-estimate), partial writes (mixhi, d16_hi) as use + def."""
import re
import sys


def regs(tok):
    out = []
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1):
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.append(int(m.group(3)))
    return out


def main():
    path, name = sys.argv[1], sys.argv[2]
    at = []
    top = 12
    args = sys.argv[3:]
    while args:
        a = args.pop(0)
        if a == '--at':
            at.append(int(args.pop(0)))
        elif a == '--top':
            top = int(args.pop(0))
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if name in l and l.startswith('_Z') and l.split(';')[0].rstrip().endswith(':'))
    end = next(i for i in range(start, len(lines)) if '.amdhsa_kernel' in lines[i] or lines[i].startswith('.Lfunc_end'))
    # instructions
    ins = []          # (line_no, op, defs, uses, label_targets, falls_through)
    label_at = {}
    for i in range(start + 1, end):
        raw = lines[i]
        l = raw.split(';')[0].rstrip()
        if not l.strip():
            continue
        if not l.startswith('\t') and l.strip().endswith(':'):
            label_at[l.strip()[:-1]] = len(ins)
            continue
        l = l.strip()
        if l.startswith('.'):
            continue
        parts = l.split(None, 1)
        op = parts[0]
        ops = [x.strip() for x in parts[1].split(',')] if len(parts) > 1 else []
        defs, uses, tgt, fall = [], [], None, True
        if op.startswith('s_cbranch'):
            tgt = ops[0]
        elif op == 's_branch':
            tgt, fall = ops[0], False
        elif op in ('s_endpgm',):
            fall = False
        elif op.startswith(('scratch_store', 'ds_write', 'global_store', 'buffer_store', 'flat_store', 'v_cmp', 'v_readlane',
                            'v_readfirstlane', 's_', 'ds_add', 'global_atomic')):
            for o in ops:
                uses += regs(o)
        elif op.startswith('v_swap'):
            for o in ops:
                uses += regs(o)
                defs += regs(o)
        else:
            if ops:
                defs += regs(ops[0])
            for o in ops[1:]:
                uses += regs(o)
            if 'mixhi' in op or 'd16_hi' in op or 'op_sel:[0,0,1]' in l or op.startswith(('v_mac', 'v_fmac', 'v_pk_fmac', 'v_dot2c')):
                uses += regs(ops[0])
        ins.append((i + 1, op, set(defs), set(uses), tgt, fall))
    n = len(ins)
    succ = [[] for _ in range(n)]
    for k, (_, op, d, u, tgt, fall) in enumerate(ins):
        if fall and k + 1 < n:
            succ[k].append(k + 1)
        if tgt is not None and tgt in label_at and label_at[tgt] < n:
            succ[k].append(label_at[tgt])
    live_in = [set() for _ in range(n)]
    changed = True
    while changed:
        changed = False
        for k in range(n - 1, -1, -1):
            out = set()
            for s in succ[k]:
                out |= live_in[s]
            li = (out - ins[k][2]) | ins[k][3]
            if li != live_in[k]:
                live_in[k] = li
                changed = True
    order = sorted(range(n), key=lambda k: -len(live_in[k]))
    print(f"{n} instructions; max live VGPRs {len(live_in[order[0]])}")
    seen = []
    for k in order:
        if all(abs(ins[k][0] - s) > 40 for s in seen):
            seen.append(ins[k][0])
            print(f"  line {ins[k][0]:6d}: {len(live_in[k]):3d} live   {ins[k][1]}")
            if len(seen) >= top:
                break

    def ranges(s):
        s = sorted(s)
        out, i = [], 0
        while i < len(s):
            j = i
            while j + 1 < len(s) and s[j + 1] == s[j] + 1:
                j += 1
            out.append(f"v{s[i]}" if i == j else f"v[{s[i]}:{s[j]}]")
            i = j + 1
        return ' '.join(out)

    for ln in at:
        k = min(range(n), key=lambda k: abs(ins[k][0] - ln))
        print(f"line {ins[k][0]} ({ins[k][1]}): {len(live_in[k])} live: {ranges(live_in[k])}")


if __name__ == '__main__':
    main()
