"""Static lint of gfx950 assembly for the one hazard hipcc cannot see in pnr_mlp_pp.h: the ping-pong MLP issues its LDS
reads as inline asm and waits for them with hand-counted `s_waitcnt lgkmcnt(N)`, so the compiler believes a read's
destination registers are valid as soon as the asm statement has "executed".  If register allocation ever inserts a copy
of such a register (or re-uses it) between the `ds_read` and the wait that covers it, the kernel computes on stale data --
silently and timing-dependently.  This walks the instruction stream in program order (branches ignored: the chunk pipeline
is straight-line), keeps the list of outstanding LDS reads exactly as the hardware counter does (in-order return), and
flags any instruction that reads or overwrites a destination that is still pending.  check() walks in program order
(straight-line code); check_cfg() follows every branch with the pending list that path really has (loops, switches).

usage: python tools/check_lds_pending.py file.s      (hipcc -S --cuda-device-only ...)"""
import re
import sys


def _regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def check(lines):
    """lines: iterable of assembly lines -> list of (line number, text, kind, line of the pending ds_read)."""
    pending, flags = [], []
    for ln, line in enumerate(lines, 1):
        s = line.split(";")[0].strip()
        if not s or s.startswith(".") or s.endswith(":"):
            continue
        parts = s.replace(",", " ").split()
        op, ops = parts[0], parts[1:]
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", s)
            if m:
                n = int(m.group(1))
                pending = pending[len(pending) - n:] if n > 0 else []
            continue
        if op.startswith("s_") or not ops:
            continue
        store = "store" in op or op.startswith("ds_write") or op.startswith("global_load_lds")
        dst = set() if store else _regs(ops[0])
        src = set().union(*[_regs(o) for o in (ops if store else ops[1:])]) if ops else set()
        for p, l in pending:
            if p & src:
                flags.append((ln, s, "reads a pending LDS destination", l))
            if p & dst:
                flags.append((ln, s, "overwrites a pending LDS destination", l))
        if op.startswith("ds_read"):
            pending.append((dst, ln))
    return flags


def _parse(lines):
    """-> (instructions [(line number, text)], {label: instruction index}, [entry instruction indices])"""
    ins, labels, entries = [], {}, []
    for ln, line in enumerate(lines, 1):
        s = line.split(";")[0].strip()
        if not s or s.startswith("."):
            if s.endswith(":"):                       # local label (.LBB0_12:)
                labels[s[:-1]] = len(ins)
            continue
        if s.endswith(":"):                           # function entry
            labels[s[:-1]] = len(ins)
            entries.append(len(ins))
            continue
        ins.append((ln, s))
    return ins, labels, entries


def check_cfg(lines):
    """Like check(), but FOLLOWS the control flow: every path through conditional branches is walked with the list of
    outstanding LDS reads it really has (k_wgrad's tile loop and shape switch are not laid out in execution order, and a
    linear walk both misses hazards across a backward branch and reports reads of other paths as pending).  States are
    (instruction, pending reads); a state seen before is not walked again, so loops terminate."""
    ins, labels, entries = _parse(lines)
    flags, seen = {}, set()
    stack = [(e, ()) for e in entries]
    while stack:
        pc, pend = stack.pop()
        pending = list(pend)
        while pc < len(ins):
            key = (pc, tuple(p[1] for p in pending))
            ln, s = ins[pc]
            parts = s.replace(",", " ").split()
            op, ops = parts[0], parts[1:]
            if op.startswith("s_cbranch") or op == "s_branch":
                if key in seen:
                    break
                seen.add(key)
                tgt = labels.get(ops[0]) if ops else None
                if op == "s_branch":
                    if tgt is None:
                        break
                    pc = tgt
                    continue
                if tgt is not None:
                    stack.append((tgt, tuple(pending)))
                pc += 1
                continue
            if op in ("s_endpgm", "s_setpc_b64", "s_swappc_b64", "s_trap"):
                break
            if op == "s_waitcnt":
                m = re.search(r"lgkmcnt\((\d+)\)", s)
                if m:
                    n = int(m.group(1))
                    pending = pending[len(pending) - n:] if n > 0 else []
                pc += 1
                continue
            if op.startswith("s_") or not ops:
                pc += 1
                continue
            store = "store" in op or op.startswith("ds_write") or op.startswith("global_load_lds")
            dst = set() if store else _regs(ops[0])
            src = set().union(*[_regs(o) for o in (ops if store else ops[1:])]) if ops else set()
            for p, l in pending:
                if p & src:
                    flags[(ln, l, "r")] = (ln, s, "reads a pending LDS destination", l)
                if p & dst:
                    flags[(ln, l, "w")] = (ln, s, "overwrites a pending LDS destination", l)
            if op.startswith("ds_read"):
                pending.append((frozenset(dst), ln))
                if len(pending) > 16:                 # lgkmcnt saturates at 15: older entries cannot be told apart any more
                    pending = pending[-16:]           # (and a loop that only issues reads would otherwise never repeat a state)
            pc += 1
    return sorted(flags.values())


if __name__ == "__main__":
    fl = check_cfg(list(open(sys.argv[1])))
    for f in fl[:40]:
        print("line %d: %s  <- %s (ds_read at line %d)" % f)
    print("flags:", len(fl))
    sys.exit(1 if fl else 0)
