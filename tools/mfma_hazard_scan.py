"""Scan gfx950 assembly (hipcc -S --cuda-device-only) for the pattern behind the wrong rows found in k_sweep_rl (DESIGN.md
section 3a''): an MFMA whose result registers are read by a VALU / LDS / VMEM instruction within a few instructions and with
fewer than NEED wait states (s_nop, other instructions) in between -- looking through unconditional branches and into both
successors of conditional ones.  hipcc normally inserts the s_nop; where an MFMA chain ended a conditional block it put
the s_nop behind the first reads.  usage: mfma_hazard_scan.py file.s [NEED=10]

Second check: 8-byte loads with sc1 (`*_load_dwordx2 ... sc1`).  Such loads of data another workgroup had
published returned the PREVIOUS launch's bytes from the XCD's L2 in k_sweep_rl (DESIGN.md section 3a'', hand-over lesson 1;
the guide's table of validated hand-offs lists dword and dwordx4 only): hand-over data is read with 16-byte sc1 loads, flags
with 4-byte ones, and no kernel of the library may contain the 8-byte form.  Reported on the line before the last."""
import re, sys

path = sys.argv[1]
NEED = int(sys.argv[2]) if len(sys.argv) > 2 else 10
lines = [l.rstrip() for l in open(path)]
label_at = {}
code = []          # (text, srcline)
for n, l in enumerate(lines):
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            label_at[m.group(1)] = len(code)
        if t.startswith(".") and not re.match(r"^\.LBB", t):
            continue
        continue
    m = re.match(r"^(\w+):", t)
    if m and not t.startswith("s_") and not t.startswith("v_"):
        label_at[m.group(1)] = len(code)
        continue
    code.append((t.split(";")[0].strip(), n + 1))

def regs(tok):
    m = re.match(r"^([av])\[(\d+):(\d+)\]$", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"^([av])(\d+)$", tok)
    return {(m.group(1), int(m.group(2)))} if m else set()

def operands(text):
    parts = text.split(None, 1)
    if len(parts) < 2:
        return []
    return [p.strip() for p in parts[1].split(",")]

hits = 0
for i, (t, ln) in enumerate(code):
    if not t.startswith("v_mfma_"):
        continue
    dst = regs(operands(t)[0])
    # successive MFMAs into the same registers are interlocked by hardware: start at the last of a chain
    if i + 1 < len(code) and code[i + 1][0].startswith("v_mfma_") and regs(operands(code[i + 1][0])[0]) == dst:
        continue
    work = [(i + 1, 0)]
    seen = set()
    while work:
        k, ws = work.pop()
        while k < len(code) and ws < NEED and (k, ws) not in seen:
            seen.add((k, ws))
            u = code[k][0]
            op = u.split()[0]
            if op == "s_nop":
                ws += int(u.split()[1]) + 1
            elif op == "s_branch":
                k = label_at.get(u.split()[1], len(code))
                ws += 1
                continue
            elif op.startswith("s_cbranch"):
                work.append((label_at.get(u.split()[1], len(code)), ws + 1))
                ws += 1
            elif op in ("s_endpgm", "s_barrier", "s_waitcnt") or op.startswith("v_mfma_"):
                ws += 1 if op != "s_barrier" else NEED
                if op.startswith("v_mfma_"):
                    break      # MFMA -> MFMA dependencies are interlocked
            else:
                ops = operands(u)
                srcs = ops[1:] if (op.startswith("v_") or op.startswith("ds_read") or "load" in op) else ops
                if op.startswith("ds_write") or "store" in op:
                    srcs = ops
                used = set().union(*[regs(o) for o in srcs]) if srcs else set()
                if used & dst:
                    print(f"{path}:{code[k][1]}: '{u}' reads the result of the MFMA at line {ln} after {ws} wait states")
                    hits += 1
                    break
                if op.startswith("v_") and ops and regs(ops[0]) & dst:
                    break      # overwritten
                ws += 1
            k += 1
bad8 = [(t, ln) for t, ln in code if re.match(r"^(global|buffer|flat)_load_dwordx2\b", t) and re.search(r"\bsc1\b", t)]
for t, ln in bad8:
    print(f"{path}:{ln}: 8-byte sc1 load '{t}'")
print(f"{len(bad8)} 8-byte sc1 loads")
print(f"{hits} suspicious reads (fewer than {NEED} wait states behind an MFMA)")
