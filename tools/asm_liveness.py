#!/usr/bin/env python3
"""Approximate VGPR liveness profile of a kernel's .s file: linear scan, a register is live from a write
to its last read before the next write (control flow ignored -- good enough to see which phase of a mostly
straight-line kernel sets the register budget).  Prints the number of live VGPRs every STEP instructions
with markers for s_barrier / s_memtime.
    python tools/asm_liveness.py file.s [step]"""
import re
import sys

path = sys.argv[1]
step = int(sys.argv[2]) if len(sys.argv) > 2 else 250
lines = open(path).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith("_ZN") and "kernel" in l and ":" in l][0]
end = [i for i, l in enumerate(lines) if i > start and "s_endpgm" in l][-1]
NODEF = ("ds_write", "global_store", "scratch_store", "buffer_store", "v_cmp", "v_cmpx", "s_", "ds_bpermute_nodef",
         "global_atomic", "v_readlane", "v_readfirstlane", "ds_swizzle_nodef")
ALSO_USE = ("v_fmac", "v_mac", "v_dot2c", "v_writelane", "v_cndmask", "_dpp", "v_mov_b32_dpp")


def regs(tok):
    out = []
    for a, b in re.findall(r"v\[(\d+):(\d+)\]", tok):
        out += list(range(int(a), int(b) + 1))
    tok2 = re.sub(r"v\[\d+:\d+\]", "", tok)
    out += [int(x) for x in re.findall(r"\bv(\d+)\b", tok2)]
    return out


ins = []
marks = {}
for l in lines[start:end]:
    l = l.split(";")[0]
    m = re.match(r"\s+([a-z_0-9]+)\s*(.*)", l)
    if not m:
        continue
    op, rest = m.group(1), m.group(2)
    if op in ("s_barrier", "s_memtime"):
        marks[len(ins)] = marks.get(len(ins), "") + " " + op
    ops = [t.strip() for t in rest.split(",")] if rest else []
    defs, uses = [], []
    if ops:
        if op.startswith(NODEF) or not op.startswith(("v_", "ds_", "global_", "scratch_", "buffer_", "flat_")):
            for t in ops:
                uses += regs(t)
        else:
            defs = regs(ops[0])
            for t in ops[1:]:
                uses += regs(t)
            if any(k in op for k in ALSO_USE):
                uses += defs
            if op.startswith(("v_mad_u64", "v_mad_i64", "v_add_co", "v_sub_co", "v_addc", "v_subb", "v_div_scale")) and len(ops) > 1:
                pass
    ins.append((op, defs, uses))
N = len(ins)
live_until = {}          # reg -> last use index of the current value
intervals = []
cur_def = {}
for i, (op, defs, uses) in enumerate(ins):
    for r in uses:
        if r in cur_def:
            live_until[r] = i
    for r in defs:
        if r in cur_def:
            intervals.append((cur_def[r], live_until.get(r, cur_def[r])))
        cur_def[r] = i
        live_until[r] = i
for r, d in cur_def.items():
    intervals.append((d, live_until.get(r, d)))
delta = [0] * (N + 2)
for a, b in intervals:
    delta[a] += 1
    delta[b + 1] -= 1
live = []
c = 0
for i in range(N):
    c += delta[i]
    live.append(c)
print(f"{N} instructions, peak {max(live)} live VGPRs at instr {live.index(max(live))}")
for i in range(0, N, step):
    seg = live[i:i + step]
    mk = "".join(f" [{k}:{v.strip()}]" for k, v in marks.items() if i <= k < i + step)
    print(f"{i:6d} max {max(seg):4d} min {min(seg):4d}{mk}")

if len(sys.argv) > 3:
    at = int(sys.argv[3])
    print(f"live across instr {at} (def -> last use), longest first:")
    rows = [(b - a, a, b) for a, b in intervals if a <= at <= b]
    rows.sort(reverse=True)
    for ln, a, b in rows:
        print(f"   def {a:6d} {ins[a][0]:24s} -> last use {b:6d} {ins[b][0]}")
