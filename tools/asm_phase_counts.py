"""Static instruction counts of a kernel's .s file, split at s_memtime stamps (phase boundaries)."""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith("_ZN") and "kernel" in l and ":" in l][0]
end = [i for i, l in enumerate(lines) if i > start and "s_endpgm" in l][-1]
seg = 0
counts = collections.defaultdict(collections.Counter)
for l in lines[start:end]:
    m = re.match(r"\s+([a-z_0-9]+)", l)
    if not m:
        continue
    op = m.group(1)
    if op == "s_memtime":
        seg += 1
        continue
    cls = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_")
           else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
    counts[seg][cls] += 1
    counts[seg]["op:" + op] += 1
show = set(int(x) for x in sys.argv[2:]) if len(sys.argv) > 2 else set()
for sgi in sorted(counts):
    c = counts[sgi]
    print(sgi, {k: v for k, v in c.items() if not k.startswith("op:")})
    if sgi in show:
        print("   ", [(k[3:], v) for k, v in c.most_common(32) if k.startswith("op:")])
