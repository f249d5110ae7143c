"""Instruction mix of one kernel of a hipcc -S dump: python tools/isa_mix.py file.s <mangled-name-substring> [start-line end-line]"""
import re, sys, collections
lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and pat in l and l.rstrip().endswith(":") or (l.startswith("_Z") and pat in l and ": " in l))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
if len(sys.argv) > 4: start, end = int(sys.argv[3]), int(sys.argv[4])
c = collections.Counter()
for l in lines[start:end]:
    m = re.match(r"\t([a-z_0-9]+)", l)
    if not m: continue
    op = m.group(1)
    if op.startswith("v_pk_"): k = "v_pk"
    elif op.startswith("v_"): k = "valu:" + op
    elif op.startswith("ds_"): k = op
    elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("scratch_"): k = op
    elif op.startswith("s_waitcnt") or op.startswith("s_barrier"): k = op
    elif op.startswith("s_"): k = "salu"
    else: k = op
    c[k] += 1
tot_v = sum(v for k, v in c.items() if k.startswith("valu") or k == "v_pk")
print("lines %d..%d  VALU total %d (pk %d)" % (start, end, tot_v, c["v_pk"]))
for k, v in c.most_common(45): print("  %-28s %d" % (k, v))
