"""Control-flow outline of one kernel of a hipcc -S dump with instruction counts per straight-line stretch:
python tools/isa_outline.py file.s <mangled-name-prefix> [min-instructions-to-print]"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
minp = int(sys.argv[3]) if len(sys.argv) > 3 else 0
s = next(i for i, l in enumerate(lines) if l.startswith(pat))
e = next(i for i in range(s + 1, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
keys = ("v", "pk", "mov", "ds", "gl", "s", "scr")
cnt = {k: 0 for k in keys}
tot = {k: 0 for k in keys}
def flush(tag):
    global cnt
    n = sum(cnt[k] for k in ("v", "ds", "gl", "s", "scr"))
    if n >= minp:
        print("%-44s VALU=%-4d pk=%-4d mov=%-3d ds=%-3d glob=%-3d salu=%-3d scratch=%d" % (tag, cnt["v"], cnt["pk"], cnt["mov"], cnt["ds"], cnt["gl"], cnt["s"], cnt["scr"]))
    for k in keys: tot[k] += cnt[k]
    cnt = {k: 0 for k in keys}
for i in range(s, e):
    l = lines[i]
    m = re.match(r"\t([a-z_0-9]+)", l)
    if re.match(r"\.LBB\d+_\d+:", l):
        flush("-- " + l.split()[0] + " @%d" % i); continue
    if not m: continue
    op = m.group(1)
    if op.startswith("s_cbranch") or op == "s_branch" or op == "s_barrier":
        flush("%s %s @%d" % (op, l.split()[-1] if "branch" in op else "", i)); continue
    if op.startswith("v_"):
        cnt["v"] += 1
        if op.startswith("v_pk"): cnt["pk"] += 1
        if op.startswith("v_mov"): cnt["mov"] += 1
    elif op.startswith("ds_"): cnt["ds"] += 1
    elif op.startswith("global_"): cnt["gl"] += 1
    elif op.startswith("scratch_"): cnt["scr"] += 1
    elif op.startswith("s_"): cnt["s"] += 1
flush("end")
print("TOTAL", tot)
