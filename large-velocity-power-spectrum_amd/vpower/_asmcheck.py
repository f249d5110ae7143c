"""Static check of the NN column kernel's hand-made LDS read pipeline (csrc/nn.hip, nc_step<.., PIPE = true>).

The step requests two LDS reads in one inline-assembly statement and waits for them in another, with the step's arithmetic in
between.  Until the wait the destination registers hold nothing: the compiler must neither read, copy, spill nor overwrite
them there.  This script reads the device assembly the build keeps next to the object file (hipcc -save-temps) and verifies
that for every such request.  vpower._ffi.build() runs it on every build of nn.hip and refuses to link on a violation;
`python -m vpower._asmcheck <file.s>` does the same by hand (exit code 1 and a report)."""
import re, sys


def regs_of(text):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(x) for x in re.findall(r"\bv(\d+)\b", text))
    return out


def check(path, kernel_pat=r"nn_column_kernel"):
    lines = open(path).read().splitlines()
    n_req, bad = 0, []
    in_kernel = False
    i = 0
    while i < len(lines):
        l = lines[i]
        if re.match(r"^_Z\w*:", l):
            in_kernel = re.search(kernel_pat, l) is not None
        if in_kernel and "ds_read_b128" in l and i > 0 and "#ASMSTART" in lines[i - 1]:
            n_req += 1
            dest = regs_of(l.split(",")[0])
            nxt = lines[i + 1]
            if "ds_read_u16" not in nxt:
                bad.append((i + 1, "request is not the read pair", l))
                i += 1
                continue
            dest |= regs_of(nxt.split(",")[0])
            j = i + 2
            while j < len(lines):
                t = lines[j].split(";")[0].strip()
                if "s_waitcnt lgkmcnt(0)" in t and "#ASMSTART" in lines[j - 1]:
                    break
                if re.match(r"^\.?\w+:", t) or t.startswith("s_cbranch") or t.startswith("s_branch") or t.startswith("s_endpgm"):
                    bad.append((j + 1, "control flow between request and wait", t))
                    break
                if t and not t.startswith(";") and not t.startswith(".") and regs_of(t) & dest:
                    bad.append((j + 1, "touches v%s before the wait" % sorted(regs_of(t) & dest), t))
                j += 1
            i = j
        i += 1
    return n_req, bad


if __name__ == "__main__":
    n, bad = check(sys.argv[1])
    print("%d pipelined requests checked, %d violations" % (n, len(bad)))
    for b in bad[:20]:
        print("  line %d: %s: %s" % b)
    sys.exit(1 if bad or n == 0 else 0)
