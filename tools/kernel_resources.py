"""Per-kernel register / LDS / scratch usage from `hipcc -Rpass-analysis=kernel-resource-usage` output.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage -c x.hip 2> res.txt ; python tools/kernel_resources.py res.txt [substring ...]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
pats = sys.argv[2:]
blocks = re.split(r"remark: Function Name: ", txt)[1:]
names = [b.split()[0].strip() for b in blocks]
if not names:     # (c++filt without arguments would wait for its standard input)
    sys.exit("no 'Function Name' remarks in %s" % sys.argv[1])
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True, stdin=subprocess.DEVNULL).stdout.strip().split("\n")
seen = set()
for b, dn in zip(blocks, dem):
    dn = dn.replace("(anonymous namespace)::", "").replace("void ", "")
    if dn in seen or (pats and not any(p in dn for p in pats)):
        continue
    seen.add(dn)
    d = dict(re.findall(r"remark:\s+([\w ]+?)(?: \[[\w/]+\])?: (\w+)", b))
    print("%-70s vgpr=%s sgpr=%s scratch=%s occ=%s spillV=%s" % (dn[:70], d.get("VGPRs"), d.get("TotalSGPRs"),
          d.get("ScratchSize"), d.get("Occupancy"), d.get("VGPRs Spill")))
