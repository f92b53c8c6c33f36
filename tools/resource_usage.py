#!/usr/bin/env python3
"""Print VGPR / scratch / occupancy per kernel from hipcc -Rpass-analysis=kernel-resource-usage output."""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
rows, cur = [], {}
for line in txt.splitlines():
    m = re.search(r"remark: (?:\S+ )?\s*(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs|LDS Size \[bytes/block\]): (.*?)(?: \[-Rpass|$)", line)
    if not m: continue
    k, v = m.group(1), m.group(2).strip()
    if k == "Function Name":
        if cur: rows.append(cur)
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
    else:
        cur[k.split(" ")[0]] = v
if cur: rows.append(cur)
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", r["name"]); n = re.sub(r"\(.*", "", n).replace("void ", "")
    print("%-34s vgpr=%-4s sgpr=%-4s scratch=%-5s occ=%-2s lds=%s" % (n, r.get("VGPRs"), r.get("SGPRs"), r.get("ScratchSize"), r.get("Occupancy"), r.get("LDS")))
