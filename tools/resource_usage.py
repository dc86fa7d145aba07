"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output (build log) per kernel."""
import re
import subprocess
import sys

log = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = re.split(r"remark: [^\n]*Function Name: ", log)[1:]
names = [b.split("\n")[0].strip() for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for b, dn in zip(blocks, dem):
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    dn = re.sub(r"dpfhe::", "", dn)
    dn = re.sub(r"\(.*", "", dn).replace("void ", "")
    if pat and not re.search(pat, dn):
        continue
    scratch, occ, lds = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
    print(f"{dn[:80]:80s} vgpr={g('VGPRs'):3d} agpr={g('AGPRs'):3d} sgpr={g('SGPRs'):3d} scratch={scratch:4d} occ={occ} lds={lds}")
