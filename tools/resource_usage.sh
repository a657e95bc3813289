#!/bin/bash
# compact per-kernel resource table (VGPRs / AGPRs / SGPRs / scratch / LDS / occupancy) of the library, optional grep pattern on the demangled name
pat=${1:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result -Rpass-analysis=kernel-resource-usage \
  "$(dirname "$0")/../vispec_amd/csrc/vispec_hip.hip" -o /tmp/ru_x.so 2>&1 | python3 -c '
import re, sys, subprocess
cur = None; rows = []
for ln in sys.stdin:
    m = re.search(r"Function Name: (\S+)", ln)
    if m: cur = {"name": m.group(1)}; rows.append(cur); continue
    for k in ("VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize \[bytes/lane\]", "Occupancy \[waves/SIMD\]", "LDS Size \[bytes/block\]"):
        m = re.search(r"\s" + k + r": (\d+)", ln)
        if m and cur is not None: cur[k.split(" ")[0]] = m.group(1)
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.split("\n")
for r, n in zip(rows, names):
    if re.search(sys.argv[1], n):
        print("%-90s v%-4s a%-4s s%-4s scr%-4s lds%-7s occ%s" % (n.split("(")[0][:90], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize"), r.get("LDS"), r.get("Occupancy")))
' "$pat"
