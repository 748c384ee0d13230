"""Summarise `make -C alphasurf_amd/csrc resource-usage` output: kernel, VGPRs, scratch, occupancy."""
import re, subprocess, sys
out = subprocess.run(["make", "-C", "alphasurf_amd/csrc", "resource-usage"], capture_output=True, text=True)
txt = out.stdout + out.stderr
pat = sys.argv[1] if len(sys.argv) > 1 else ""
cur = None
rows = {}
for line in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur).replace("void asurf::", "")
        rows[cur] = {}
        continue
    for key, rx in (("v", r" VGPRs: (\d+)"), ("s", r"ScratchSize \[bytes/lane\]: (\d+)"),
                    ("o", r"Occupancy \[waves/SIMD\]: (\d+)"), ("l", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(rx, line)
        if m and cur:
            rows[cur][key] = int(m.group(1))
for k, r in rows.items():
    if pat in k:
        print(f"{k:48s} vgpr {r.get('v'):4d} scratch {r.get('s'):4d} occ {r.get('o')} lds {r.get('l')}")
