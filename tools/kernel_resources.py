"""VGPR / LDS / occupancy of every kernel of a .hip unit (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
-> profiles/rNN_kernel_resources.txt"""
import re, subprocess, sys, os
src = sys.argv[1]
r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-c", src, "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
rows, cur = [], None
for line in r.stderr.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = m.group(2)
for r_ in rows:
    print(f"{r_['name'][:110]:110s} VGPR {r_.get('VGPRs','?'):>4} AGPR {r_.get('AGPRs','?'):>3} LDS {r_.get('LDS Size','?'):>6} waves/SIMD {r_.get('Occupancy','?')} spill {r_.get('VGPRs Spill','?')}")
