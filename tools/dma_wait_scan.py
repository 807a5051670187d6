"""Scan the gfx950 assembly of the LDS-DMA kernels for s_waitcnt vmcnt instructions the COMPILER inserted (outside inline asm): hipcc treats
an LDS-DMA (buffer_load ... lds) as a pending LDS write and, unless it can prove no alias, waits for it before the next ds_read --
which drains a prefetch that was meant to stay in flight under the MFMAs.
    python3 tools/dma_wait_scan.py [file.hip ...]        (default: every csrc/*.hip with an LDS-DMA; compiles with hipcc -S)"""
import os
import subprocess
import sys
from collections import Counter

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "npp_amd", "csrc")


def scan(path):
    asm = f"/tmp/dma_wait_scan_{os.path.basename(path)}.s"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-munsafe-fp-atomics", "-std=c++17", "-Wno-unused-result",
                    '-DNPP_SRC_HASH="x"', "-S", "--cuda-device-only", path, "-o", asm], check=True, stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
    inasm, cur, stats = False, None, {}
    for i, l in enumerate(lines):
        if l.startswith("_ZN") and ":" in l and "@" in l:
            cur = l.split(":")[0]
            stats[cur] = [0, 0, 0, []]
        if "ASMSTART" in l:
            inasm = True
        if "ASMEND" in l:
            inasm = False
        if cur is None:
            continue
        if "s_waitcnt" in l and "vmcnt" in l:
            if inasm:
                stats[cur][0] += 1
            else:
                stats[cur][1] += 1
                nxt = next((x.strip() for x in lines[i + 1:i + 4] if x.strip() and not x.strip().startswith(";")), "")
                stats[cur][3].append(nxt.split()[0] if nxt else "")
        if "lds" in l and "buffer_load" in l:
            stats[cur][2] += 1
    for k, v in stats.items():
        if v[2] == 0:
            continue
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        name = name.replace("(anonymous namespace)::", "")[:100]
        print(f"{os.path.basename(path):22s} own waits {v[0]:3d}  compiler waits {v[1]:3d}  dma {v[2]:3d}  followed by {dict(Counter(v[3]))}  {name}")


if __name__ == "__main__":
    files = sys.argv[1:]
    if not files:
        files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip") and "buffer_load_lds" in open(os.path.join(CSRC, f)).read()]
    for f in files:
        scan(f)
