"""Last HIP API calls before the N-th hipStreamEndCapture of an AMD_LOG_LEVEL=3 log: names + a few arguments, for diffing two runs."""
import re, sys
path, nth, count = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
calls = []
seen = 0
pat = re.compile(r"\x1b\[32m (hip\w+) \((.*)\) \x1b\[0m")
with open(path, errors="replace") as f:
    for ln in f:
        m = pat.search(ln)
        if not m:
            continue
        name = m.group(1)
        if name in ("hipGetDevice", "hipSetDevice", "hipGetLastError", "hipPeekAtLastError", "hipDeviceGetAttribute"):
            continue
        calls.append(name + " " + m.group(2)[:80])
        if name == "hipStreamEndCapture":
            seen += 1
            if seen == nth:
                break
print("\n".join(calls[-count:]))
from collections import Counter
c = Counter(x.split()[0] for x in calls)
print("TOTALS", dict(c.most_common(12)))
