import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NPP_BENCH_SUPERVISE"] = "0"
import bench as B
from npp_amd import _ops as K
sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--graph", "0", "--no-cpu-baseline", "--no-prof"]
try:
    B.main()
except SystemExit:
    pass
print("BN_SUMS_STATS (tickets, delivered, consumed) over the eager steps run:", K.BN_SUMS_STATS, file=sys.stderr)
