#!/bin/bash
# GPU box: rocprofv3 kernel stats of the 1-rank rehearsal of the N > 1 path (bench.py --force-dist, NPP_P2P_ALONE=1) -> top table
# (NPP_P2P_SELFTEST=0: the 2000 acceptance exchanges of comm.enable_p2p are set-up, not step work)
export NPP_BENCH_SUPERVISE=0 GPU_MAX_HW_QUEUES=2 NPP_P2P_ALONE=1 NPP_STREAMS=1 NPP_P2P_SELFTEST=0
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/qsd
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/qsd -- python3 bench.py --force-dist --steps 5 --warmup 2 --graph 0 --no-cpu-baseline --no-prof > gpurun_out/quick_stats_dist.log 2>&1
tail -1 gpurun_out/quick_stats_dist.log | cut -c1-200
s=$(find /tmp/qsd -name "*kernel_stats.csv" | head -1)
cp "$s" gpurun_out/quick_stats_dist.csv
python3 - <<'PY'
import csv
def load(p):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(p))}
d = load("gpurun_out/quick_stats_dist.csv")
try:
    l = load("profiles/r05_kernel_stats_bs16_bf16_eager_1stream.csv")
except Exception:
    l = {}
# steps covered by each trace: every step launches pack_weights_batched_kernel exactly once
def nsteps(t, default):
    return next((v[0] for k, v in t.items() if "pack_weights_batched_kernel" in k), default)
steps, lsteps = nsteps(d, 7), nsteps(l, 15)
d = {k: (v[0] / steps, v[1] / steps) for k, v in d.items()}
l = {k: (v[0] / lsteps, v[1] / lsteps) for k, v in l.items()}
steps = 1
names = sorted(set(d) | set(l), key=lambda n: -abs(d.get(n, (0, 0))[1] - l.get(n, (0, 0))[1]))
print(f"dist: {sum(v[0] for v in d.values()) / steps:.0f} launches/step, {sum(v[1] for v in d.values()) / steps / 1e6:.2f} ms kernel time;  local: {sum(v[0] for v in l.values()) / steps:.0f}, {sum(v[1] for v in l.values()) / steps / 1e6:.2f}")
for n in names[:45]:
    a, b = d.get(n, (0, 0)), l.get(n, (0, 0))
    print(f"{n.replace('void (anonymous namespace)::', '')[:70]:70s} dist {a[0] / steps:6.1f} {a[1] / steps / 1e6:6.3f} ms | local {b[0] / steps:6.1f} {b[1] / steps / 1e6:6.3f} ms")
PY
