#!/bin/bash
# GPU box: conv_g8 ablation table + LDS counters -> gpurun_out/g8_ablation.txt (copy to profiles/r04_g8_ablation.txt)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/g8_ablation.txt; raw=gpurun_out/g8_ablation_raw.txt; : > $raw
for d in 0 1 2 4 8 15; do
  NPP_G8_DBG=$d timeout -k 10 200 python3 tools/g8_ablation.py 2>&1 | grep "^dbg" >> $raw || exit 1
done
python3 - $raw > $out <<'PY'
import sys, collections
rows = collections.OrderedDict()
for ln in open(sys.argv[1]):
    f = ln.split()
    key = (f[2], f[1])
    rows.setdefault(key, {})[f[0].split("=")[1]] = float(f[3])
print("conv_g8 ablation, N=16, 96x96, bf16, graph-replayed, us per launch (NPP_G8_DBG bits: 1 no epilogue, 2 no MFMA, 4 no DMA, 8 no fragment reads; 15 = barriers + loop skeleton only)")
print("1024->384 runs the BN=128 tile (4 x 2 waves of 64 x 64), the others BN=256 (2 x 4 waves of 128 x 64).  'full' = forward with input ReLU + statistics epilogue (the in-model form)")
cols = ["0", "full", "1", "2", "4", "8", "15"]
print(f"{'shape':12s} {'dir':6s} " + " ".join(f"{('dbg ' + c) if c != 'full' else c:>9s}" for c in cols) + "   TF/s(dbg 0)")
for (shape, d), v in rows.items():
    cin, cout = (int(t) for t in shape.split("->"))
    gf = 2.0 * 16 * 96 * 96 * cin * cout / 1e9
    print(f"{shape:12s} {d:6s} " + " ".join(f"{v[c]:9.1f}" if c in v else f"{'-':>9s}" for c in cols) + f"   {gf / v['0'] * 1e3:7.0f}")
PY
cat $out
# LDS counters of the unablated kernels (one --pmc pass; counters are per dispatch, summed over the g8 dispatches of the run)
rm -rf /tmp/g8pmc
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d /tmp/g8pmc -- python3 tools/g8_ablation.py > /dev/null 2>&1
f=$(find /tmp/g8pmc -name "*counter_collection.csv" | head -1)
python3 - "$f" >> $out <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
try:
    for r in csv.DictReader(open(sys.argv[1])):
        k = r["Kernel_Name"]
        if "conv_g8_kernel" not in k:
            continue
        k = k.replace("void (anonymous namespace)::", "").split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
    print()
    print("LDS counters per conv_g8 dispatch (rocprofv3 --pmc, averaged over the dispatches of tools/g8_ablation.py, dbg 0):")
    for k, v in acc.items():
        d = {c: v[c] / max(n[(k, c)], 1) for c in v}
        conf, act = d.get("SQ_LDS_BANK_CONFLICT", 0.0), d.get("SQ_LDS_IDX_ACTIVE", 0.0)
        print(f"  {k}: " + ", ".join(f"{c} {d[c]:.4g}" for c in sorted(d)) + (f"  -> bank-conflict cycles / LDS-active cycles = {conf / act:.3f}" if act else ""))
except Exception as e:      # noqa: BLE001
    print("counter pass failed:", e)
PY
tail -8 $out
