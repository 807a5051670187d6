#!/bin/bash
# GPU box: what a PERFECT forward-only depthwise -> pointwise fusion could gain: the stride-1 depthwise forward launches are skipped (their
# output zero-filled -- the write a fused forward still owes the pointwise weight gradient; results wrong by design), the backward pass is untouched.  Hack on the box only.  A/B/A/B.
cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_dw_fwd_bound.txt; : > $out
cp npp_amd/_ops.py /tmp/_ops.py.orig
python3 - <<'PY'
p = "npp_amd/_ops.py"
s = open(p).read()
old = """        check(lib().npp_dwconv_fwd(_byref(x), wf.data_ptr(), _byref(y), C.byref(g), stream_ptr()), "npp_dwconv_fwd")"""
new = """        if os.environ.get("NPP_HACK_NO_DW_FWD") == "1" and stride == 1:
            y.zero_()      # (the write a fused kernel still owes the backward pass: the pointwise weight gradient reads this tensor)
        else:
            check(lib().npp_dwconv_fwd(_byref(x), wf.data_ptr(), _byref(y), C.byref(g), stream_ptr()), "npp_dwconv_fwd")"""
assert old in s
open(p, "w").write(s.replace(old, new))
PY
for h in 0 1 0 1; do
  NPP_HACK_NO_DW_FWD=$h timeout -k 10 280 python3 bench.py --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('NPP_HACK_NO_DW_FWD=$h', d['ms_per_step'], 'ms', d['value'], 'img/s')" >> $out || echo "NPP_HACK_NO_DW_FWD=$h failed" >> $out
done
cp /tmp/_ops.py.orig npp_amd/_ops.py
cat $out
