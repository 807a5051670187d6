#!/bin/bash
# GPU box: UPPER BOUND of a fused depthwise -> pointwise DilConvS.  A hack that exists only on the box: every stride-1
# DilConvS skips its depthwise conv altogether (forward, data gradient and weight gradient gone; the 1x1 reads x through its ReLU) -- results
# wrong by design.  A real fusion keeps the depthwise arithmetic and its backward, so it can only gain LESS than this.  A/B/A/B on one box.
cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_dw_upper_bound.txt; : > $out
cp npp_amd/operations.py /tmp/operations.py.orig
python3 - <<'PY'
p = "npp_amd/operations.py"
s = open(p).read()
old = """        y = K.dwconv2d(x, dw.weight, dw.stride[0], dw.padding[0], dw.dilation[0], relu_in=True)
        y, st = K.conv2d(y, pw.weight, None, 1, 0, 1, relu_in=False, want_stats=_use_batch_stats(bn), private_in=True)"""
new = """        import os
        if os.environ.get("NPP_HACK_NO_DW") == "1" and dw.stride[0] == 1:
            y, st = K.conv2d(x, pw.weight, None, 1, 0, 1, relu_in=True, want_stats=_use_batch_stats(bn))
        else:
            y = K.dwconv2d(x, dw.weight, dw.stride[0], dw.padding[0], dw.dilation[0], relu_in=True)
            y, st = K.conv2d(y, pw.weight, None, 1, 0, 1, relu_in=False, want_stats=_use_batch_stats(bn), private_in=True)"""
assert old in s
open(p, "w").write(s.replace(old, new))
PY
for h in 0 1 0 1; do
  NPP_HACK_NO_DW=$h timeout -k 10 280 python3 bench.py --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('NPP_HACK_NO_DW=$h', d['ms_per_step'], 'ms', d['value'], 'img/s')" >> $out || echo "NPP_HACK_NO_DW=$h failed" >> $out
done
cp /tmp/operations.py.orig npp_amd/operations.py
cat $out
