"""Per-shape conv kernel time of one training step.

  stage 1 (under rocprofv3 --kernel-trace):  python3 tools/shape_prof.py run  <log.json>
  stage 2:                                    python3 tools/shape_prof.py join <log.json> <kernel_trace.csv>
Stage 1 runs eager steps with npp_amd._ops.SHAPE_LOG on (one record per dense-conv launch, in launch order); stage 2
pairs the records with the conv kernels of the trace (same order) and prints time / TFLOP/s per (kind, shape)."""
import csv, json, os, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

FWD_K = ("conv_s1_kernel", "conv_igemm_kernel", "conv_g8_kernel", "conv_g4_kernel", "conv_h3_kernel", "conv_c32_kernel", "conv_thin_out_kernel",
         "conv_thin_in_kernel")
FIN_K = "conv_s1_finish_kernel"
WG_K = ("wgrad_tap_kernel", "conv_wgrad_kernel", "conv_wgrad_g4_kernel", "conv_wgrad_h3_kernel", "conv_wgrad_narrow_kernel", "conv_wgrad_thin_kernel",
        "conv_wgrad_g3_kernel")


def run(path, steps=3):
    os.environ["NPP_STREAMS"] = "1"      # launch order == execution order: the join pairs records with trace rows
    import torch
    import bench
    from npp_amd import _ops as K
    from npp_amd.model_augment import Network, set_compute_dtype
    from npp_amd.criterion import Criterion_par, Criterion_pose
    from npp_amd.synth import synth_batch
    dev = torch.device("cuda:0")
    set_compute_dtype(torch.bfloat16)
    torch.manual_seed(0)
    net = Network(bench.cfg_ns()).to(dev).train()
    cp, cq = Criterion_pose(out_len=2).to(dev), Criterion_par(out_len=2).to(dev)
    images, lpar, lpose, _ = synth_batch(16, 384, seed=0)
    images = torch.from_numpy(images).to(dev)
    lpar = [torch.from_numpy(a).to(dev) for a in lpar]
    lpose = [torch.from_numpy(a[:, :-1].copy()).to(dev) for a in lpose]
    K.SHAPE_LOG = []
    marks = []
    for _ in range(steps):
        marks.append(len(K.SHAPE_LOG))
        pose_list, par_list = net(images)
        loss = (cq(par_list, lpar).unsqueeze(0) + cp(pose_list, lpose).unsqueeze(0)).mean()
        net.zero_grad(set_to_none=True)
        loss.backward()
    torch.cuda.synchronize()
    json.dump({"log": K.SHAPE_LOG, "marks": marks}, open(path, "w"))


def join(log_path, trace_path):
    d = json.load(open(log_path))
    log = d["log"]
    rows, fins = [], []
    with open(trace_path) as f:
        for r in csv.DictReader(f):
            nm = r["Kernel_Name"]
            kind = "f" if any(k in nm for k in FWD_K) else ("w" if any(k in nm for k in WG_K) else None)
            if FIN_K in nm:               # split-K second launch: its time belongs to the conv launched just before
                fins.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
                continue
            if kind:
                rows.append((int(r["Start_Timestamp"]), kind, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                             nm.split("<")[0].split("::")[-1] + "<" + nm.split("<")[1].split(">")[0].replace("unsigned short, ", "") + ">"))
    rows.sort()
    import bisect
    starts = [r[0] for r in rows]
    for ts, dur in fins:
        i = bisect.bisect_right(starts, ts) - 1
        while i >= 0 and rows[i][1] != "f":
            i -= 1
        if i >= 0:
            r0 = rows[i]
            rows[i] = (r0[0], r0[1], r0[2] + dur, r0[3] + "+finish")
    fq = [r for r in rows if r[1] == "f"]
    wq = [r for r in rows if r[1] == "w"]
    lf = [l for l in log if l[0] in ("fwd", "dgrad")]
    lw = [l for l in log if l[0] == "wgrad"]
    assert len(fq) == len(lf) and len(wq) == len(lw), (len(fq), len(lf), len(wq), len(lw))
    agg = defaultdict(lambda: [0, 0.0, set()])
    first = d["marks"][1] if len(d["marks"]) > 1 else 0          # skip step 0 (weight packing, cold caches)
    nsteps = max(len(d["marks"]) - 1, 1)
    for lst, q in ((lf, fq), (lw, wq)):
        skip = sum(1 for l in log[:first] if l in lst or True) if False else None
    # per-record pairing, dropping everything logged before the second step
    idx_f = idx_w = 0
    for i, l in enumerate(log):
        if l[0] == "wgrad":
            r = wq[idx_w]; idx_w += 1
        else:
            r = fq[idx_f]; idx_f += 1
        if i < first:
            continue
        a = agg[tuple(l)]
        a[0] += 1; a[1] += r[2]; a[2].add(r[3])
    tot = sum(a[1] for a in agg.values())
    print(f"{nsteps} steps, conv kernel time {tot / nsteps / 1e3:.2f} ms/step")
    print(f"{'kind':>6} {'shape':>34} {'n/step':>6} {'avg_us':>8} {'ms/step':>8} {'TF/s':>7}  kernel")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        kind, n, ci, h, w, co, kh, kw, st, dil = k
        oh, ow = (h + st - 1) // st, (w + st - 1) // st
        gf = 2.0 * n * oh * ow * co * ci * kh * kw / 1e9
        avg = a[1] / a[0]
        shape = f"{ci}->{co} k{kh}x{kw} {h}x{w} s{st} d{dil}"
        print(f"{kind:>6} {shape:>34} {a[0] // nsteps:6d} {avg:8.1f} {a[1] / nsteps / 1e3:8.3f} {gf / avg * 1e3:7.1f}  {','.join(sorted(a[2]))}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        join(sys.argv[2], sys.argv[3])
