"""Where a one-stream scene pass leaves the GPU idle: gaps between consecutive kernels of a rocprofv3 --kernel-trace of bench.py (--quick), per scene pass, attributed to the
kernel that FOLLOWS the gap (= what the host was late launching).  usage: gpu_idle.py <dir with *_kernel_trace.csv>"""
import csv, glob, json, os, sys
from collections import defaultdict
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# scene passes start with the first FeatureNet convolution after a marching-cubes / colour kernel: use the first kernel of build_volume = k_conv2d_x3<..> preceded by a gap
starts = [i for i, r in enumerate(rows) if "k_conv2d" in r[2] and (i == 0 or "k_conv2d" not in rows[i - 1][2]) and (i == 0 or any(t in rows[i - 1][2] for t in ("k_color_pts", "k_mesh", "k_mc_", "copyBuffer", "elementwise", "fill")))]
short = lambda n: n.split("(")[0].replace("void ", "").replace("o2345::", "")[:48]
res = {"kernels": len(rows), "pass_starts": len(starts)}
passes = []
for a, b in zip(starts[:-1], starts[1:]):
    seg = rows[a:b]
    wall = (seg[-1][1] - seg[0][0]) / 1e6
    if not (40 < wall < 120):
        continue
    busy = sum(e - s for s, e, _ in seg) / 1e6
    gaps = defaultdict(lambda: [0, 0.0])
    tot_gap = 0.0
    for (s0, e0, n0), (s1, e1, n1) in zip(seg[:-1], seg[1:]):
        g = (s1 - e0) / 1e3
        if g > 0:
            tot_gap += g
            if g > 4:
                gaps[short(n1)][0] += 1
                gaps[short(n1)][1] += g
    top = sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]
    passes.append({"kernels": len(seg), "wall_ms": round(wall, 3), "kernel_ms_sum": round(busy, 3), "gap_ms_total": round(tot_gap / 1e3, 3),
                   "gaps_over_4us_by_following_kernel(count, us)": [(k, v[0], round(v[1], 1)) for k, v in top]})
res["passes"] = passes[-3:]
print(json.dumps(res, indent=1))
