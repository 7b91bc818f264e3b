"""Turn rocprofv3 output into the small summaries kept under profiles/.

  python tools/summarize_rocprof.py stats <dir with *_kernel_stats.csv> <out.md> "<title line>"
  python tools/summarize_rocprof.py trace <dir with *_kernel_trace.csv> <out.md> "<title line>"      (per launch-size class)
  python tools/summarize_rocprof.py pmc <dir with one sub-directory per --pmc pass> <out.json>
       -> {kernel: {counter: mean value per launch}} for the o2345 kernels, averaged over the launches within 20 % of the largest value
          (the timed full-size launches, not the tiny ones of the setup render)
"""
import collections, csv, glob, json, os, re, sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("o2345::", "")


def stats(d, out, title):
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True))[0]
    rows = list(csv.DictReader(open(f)))
    with open(out, "w") as o:
        o.write(f"# {title}\n\n| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
        for r in rows[:40]:
            n = r["Name"] if len(r["Name"]) <= 100 else r["Name"][:97] + "..."
            o.write(f"| `{n}` | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.1f} | {r['Percentage']} |\n")
    csv_out = out[:-3] + ".csv"
    with open(csv_out, "w") as o:
        o.write(open(f).read())


def trace(d, out, title):
    """Per-kernel rows from the kernel TRACE (not the stats file), one row per launch-size class: persistent kernels launch the same grid for
    every problem size, so the launches of a kernel are clustered by duration (a new class wherever consecutive sorted durations differ by
    more than 2.5x) -- a render-size launch and a 0.4 ms vertex-colour launch of the same kernel never share an average."""
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        per[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    rows = []
    for k, ds in per.items():
        ds.sort()
        grp = [ds[0]]
        for x in ds[1:]:
            if x > 2.5 * grp[-1] and x > grp[-1] + 5.0:
                rows.append((k, grp)); grp = []
            grp.append(x)
        rows.append((k, grp))
    total = sum(sum(g) for _, g in rows)
    rows.sort(key=lambda r: -sum(r[1]))
    with open(out, "w") as o:
        o.write(f"# {title}\n\n(one row per kernel and launch-size class; times in microseconds)\n\n| kernel | launches | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
        for k, g in rows[:60]:
            n = k if len(k) <= 90 else k[:87] + "..."
            o.write(f"| `{n}` | {len(g)} | {sum(g) / 1e3:.3f} | {sum(g) / len(g):.1f} | {g[0]:.1f} | {g[-1]:.1f} | {100 * sum(g) / total:.2f} |\n")


def provenance():
    """What the numbers were measured on: the kernel sources' hash (one-2-3-45_amd/build.py:sources_sha -- bench.py refuses to print counter
    numbers whose hash differs from the tree it runs in), bench.py's hash (what the driver records as bench_py_sha16) and, when .git travels, HEAD."""
    import hashlib
    import importlib
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    meta = {"kernel_sources_sha": importlib.import_module("one-2-3-45_amd.build").sources_sha(),
            "bench_py_sha16": hashlib.sha256(open(os.path.join(root, "bench.py"), "rb").read()).hexdigest()[:16]}
    try:
        meta["git_head"] = subprocess.check_output(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        meta["git_head"] = None                   # the gpurun snapshot carries no .git
    return meta


def pmc(d, out):
    res = collections.defaultdict(dict)
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        per = collections.defaultdict(lambda: collections.defaultdict(list))      # kernel -> counter -> [(grid, value)]
        for r in csv.DictReader(open(f)):
            if "o2345::" not in r["Kernel_Name"]:
                continue
            per[short(r["Kernel_Name"])][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
        for k, cs in per.items():
            for c, vals in cs.items():
                # persistent kernels launch the same grid for every problem size: keep the full-size launches by VALUE
                top = max(v[1] for v in vals)
                big = [v[1] for v in vals if v[1] >= 0.8 * top]
                res[k][c] = sum(big) / len(big)
                res[k]["launches_averaged"] = len(big)
    res["_meta"] = provenance()
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3], sys.argv[4])
    elif sys.argv[1] == "trace":
        trace(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        pmc(sys.argv[2], sys.argv[3])
