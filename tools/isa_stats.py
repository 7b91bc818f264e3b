"""Static instruction mix of one kernel of a csrc/*.hip file (no GPU needed): compiles with -save-temps and histograms the ISA.

    python tools/isa_stats.py color_mfma.hip 'k_color_mfmaILi8ELb1' [--top 30]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib
B = importlib.import_module("one-2-3-45_amd.build")


def main():
    src, pat = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 25
    d = tempfile.mkdtemp(prefix="isa_")
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + ["-save-temps", "-c", os.path.join(B.CSRC, src), "-o", os.path.join(d, "o.o")]
    subprocess.check_call(cmd, cwd=d, stderr=subprocess.DEVNULL)
    sfile = [f for f in os.listdir(d) if f.endswith("gfx950.s")][0]
    s = open(os.path.join(d, sfile)).read().split("\n")
    starts = [i for i, l in enumerate(s) if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", l)]
    for st in starts:
        name = s[st].split(":")[0]
        end = next(i for i in range(st, len(s)) if s[i].startswith(".Lfunc_end"))
        body = s[st:end]
        ins = [l.strip().split()[0] for l in body if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        c = collections.Counter(ins)
        valu = sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma"))
        meta = {}
        for l in s[end:end + 400]:
            m = re.match(r"\s*; (NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize|TotalNumVgprs|SGPRBlocks|NumSgprs): (\d+)", l)
            if m and m.group(1) not in meta:
                meta[m.group(1)] = int(m.group(2))
        print(f"{name}\n  total {sum(c.values())}  VALU {valu}  MFMA {sum(v for k, v in c.items() if k.startswith('v_mfma'))}  "
              f"DS {sum(v for k, v in c.items() if k.startswith('ds_'))}  VMEM {sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_', 'scratch_', 'flat_')))}  "
              f"SALU {sum(v for k, v in c.items() if k.startswith('s_'))}  scratch {sum(v for k, v in c.items() if k.startswith('scratch_'))}  {meta}")
        print("  " + "  ".join(f"{k}:{v}" for k, v in c.most_common(top)))


if __name__ == "__main__":
    main()
