"""ISA-level bisect of the co-resident-MFMA corruption (profiles/NOTES.md): takes the device assembly of csrc/costvol.hip compiled WITH packed-FP32 instructions,
rewrites chosen groups of v_pk_*_f32 instructions of k_costvol_gather<16> into pairs of plain VALU instructions (everything else byte-identical), and assembles one
code object per variant.  usage: make_variants.py <outdir>"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = sys.argv[1]
os.makedirs(OUT, exist_ok=True)
LLVM = "/opt/rocm/lib/llvm/bin"
KERNEL = "_ZN5o234516k_costvol_gatherILi16EEEvPKfS2_iiiNS_7VolGeomEPKhPKiiPf"
src_s = os.path.join(OUT, "costvol_pk.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "--cuda-device-only", "-S",
                       os.path.join(ROOT, "one-2-3-45_amd", "csrc", "costvol.hip"), "-o", src_s], stderr=subprocess.DEVNULL)
text = open(src_s).read()
m = re.search(r"^%s:.*?\n(.*?)\.Lfunc_end\d+:" % KERNEL, text, re.S | re.M)
body = m.group(1)
lines = body.split("\n")


def parse_mods(rest):
    mods = {"op_sel": None, "op_sel_hi": None, "neg_lo": None, "neg_hi": None}
    for k in list(mods):
        mm = re.search(r"\b%s:\[([01,]+)\]" % k, rest)
        if mm:
            mods[k] = [int(x) for x in mm.group(1).split(",")]
            rest = rest.replace(mm.group(0), "")
    return rest.strip(), mods


def half(op, sel):
    """op = 'v[a:b]' | 's[a:b]' | literal; sel 0 = low half, 1 = high half."""
    mm = re.match(r"([vs])\[(\d+):(\d+)\]$", op)
    if mm:
        return "%s%d" % (mm.group(1), int(mm.group(2)) + sel)
    return op                                                     # inline constant: the same value for both halves


def regs_of(opnd):
    mm = re.match(r"v(\d+)$", opnd.lstrip("-"))
    return {int(mm.group(1))} if mm else set()


def split_pk(line):
    ins = line.strip()
    mm = re.match(r"(v_pk_(mul|add|fma)_f32)\s+(.*)$", ins)
    op = mm.group(2)
    rest, mods = parse_mods(mm.group(3))
    ops = [o.strip() for o in rest.split(",")]
    dst, srcs = ops[0], ops[1:]
    n = len(srcs)
    op_sel = mods["op_sel"] or [0] * n
    op_sel_hi = mods["op_sel_hi"] or [1] * n
    neg_lo = mods["neg_lo"] or [0] * n
    neg_hi = mods["neg_hi"] or [0] * n
    lo = [("-" if neg_lo[i] else "") + half(srcs[i], op_sel[i]) for i in range(n)]
    hi = [("-" if neg_hi[i] else "") + half(srcs[i], op_sel_hi[i]) for i in range(n)]
    d = int(re.match(r"v\[(\d+):", dst).group(1))
    mn = {"mul": "v_mul_f32_e64", "add": "v_add_f32_e64", "fma": "v_fma_f32"}[op]
    i_lo = "\t%s v%d, %s" % (mn, d, ", ".join(lo))
    i_hi = "\t%s v%d, %s" % (mn, d + 1, ", ".join(hi))
    reads_lo = set().union(*[regs_of(x) for x in lo])
    reads_hi = set().union(*[regs_of(x) for x in hi])
    if d not in reads_hi:
        return [i_lo, i_hi]
    if (d + 1) not in reads_lo:
        return [i_hi, i_lo]
    # both orders conflict: the kernel's allocation is rounded up to 64 registers and it uses v0..v59, so v60 is a free temporary
    hi_t = [re.sub(r"\bv%d\b" % d, "v60", x) for x in hi]
    return ["\tv_mov_b32_e32 v60, v%d" % d, i_lo, "\t%s v%d, %s" % (mn, d + 1, ", ".join(hi_t))]


# ---- regions of the kernel ------------------------------------------------------------------------------------------
def region_of(i):
    """'epi' (mean / var), 'proj' (projection + tap set-up of the lane's own view), 'acc' (running sums s1 / s2 += f, f*f), 'tap' (a*w, f += a*w)."""
    ins = lines[i].strip()
    if i < first_loop:
        return "epi"
    if proj_start <= i < proj_end:
        return "proj"
    d = int(re.search(r"v\[(\d+):", ins).group(1))
    if 8 <= d <= 15:
        return "acc"
    mm = re.match(r"v_pk_mul_f32 v\[\d+:\d+\], (v\[\d+:\d+\]), (v\[\d+:\d+\])$", ins)
    if mm and mm.group(1) == mm.group(2):
        return "acc"                                              # f * f
    return "tap"


end_first = next(i for i, l in enumerate(lines) if "s_endpgm" in l)
first_loop = end_first
proj_start = next(i for i, l in enumerate(lines) if i > end_first and "s_and_saveexec_b64" in l)
proj_end = next(i for i, l in enumerate(lines) if i > proj_start and "s_or_b64 exec, exec" in l)
pk = [i for i, l in enumerate(lines) if l.strip().startswith("v_pk_")]
regions = {i: region_of(i) for i in pk}
count = {}
for r in regions.values():
    count[r] = count.get(r, 0) + 1
print("packed instructions per region:", count, file=sys.stderr)

VARIANTS = {"orig": set(), "all": {"epi", "proj", "acc", "tap"}, "only_proj": {"proj"}, "only_tap": {"tap"}, "only_acc": {"acc"}, "only_epi": {"epi"},
            "all_but_tap": {"epi", "proj", "acc"}, "all_but_acc": {"epi", "proj", "tap"}, "all_but_proj": {"epi", "tap", "acc"}}
# second level: inside the projection block, every packed instruction on its own ("proj3" = the fourth one)
proj_ids = [i for i in pk if regions[i] == "proj"]
for k, i in enumerate(proj_ids):
    regions[i] = "proj%d" % k
PROJ = {"proj%d" % k for k in range(len(proj_ids))}
if "--level2" in sys.argv:
    VARIANTS = {"orig": set(), "proj_all_split": set(PROJ)}
    for k in range(len(proj_ids)):
        VARIANTS["proj_keep_only_%02d" % k] = PROJ - {"proj%d" % k}          # exactly ONE packed instruction left in the projection block
else:
    VARIANTS = {k: (v - {"proj"}) | (PROJ if "proj" in v else set()) for k, v in VARIANTS.items()}
for name, regs in VARIANTS.items():
    out = []
    for i, l in enumerate(lines):
        if i in regions and regions[i] in regs:
            out.extend(split_pk(l))
        else:
            out.append(l)
    new_text = text.replace(body, "\n".join(out))
    s_path = os.path.join(OUT, name + ".s")
    open(s_path, "w").write(new_text)
    o_path = os.path.join(OUT, name + ".o")
    subprocess.check_call([os.path.join(LLVM, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s_path, "-o", o_path])
    subprocess.check_call([os.path.join(LLVM, "ld.lld"), "-shared", o_path, "-o", os.path.join(OUT, name + ".hsaco")])
    os.remove(o_path)
print("built", sorted(VARIANTS), file=sys.stderr)
