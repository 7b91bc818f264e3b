"""Numerical mode of the network kernels.

``PRECISION`` (environment ``O2345_PRECISION``) selects how the dense layers of the SDF and colour networks are evaluated:

* ``"f16x3"`` (default)  every fp32 operand is split into two f16 halves (hi + lo, 22 significant bits) and each product is
  accumulated in fp32 as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16.  fp32-class accuracy (measured deviation from the
  oracle within 1.5x of the exact fp32 MFMA chain's), 5.3x less matrix time than the fp32 MFMA.
* ``"fp32"``  v_mfma_f32_32x32x2_f32, the exact fp32 FMA chain (strict mode).

(Rounds 1-2 also had a ``"bf16"`` SDF mode for BASELINE config 2's "bf16 SDF MLP" wording.  It was 1000x less accurate than the default and, since the
register fix of the default gradient kernel in round 3, slower as well (10.9 vs 9.9 ms): removed.)

The convolutions (FeatureNet, compress layer, sparse cost-regularisation network) follow the COLOUR mode: split-f16 on the matrix cores by
default, fp32 VALU kernels in ``"fp32"`` mode -- per object when a mode is handed to ``pipeline.SceneWeights`` / ``featurenet.set_precision``,
else this global one.  Cost-volume gather, sampling, compositing and marching cubes are precision-independent (fp32 / fp64 / integer)."""
import os

PRECISIONS = ("f16x3", "fp32")
PRECISION = os.environ.get("O2345_PRECISION", "f16x3")
if PRECISION not in PRECISIONS:
    raise ValueError(f"O2345_PRECISION must be one of {PRECISIONS}, got {PRECISION!r}")


def sdf_precision(p=None):
    p = PRECISION if p is None else p
    if p not in PRECISIONS:
        raise ValueError(f"unknown precision {p!r}")
    return p


def color_precision(p=None):
    p = PRECISION if p is None else p
    if p not in PRECISIONS:
        raise ValueError(f"unknown precision {p!r}")
    return p


# ---- tolerance-bounded colour work removal (include/o2345.h, O2345RenderIO.weight_cull) ----------------------------------------
# render(): the colour network (59 % of a render call) is evaluated only on occupied samples whose compositing weight w = alpha * T -- known after the
# SDF + gradient pass, before any colour -- is >= WEIGHT_CULL.  Colours lie in [0, 1], so a ray's colour moves by at most 128 * 2^-24 = 7.6e-6, a quarter
# of the 3e-5 stage tolerance; depth, weights, gradients and the colour mask are untouched.  What qualifies are the samples BEHIND a surface (T ~ 0):
# 6 % of the occupied samples at the untrained variance 0.2 (inv_s 7.4), 27 % / 47 % at a trained model's 0.45 / 0.65 (bench.py, `trained_regime`);
# free-space samples keep the reference's +1e-5 (w ~ 1e-5) and are never dropped.  O2345_WEIGHT_CULL=0 selects the exhaustive path (every occupied
# sample, the reference's work); both are tested.
# The bound assumes blended colours in [0, 1] (the reference's colour maps are images in [0, 1], One2345_eval_new_data.py:199-201; the blend is a convex
# combination of them): for maps of another range the bound scales with that range -- set O2345_WEIGHT_CULL accordingly.  "" or "0" = off, like the other knobs.
WEIGHT_CULL = float(os.environ.get("O2345_WEIGHT_CULL", "").strip() or 0.0) if "O2345_WEIGHT_CULL" in os.environ else 2.0 ** -24
if not (0.0 <= WEIGHT_CULL < 1.0):
    raise ValueError(f"O2345_WEIGHT_CULL must be in [0, 1), got {WEIGHT_CULL}")


def weight_cull(w=None):
    w = WEIGHT_CULL if w is None else float(w)
    if not (0.0 <= w < 1.0):
        raise ValueError(f"weight_cull must be in [0, 1), got {w}")
    return w


def warm_aten():
    """O2345_WARM_ATEN (default on): ops.preload() also launches, once on tiny tensors, the few ATen kernels the unchanged trainer runs inside its own timing
    brackets, so that their first-launch cost in a fresh process (2 - 14 ms each) is paid when the weights are loaded."""
    return os.environ.get("O2345_WARM_ATEN", "1") not in ("", "0")


def prepack_resolutions():
    """O2345_PREPACK_RESOLUTIONS (default "256" = run.py's --resolution, run.py:61-67): extraction-lattice resolutions whose layer-0 tables are built when the
    weights are loaded (2 ms each) instead of inside the first extract_fields call; any other resolution is built on first use."""
    v = os.environ.get("O2345_PREPACK_RESOLUTIONS", "256")
    return tuple(int(x) for x in v.replace(",", " ").split())

