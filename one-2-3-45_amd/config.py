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
