"""hypdiff_amd -- MI355X-native hot path of HypDiff (hyperbolic encode + direction diffusion).

Module layout mirrors the reference so existing call sites keep working:
    hypdiff_amd.manifolds.{PoincareBall, Hyperboloid}
    hypdiff_amd.layers.hyp_layers.{HypLinear, HypAgg, HypAct, HyperbolicGraphConvolution, HNNLayer}
    hypdiff_amd.models.encoders.{HGCN, HNN}
    hypdiff_amd.models.direction_diffusion.{diffusion, mlp}
All arithmetic runs in hand-written HIP kernels (libhypdiff_hip.so, include/hypdiff.h).
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"
