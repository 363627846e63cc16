from .base import Manifold
from .hyperboloid import Hyperboloid
from .poincare import PoincareBall

__all__ = ["Manifold", "Hyperboloid", "PoincareBall"]


def __getattr__(name):
    # `getattr(manifolds, args.manifold)` (models/encoders.py:51,94): the Euclidean manifold is the
    # identity on every map, i.e. the reference's Euclidean baselines -- outside the accelerated path
    if name == "Euclidean":
        raise NotImplementedError("manifold 'Euclidean' (identity maps: the Euclidean baselines) is outside "
                                  "hypdiff_amd's hot path; use PoincareBall or Hyperboloid")
    raise AttributeError(name)
