from .base import Manifold, ManifoldParameter
from .euclidean import Euclidean
from .hyperboloid import Hyperboloid
from .poincare import PoincareBall

__all__ = ["Manifold", "ManifoldParameter", "Euclidean", "Hyperboloid", "PoincareBall"]
