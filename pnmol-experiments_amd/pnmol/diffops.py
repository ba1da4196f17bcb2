"""Differential operators (reference: src/pnmol/diffops.py).

The reference composes operators over `jax.grad`/`jax.jacrev`.  The filter path only ever applies
`laplace()` (pde/examples.py:53, discretize.py:51-52) and `gradient()` (Neumann rows,
discretize.py:128-131) to covariance kernels, so here an operator applied to `kernel.pairwise`
resolves to the kernel's closed-form derivative.  Other operators of the reference's algebra
are outside the hot-path scope.
"""

from . import kernels as _kernels


class DifferentialOperator:
    def __init__(self, name):
        self.name = name

    def __call__(self, fun, argnums=0):
        owner = getattr(fun, "__self__", None)
        if not isinstance(owner, _kernels.Kernel):
            raise NotImplementedError(
                f"diffops.{self.name}() acts on `kernel.pairwise` only (closed-form derivatives, no autodiff)")
        return owner.derivative(((self.name, argnums),)).pairwise

    def __repr__(self):
        return f"diffops.{self.name}()"


def laplace():
    """Laplace operator (diffops.py:189-202)."""
    return DifferentialOperator("laplace")


def gradient():
    """Gradient; scalar derivative in 1-d (diffops.py:167-174)."""
    return DifferentialOperator("gradient")
