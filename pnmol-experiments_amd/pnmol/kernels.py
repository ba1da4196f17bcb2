"""Covariance kernels (reference: src/pnmol/kernels.py).

Same classes and call conventions; derivatives of kernels (the reference obtains them with
jax autodiff through `pnmol.diffops`) are closed-form here -- see `Kernel.derivative`.
"""

import concurrent.futures
import math
import os

import numpy as np
import scipy.linalg


_GRAM_PARALLEL_FROM = 1 << 20     # entries
_GRAM_CHUNK_ENTRIES = 1 << 18     # per task (2 MB of fp64 per temporary)


def _gram_threads():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(8, n))


class Kernel:
    """k(x, y) scalar for 1-d inputs; k(X, Y) diagonal for equal shapes (N, dim);
    k(X, Y.T-like (dim, K)) full Gram (N, K)  (kernels.py:16-47)."""

    def pairwise(self, x, y):
        return self._eval(np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64))

    def _eval(self, X, Y):  # broadcasting over leading axes, last axis = spatial dimension
        raise NotImplementedError

    def __call__(self, X, Y):
        X, Y = np.asarray(X, dtype=np.float64), np.asarray(Y, dtype=np.float64)
        if X.ndim == Y.ndim <= 1 or X.shape == Y.shape:
            return self._eval(X, Y)
        Yt = Y.T[None, :, :]
        if X.shape[0] * Yt.shape[1] < _GRAM_PARALLEL_FROM:
            return self._eval(X[:, None, :], Yt)
        # large Gram matrices (64x64 mesh: 4096^2 entries, 0.6 s of the set-up as one expression): row chunks on a few
        # threads -- the ufuncs release the GIL, the chunks' temporaries stay in cache, every entry is computed by the same
        # elementwise operations as above (bit-identical)
        out = np.empty((X.shape[0], Yt.shape[1]))
        step = max(64, _GRAM_CHUNK_ENTRIES // Yt.shape[1])

        def rows(i):
            out[i:i + step] = self._eval(X[i:i + step, None, :], Yt)

        with concurrent.futures.ThreadPoolExecutor(max_workers=_gram_threads()) as pool:
            list(pool.map(rows, range(0, X.shape[0], step)))
        return out

    def __add__(self, other):  # kernels.py:50-55
        return _Sum(self, other)

    def derivative(self, spec):
        """Closed-form derivative kernel; `spec` is a tuple of (operator, argnum)."""
        raise NotImplementedError(f"{type(self).__name__} has no closed-form derivative {spec}")

    def __str__(self):
        return f"{self.__class__.__name__}()"


class _Sum(Kernel):
    def __init__(self, a, b):
        self.a, self.b = a, b

    def _eval(self, X, Y):
        return self.a._eval(X, Y) + self.b._eval(X, Y)

    def derivative(self, spec):
        return _Sum(self.a.derivative(spec), self.b.derivative(spec))


class Lambda(Kernel):
    """Kernel from a broadcasting pairwise function (kernels.py:58-65)."""

    def __init__(self, fun, /, parent=None, spec=()):
        self._fun, self.parent, self.spec = fun, parent, tuple(spec)

    def _eval(self, X, Y):
        return self._fun(X, Y)

    def derivative(self, spec):
        if self.parent is None:
            raise NotImplementedError("derivatives of an arbitrary Lambda kernel need autodiff (out of scope)")
        return self.parent.derivative(self.spec + tuple(spec))


_SPECS = {
    (("laplace", 0),): "laplace_x",
    (("laplace", 0), ("laplace", 1)): "laplace_xy",
    (("gradient", 0),): "grad_x_1d",
    (("gradient", 0), ("gradient", 1)): "grad_xy_1d",
}


class _ClosedForm(Kernel):
    def derivative(self, spec):
        name = _SPECS.get(tuple(spec))
        if name is None or not hasattr(self, name):
            raise NotImplementedError(f"{type(self).__name__}: no closed form for {spec}")
        return Lambda(getattr(self, name), parent=self, spec=spec)


class _RadialKernel(_ClosedForm):
    def __init__(self, *, output_scale=1.0, input_scale=1.0):
        self._output_scale, self._input_scale = output_scale, input_scale

    output_scale = property(lambda self: self._output_scale)
    input_scale = property(lambda self: self._input_scale)
    output_scale_squared = property(lambda self: self._output_scale ** 2)
    input_scale_squared = property(lambda self: self._input_scale ** 2)


class SquareExponential(_RadialKernel):
    """s^2 exp(-l^2 |x-y|^2 / 2)  (kernels.py:107-111; the input scale multiplies)."""

    def _eval(self, X, Y):
        r2 = np.sum((X - Y) ** 2, axis=-1) * self.input_scale ** 2
        return self.output_scale ** 2 * np.exp(-r2 / 2.0)

    def laplace_x(self, X, Y):
        c, n = self.input_scale ** 2, X.shape[-1]
        r2 = np.sum((X - Y) ** 2, axis=-1)
        return (c * c * r2 - c * n) * self._eval(X, Y)

    def laplace_xy(self, X, Y):
        c, n = self.input_scale ** 2, X.shape[-1]
        r2 = np.sum((X - Y) ** 2, axis=-1)
        phi = c * c * r2 - c * n
        return (phi * phi - 4.0 * c ** 3 * r2 + 2.0 * c * c * n) * self._eval(X, Y)

    def grad_x_1d(self, X, Y):
        c, u = self.input_scale ** 2, (X - Y)[..., 0]
        return -c * u * self._eval(X, Y)

    def grad_xy_1d(self, X, Y):
        c, u = self.input_scale ** 2, (X - Y)[..., 0]
        return (c - c * c * u * u) * self._eval(X, Y)


class Matern52(_RadialKernel):
    """kernels.py:114-124.  1-d derivatives only; at x == y they equal the Maclaurin values the
    reference hard-codes over the autodiff NaN (discretize.py:184-197)."""

    def _eval(self, X, Y):
        r = np.sqrt(5.0 * np.sum((X - Y) ** 2, axis=-1) * self.input_scale ** 2)
        return self.output_scale ** 2 * (1.0 + r + r * r / 3.0) * np.exp(-r)

    def _ar(self, X, Y):
        if X.shape[-1] != 1:
            raise NotImplementedError("Matern52 derivatives are implemented in 1-d only")
        return math.sqrt(5.0) * self.input_scale, np.abs((X - Y)[..., 0])

    def laplace_x(self, X, Y):
        a, r = self._ar(X, Y)
        return self.output_scale ** 2 * (a * a / 3.0) * (a * a * r * r - a * r - 1.0) * np.exp(-a * r)

    def laplace_xy(self, X, Y):
        a, r = self._ar(X, Y)
        return self.output_scale ** 2 * (a ** 4 / 3.0) * (3.0 - 5.0 * a * r + a * a * r * r) * np.exp(-a * r)

    def grad_x_1d(self, X, Y):
        a, r = self._ar(X, Y)
        return -self.output_scale ** 2 * (a * a / 3.0) * (X - Y)[..., 0] * (1.0 + a * r) * np.exp(-a * r)

    def grad_xy_1d(self, X, Y):
        a, r = self._ar(X, Y)
        return self.output_scale ** 2 * (a * a / 3.0) * (1.0 + a * r - a * a * r * r) * np.exp(-a * r)


class Polynomial(_ClosedForm):
    """(x.y + c)^p  (kernels.py:127-144)."""

    def __init__(self, *, order=2, const=1.0):
        self._order, self._const = order, const

    order = property(lambda self: self._order)
    const = property(lambda self: self._const)

    def _eval(self, X, Y):
        return (np.sum(X * Y, axis=-1) + self.const) ** self.order

    @staticmethod
    def _pw(base, e):
        return np.ones_like(base) if e <= 0 else base ** e

    def laplace_x(self, X, Y):
        p, s = self.order, np.sum(X * Y, axis=-1) + self.const
        return p * (p - 1) * self._pw(s, p - 2) * np.sum(Y * Y, axis=-1)

    def laplace_xy(self, X, Y):
        p, n = self.order, X.shape[-1]
        xy = np.sum(X * Y, axis=-1)
        s = xy + self.const
        x2, y2 = np.sum(X * X, axis=-1), np.sum(Y * Y, axis=-1)
        t = (p - 2) * (p - 3) * self._pw(s, p - 4) * x2 * y2
        t = t + 4.0 * (p - 2) * self._pw(s, p - 3) * xy + 2.0 * n * self._pw(s, p - 2)
        return p * (p - 1) * t

    def grad_x_1d(self, X, Y):
        p, s = self.order, np.sum(X * Y, axis=-1) + self.const
        return p * self._pw(s, p - 1) * Y[..., 0]

    def grad_xy_1d(self, X, Y):
        p, s = self.order, np.sum(X * Y, axis=-1) + self.const
        return p * self._pw(s, p - 1) + p * (p - 1) * self._pw(s, p - 2) * X[..., 0] * Y[..., 0]


class WhiteNoise(Kernel):
    """s^2 [x == y]  (kernels.py:147-157)."""

    def __init__(self, *, output_scale=1.0):
        self._output_scale = output_scale

    output_scale = property(lambda self: self._output_scale)

    def _eval(self, X, Y):
        return self.output_scale ** 2 * np.all(X == Y, axis=-1)


class _StackedKernel(Kernel):  # kernels.py:160-176
    def __init__(self, *, kernel_list):
        self.kernel_list = kernel_list

    def __call__(self, X, Y):
        grams = [k(X, Y) for k in self.kernel_list]
        if np.shape(X) == np.shape(Y):
            return np.concatenate(grams)
        return scipy.linalg.block_diag(*grams)


def duplicate(kernel, num):
    """Block-diagonal stack of `num` copies of a kernel (kernels.py:179-184)."""
    return _StackedKernel(kernel_list=[kernel] * num)
