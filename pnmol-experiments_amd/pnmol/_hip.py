"""ctypes binding of `libpnmol_hip.so` (C ABI declared in include/pnmol_hip.h).

There is no CPU fallback: if the library or a GPU is missing, the solver classes raise.
"""

import ctypes
import os
import pathlib

import numpy as np


_c_double_p = ctypes.POINTER(ctypes.c_double)
_HERE = pathlib.Path(__file__).resolve().parent
LIB_PATH = pathlib.Path(os.environ.get("PNMOL_HIP_LIB", _HERE.parent / "lib" / "libpnmol_hip.so"))


class PnmolHipError(RuntimeError):
    """A call into libpnmol_hip.so failed."""


class StepOut(ctypes.Structure):
    _fields_ = [
        ("t_new", ctypes.c_double),
        ("diffusion_squared_local", ctypes.c_double),
        ("sigma2_whitened", ctypes.c_double),
        ("error_sigma2", ctypes.c_double),
        ("info", ctypes.c_int),
    ]


class FilterDesc(ctypes.Structure):
    _fields_ = [
        ("d", ctypes.c_int),
        ("num_derivatives", ctypes.c_int),
        ("nB", ctypes.c_int),
        ("L", _c_double_p),
        ("B", _c_double_p),
        ("E_sqrtm", _c_double_p),
        ("R_sqrtm", _c_double_p),
        ("Gamma", _c_double_p),
        ("d_state", ctypes.c_int),
        ("dtype", ctypes.c_int),      # 0 fp64, 1 fp32 covariance (include/pnmol_hip.h)
        ("K", _c_double_p),           # optional Gram matrix Gamma Gamma^T (saves the library an O(d^3) host loop)
    ]


# name -> (restype, argtypes); every symbol include/pnmol_hip.h declares
_vp = ctypes.c_void_p
SYMBOLS = {
    "pnmol_abi_version": (ctypes.c_int, []),
    "pnmol_device_count": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    "pnmol_ctx_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_vp)]),
    "pnmol_ctx_destroy": (ctypes.c_int, [_vp]),
    "pnmol_ctx_synchronize": (ctypes.c_int, [_vp]),
    "pnmol_fd_solve_batched": (ctypes.c_int, [_vp, _c_double_p, _c_double_p, _c_double_p, ctypes.c_int, ctypes.c_int,
                                              _c_double_p, _c_double_p]),
    "pnmol_cholesky_lower": (ctypes.c_int, [_vp, _c_double_p, ctypes.c_int, _c_double_p]),
    "pnmol_last_error": (ctypes.c_char_p, [_vp]),
    "pnmol_filter_create": (ctypes.c_int, [_vp, ctypes.POINTER(FilterDesc), ctypes.POINTER(_vp)]),
    "pnmol_filter_destroy": (ctypes.c_int, [_vp]),
    "pnmol_filter_set_error_model": (ctypes.c_int, [_vp, ctypes.c_double, _c_double_p, _c_double_p]),
    "pnmol_filter_predict_mean": (ctypes.c_int, [_vp, _vp, ctypes.c_double, _c_double_p]),
    "pnmol_filter_set_operator": (ctypes.c_int, [_vp, _c_double_p, _c_double_p]),
    "pnmol_filter_set_operator_diagonal": (ctypes.c_int, [_vp, _c_double_p, _c_double_p]),
    "pnmol_state_create": (ctypes.c_int, [_vp, ctypes.POINTER(_vp)]),
    "pnmol_state_destroy": (ctypes.c_int, [_vp]),
    "pnmol_state_clone": (ctypes.c_int, [_vp, ctypes.POINTER(_vp)]),
    "pnmol_state_set": (ctypes.c_int, [_vp, ctypes.c_double, _c_double_p, _c_double_p]),
    "pnmol_state_set_sqrtm": (ctypes.c_int, [_vp, ctypes.c_double, _c_double_p, _c_double_p]),
    "pnmol_state_get_time": (ctypes.c_int, [_vp, _c_double_p]),
    "pnmol_state_get_mean": (ctypes.c_int, [_vp, _c_double_p]),
    "pnmol_state_get_cov_sqrtm": (ctypes.c_int, [_vp, _c_double_p]),
    "pnmol_state_get_cov": (ctypes.c_int, [_vp, _c_double_p]),
    "pnmol_state_get_marginal_var": (ctypes.c_int, [_vp, _c_double_p]),
    "pnmol_filter_step": (ctypes.c_int, [_vp, _vp, ctypes.c_double, _vp, ctypes.POINTER(StepOut), _c_double_p]),
    "pnmol_filter_steps": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_double, _c_double_p, _c_double_p,
                                          ctypes.POINTER(StepOut)]),
    "pnmol_filter_steps_begin": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_double]),
    "pnmol_filter_steps_end": (ctypes.c_int, [_vp, _vp, _c_double_p, _c_double_p, ctypes.POINTER(StepOut)]),
    "pnmol_filter_prepare_steps": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_double]),
    "pnmol_filter_last_steps_ms": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_float)]),
    "pnmol_filter_prepare_error_model": (ctypes.c_int, [_vp, ctypes.c_double]),
    "pnmol_filter_debug_read": (ctypes.c_int, [_vp, ctypes.c_int, _c_double_p, ctypes.c_long]),
    "pnmol_filter_debug_poison": (ctypes.c_int, [_vp]),
    "pnmol_filter_dims": (ctypes.c_int, [_vp] + [ctypes.POINTER(ctypes.c_int)] * 5),
    "pnmol_filter_sweep_layout": (ctypes.c_int, [_vp] + [ctypes.POINTER(ctypes.c_int)] * 2),
    # include/pnmol_sqrt.h
    "pnmol_qr_r": (ctypes.c_int, [_vp, _c_double_p, ctypes.c_int, ctypes.c_int, _c_double_p]),
    "pnmol_qr_last_ms": (ctypes.c_int, [ctypes.POINTER(ctypes.c_float)]),
    "pnmol_sqrt_propagate_cholesky_factor": (ctypes.c_int, [_vp, _c_double_p, ctypes.c_int, ctypes.c_int, _c_double_p,
                                                             ctypes.c_int, _c_double_p]),
    "pnmol_sqrt_update": (ctypes.c_int, [_vp, _c_double_p, ctypes.c_int, ctypes.c_int, _c_double_p, _c_double_p,
                                         _c_double_p, _c_double_p, _c_double_p]),
    "pnmol_sqrt_filter_create": (ctypes.c_int, [_vp, ctypes.POINTER(FilterDesc), ctypes.POINTER(_vp)]),
    "pnmol_sqrt_filter_destroy": (ctypes.c_int, [_vp]),
    "pnmol_sqrt_filter_set_state": (ctypes.c_int, [_vp, ctypes.c_double, _c_double_p, _c_double_p]),
    "pnmol_sqrt_filter_get_state": (ctypes.c_int, [_vp, _c_double_p, _c_double_p, _c_double_p]),
    "pnmol_sqrt_filter_predict_mean": (ctypes.c_int, [_vp, ctypes.c_double, _c_double_p]),
    "pnmol_sqrt_filter_set_operator": (ctypes.c_int, [_vp, _c_double_p, _c_double_p]),
    "pnmol_sqrt_filter_prepare_error_model": (ctypes.c_int, [_vp, ctypes.c_double]),
    "pnmol_sqrt_filter_step": (ctypes.c_int, [_vp, ctypes.c_double, ctypes.POINTER(StepOut), _c_double_p]),
    "pnmol_sqrt_filter_steps": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_double, _c_double_p, _c_double_p,
                                               ctypes.POINTER(StepOut)]),
    "pnmol_sqrt_filter_last_steps_ms": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_float)]),
    "pnmol_sqrt_update_no_meascov": (ctypes.c_int, [_vp, _c_double_p, ctypes.c_int, ctypes.c_int, _c_double_p,
                                                    _c_double_p, _c_double_p, _c_double_p]),
}

_lib = None


def load_library():
    """Load the shared library (no GPU needed for this) and declare all prototypes."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise PnmolHipError(
                f"{LIB_PATH} not found: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
                "pnmol has no CPU fallback for the filter step."
            )
        lib = ctypes.CDLL(str(LIB_PATH))
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def _dp(a):
    return a.ctypes.data_as(_c_double_p)


def _f64(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if shape is not None and a.shape != shape:
        raise ValueError(f"expected shape {shape}, got {a.shape}")
    return a


class Context:
    """One HIP device + stream (`pnmol_ctx`)."""

    _cache = {}

    def __init__(self, device=0):
        self.lib = load_library()
        n = ctypes.c_int(0)
        rc = self.lib.pnmol_device_count(ctypes.byref(n))
        if rc != 0 or n.value < 1:
            raise PnmolHipError("no HIP device visible: the PNMOL filter step needs an AMD GPU (MI355X / gfx950)")
        h = _vp()
        rc = self.lib.pnmol_ctx_create(int(device), ctypes.byref(h))
        if rc != 0:
            raise PnmolHipError(f"pnmol_ctx_create(device={device}) failed with {rc}")
        self.handle, self.device = h, int(device)

    @classmethod
    def default(cls, device=None):
        if device is None:
            device = int(os.environ.get("PNMOL_HIP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
            n = ctypes.c_int(0)
            load_library().pnmol_device_count(ctypes.byref(n))
            if n.value > 0:
                device %= n.value
        if device not in cls._cache:
            cls._cache[device] = cls(device)
        return cls._cache[device]

    def cholesky(self, A):
        """Lower Cholesky factor of a symmetric positive definite matrix on the device (`pnmol_cholesky_lower`):
        `jnp.linalg.cholesky(spatial_kernel(X, X.T))`, white.py:84-85."""
        A = _f64(A)
        n = A.shape[0]
        if A.shape != (n, n):
            raise ValueError(f"expected a square matrix, got {A.shape}")
        L = np.empty((n, n))
        self.check(self.lib.pnmol_cholesky_lower(self.handle, _dp(A), n, _dp(L)), "pnmol_cholesky_lower")
        return L

    def fd_solve_batched(self, gram, lk, llk):
        """weights (N, s), uncertainty (N,) of N kernel-FD stencils (`pnmol_fd_solve_batched`; discretize.py:177-201)."""
        gram, lk, llk = _f64(gram), _f64(lk), _f64(llk)
        N, s = lk.shape
        if gram.shape != (N, s, s) or llk.shape != (N,):
            raise ValueError(f"shapes {gram.shape}, {lk.shape}, {llk.shape} do not describe N stencils of size s")
        w, u = np.empty((N, s)), np.empty(N)
        self.check(self.lib.pnmol_fd_solve_batched(self.handle, _dp(gram), _dp(lk), _dp(llk), N, s, _dp(w), _dp(u)),
                   "pnmol_fd_solve_batched")
        return w, u

    def qr_r(self, A):
        """R factor (upper, diag >= 0) of A on the device: `jnp.linalg.qr(A, mode="r")` (base/sqrt.py:21)."""
        A = _f64(A)
        rows, cols = A.shape
        R = np.empty((cols, cols))
        self.check(self.lib.pnmol_qr_r(self.handle, _dp(A), rows, cols, _dp(R)), "pnmol_qr_r")
        return R

    def sqrt_propagate(self, S1, S2=None):
        S1 = _f64(S1)
        n, k1 = S1.shape
        out = np.empty((n, n))
        if S2 is None:
            rc = self.lib.pnmol_sqrt_propagate_cholesky_factor(self.handle, _dp(S1), n, k1, None, 0, _dp(out))
        else:
            S2 = _f64(S2)
            if S2.shape[0] != n:
                raise ValueError(f"S1 {S1.shape} and S2 {S2.shape} must have the same number of rows")
            rc = self.lib.pnmol_sqrt_propagate_cholesky_factor(self.handle, _dp(S1), n, k1, _dp(S2), S2.shape[1], _dp(out))
        self.check(rc, "pnmol_sqrt_propagate_cholesky_factor")
        return out

    def sqrt_update(self, H, C, meascov_sqrtm=None):
        H, C = _f64(H), _f64(C)
        m, D = H.shape
        if C.shape != (D, D):
            raise ValueError(f"cov_cholesky must be {(D, D)}, got {C.shape}")
        if m > D:
            raise ValueError("update_sqrt needs output_dim <= input_dim (base/sqrt.py:56-58)")
        C_new, gain, Sl = np.empty((D, D)), np.empty((D, m)), np.empty((m, m))
        if meascov_sqrtm is None:
            rc = self.lib.pnmol_sqrt_update_no_meascov(self.handle, _dp(H), m, D, _dp(C), _dp(C_new), _dp(gain), _dp(Sl))
        else:
            E = _f64(meascov_sqrtm, (m, m))
            rc = self.lib.pnmol_sqrt_update(self.handle, _dp(H), m, D, _dp(C), _dp(E), _dp(C_new), _dp(gain), _dp(Sl))
        self.check(rc, "pnmol_sqrt_update")
        return C_new, gain, Sl

    def qr_last_ms(self):
        ms = ctypes.c_float(0)
        self.lib.pnmol_qr_last_ms(ctypes.byref(ms))
        return ms.value

    def check(self, rc, what):
        if rc != 0:
            msg = self.lib.pnmol_last_error(self.handle)
            raise PnmolHipError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def synchronize(self):
        self.check(self.lib.pnmol_ctx_synchronize(self.handle), "pnmol_ctx_synchronize")


class Filter:
    """`pnmol_filter`: the measurement/prior model of one discretised PDE on the device."""

    DTYPES = {"f64": 0, "f32": 1}

    def __init__(self, ctx, *, L, B, E_sqrtm, R_sqrtm, Gamma, num_derivatives, dtype="f64", K=None):
        """dtype "f32": the covariance and its bulk kernels (predict, H-apply, down-date) in fp32; the factorisation of
        the innovation matrix, the mean and all scalars stay fp64 (`pnmol_filter_desc.dtype`)."""
        self.ctx, self.lib = ctx, ctx.lib
        d, ds = L.shape          # ds = d (white-noise model) or 2d (latent-force model, state [u; eps])
        nB = B.shape[0]
        self._keep = [_f64(L, (d, ds)), _f64(B, (nB, ds)), _f64(E_sqrtm, (d, d)), _f64(R_sqrtm, (nB, nB)),
                      _f64(Gamma, (ds, ds))]
        gram = _f64(self._keep[4] @ self._keep[4].T if K is None else K, (ds, ds))   # (BLAS here; a scalar loop in the library)
        desc = FilterDesc(d, int(num_derivatives), nB, *[_dp(a) for a in self._keep], ds, self.DTYPES[dtype], _dp(gram))
        self.dtype = dtype
        self.ds = ds
        h = _vp()
        ctx.check(self.lib.pnmol_filter_create(ctx.handle, ctypes.byref(desc), ctypes.byref(h)), "pnmol_filter_create")
        self.handle = h
        self.d, self.n, self.nB, self.m = d, int(num_derivatives) + 1, nB, d + nB
        self.error_model_dt = None
        # raw handles of the live pnmol_state objects of this filter, keyed by id(State): the C ABI refuses to destroy a
        # filter that still has states (include/pnmol_hip.h, "Lifetimes").  A plain dict, not weak references: the
        # cyclic collector clears weakrefs to unreachable objects BEFORE it runs finalisers, so a WeakSet would be empty
        # exactly when a filter and its states die together -- and the states' device buffers would leak.
        self._live = {}

    def __del__(self):
        # The cyclic garbage collector (and interpreter shutdown) finalises a filter and its states in ANY order: the
        # states' handles go first here; a State finalised later finds its entry gone and does nothing (State._destroy).
        live = getattr(self, "_live", None)
        while live:
            _, sh = live.popitem()
            self.lib.pnmol_state_destroy(sh)
        h, self.handle = getattr(self, "handle", None), None
        if h:
            self.lib.pnmol_filter_destroy(h)

    def sweep_layout(self):
        """(kernel, xcd_home) of this filter's sweep launch: `pnmol_filter_sweep_layout`."""
        k, x = ctypes.c_int(0), ctypes.c_int(0)
        self.lib.pnmol_filter_sweep_layout(self.handle, ctypes.byref(k), ctypes.byref(x))
        return {"kernel": ("per_panel", "k_sweep", "k_sweep_rl")[k.value], "xcd_home": x.value}

    def dims(self):
        v = [ctypes.c_int(0) for _ in range(5)]
        self.lib.pnmol_filter_dims(self.handle, *[ctypes.byref(x) for x in v])
        return dict(zip(("d", "n", "m", "dp", "mp"), (x.value for x in v)))

    def prepare_error_model(self, dt):
        """Step-invariant part of `estimate_error` for step size dt, computed on the device from the current operator."""
        self.ctx.check(self.lib.pnmol_filter_prepare_error_model(self.handle, float(dt)), "pnmol_filter_prepare_error_model")
        self.error_model_dt = float(dt)

    def set_error_model(self, dt, Sq_inv, Sq_diag):
        a, b = _f64(Sq_inv, (self.m, self.m)), _f64(Sq_diag, (self.m,))
        self.ctx.check(self.lib.pnmol_filter_set_error_model(self.handle, float(dt), _dp(a), _dp(b)),
                       "pnmol_filter_set_error_model")
        self.error_model_dt = float(dt)

    def new_state(self):
        return State(self)

    def predict_mean(self, state_in, dt):
        out = np.empty(self.d)
        self.ctx.check(self.lib.pnmol_filter_predict_mean(self.handle, state_in.handle, float(dt), _dp(out)),
                       "pnmol_filter_predict_mean")
        return out

    def set_operator(self, M, shift=None):
        a = _f64(M, (self.d, self.ds))
        b = _f64(shift, (self.d,)) if shift is not None else None
        self.ctx.check(self.lib.pnmol_filter_set_operator(self.handle, _dp(a), _dp(b) if b is not None else None),
                       "pnmol_filter_set_operator")
        self.error_model_dt = None

    def set_operator_diagonal(self, jdiag, shift=None):
        """M = L + diag(jdiag) (pointwise nonlinearity): `pnmol_filter_set_operator_diagonal`."""
        a = _f64(jdiag, (self.d,))
        b = _f64(shift, (self.d,)) if shift is not None else None
        self.ctx.check(self.lib.pnmol_filter_set_operator_diagonal(self.handle, _dp(a), _dp(b) if b is not None else None),
                       "pnmol_filter_set_operator_diagonal")
        self.error_model_dt = None

    def step(self, state_in, dt, want_error=True):
        out = State(self)
        info = StepOut()
        err = np.empty(self.d) if want_error else None
        rc = self.lib.pnmol_filter_step(self.handle, state_in.handle, float(dt), out.handle, ctypes.byref(info),
                                        _dp(err) if want_error else None)
        self.ctx.check(rc, "pnmol_filter_step")
        return out, info, err

    def steps(self, state, k, dt, want_means=True, want_stds=True):
        means = np.empty((k, self.d)) if want_means else None
        stds = np.empty((k, self.d)) if want_stds else None
        infos = (StepOut * k)()
        rc = self.lib.pnmol_filter_steps(self.handle, state.handle, int(k), float(dt),
                                         _dp(means) if want_means else None, _dp(stds) if want_stds else None, infos)
        self.ctx.check(rc, "pnmol_filter_steps")
        return means, stds, infos

    def steps_begin(self, state, k, dt):
        self.ctx.check(self.lib.pnmol_filter_steps_begin(self.handle, state.handle, int(k), float(dt)),
                       "pnmol_filter_steps_begin")
        self._pending_k = int(k)

    def steps_end(self, state, want_means=True, want_stds=True):
        k = self._pending_k
        means = np.empty((k, self.d)) if want_means else None
        stds = np.empty((k, self.d)) if want_stds else None
        infos = (StepOut * k)()
        rc = self.lib.pnmol_filter_steps_end(self.handle, state.handle, _dp(means) if want_means else None,
                                             _dp(stds) if want_stds else None, infos)
        self.ctx.check(rc, "pnmol_filter_steps_end")
        return means, stds, infos

    def prepare_steps(self, state, k, dt):
        self.ctx.check(self.lib.pnmol_filter_prepare_steps(self.handle, state.handle, int(k), float(dt)),
                       "pnmol_filter_prepare_steps")

    def last_steps_ms(self):
        ms = ctypes.c_float(0.0)
        self.lib.pnmol_filter_last_steps_ms(self.handle, ctypes.byref(ms))
        return float(ms.value)

    def debug_poison(self):
        self.ctx.check(self.lib.pnmol_filter_debug_poison(self.handle), "pnmol_filter_debug_poison")

    def debug_read(self, which, count):
        out = np.empty(int(count))
        self.ctx.check(self.lib.pnmol_filter_debug_read(self.handle, int(which), _dp(out), int(count)),
                       "pnmol_filter_debug_read")
        return out


class State:
    """`pnmol_state`: device-resident (t, mean, covariance) of one filter state."""

    def __init__(self, flt, _handle=None):
        self.filter, self.lib, self.ctx = flt, flt.lib, flt.ctx
        if _handle is None:
            _handle = _vp()
            self.ctx.check(self.lib.pnmol_state_create(flt.handle, ctypes.byref(_handle)), "pnmol_state_create")
        self.handle = _handle
        flt._live[id(self)] = _handle

    def _destroy(self):
        h, self.handle = getattr(self, "handle", None), None
        flt = getattr(self, "filter", None)
        # (no entry: Filter.__del__ ran first and has destroyed this state's handle already)
        if h and flt is not None and flt._live.pop(id(self), None) is not None:
            self.lib.pnmol_state_destroy(h)

    def __del__(self):
        self._destroy()

    def clone(self):
        h = _vp()
        self.ctx.check(self.lib.pnmol_state_clone(self.handle, ctypes.byref(h)), "pnmol_state_clone")
        return State(self.filter, h)

    def set(self, t, mean_nd, cov_DD):
        f = self.filter
        D = f.n * f.ds
        a, b = _f64(mean_nd, (f.n, f.ds)), _f64(cov_DD, (D, D))
        self.ctx.check(self.lib.pnmol_state_set(self.handle, float(t), _dp(a), _dp(b)), "pnmol_state_set")

    def set_sqrtm(self, t, mean_nd, cov_sqrtm_DD):
        """As `set`, from any square root C of the covariance; C C^T is formed on the device."""
        f = self.filter
        D = f.n * f.ds
        a, b = _f64(mean_nd, (f.n, f.ds)), _f64(cov_sqrtm_DD, (D, D))
        self.ctx.check(self.lib.pnmol_state_set_sqrtm(self.handle, float(t), _dp(a), _dp(b)), "pnmol_state_set_sqrtm")

    @property
    def t(self):
        t = ctypes.c_double(0.0)
        self.lib.pnmol_state_get_time(self.handle, ctypes.byref(t))
        return t.value

    def mean(self):
        out = np.empty((self.filter.n, self.filter.ds))
        self.ctx.check(self.lib.pnmol_state_get_mean(self.handle, _dp(out)), "pnmol_state_get_mean")
        return out

    def marginal_var(self):
        out = np.empty((self.filter.n, self.filter.ds))
        self.ctx.check(self.lib.pnmol_state_get_marginal_var(self.handle, _dp(out)), "pnmol_state_get_marginal_var")
        return out

    def cov_sqrtm(self):
        """Lower-triangular Cholesky factor of the covariance (device Cholesky), reference state order."""
        D = self.filter.n * self.filter.ds
        out = np.empty((D, D))
        self.ctx.check(self.lib.pnmol_state_get_cov_sqrtm(self.handle, _dp(out)), "pnmol_state_get_cov_sqrtm")
        return out

    def cov(self):
        D = self.filter.n * self.filter.ds
        out = np.empty((D, D))
        self.ctx.check(self.lib.pnmol_state_get_cov(self.handle, _dp(out)), "pnmol_state_get_cov")
        return out


class SqrtFilter:
    """`pnmol_sqrt_filter`: the white-noise EK1 step in square-root (QR) form, one device-resident state."""

    def __init__(self, ctx, *, L, B, E_sqrtm, R_sqrtm, Gamma, num_derivatives, dtype="f64"):
        """L (d, ds), B (nB, ds), Gamma (ds, ds) with ds = d (white-noise) or 2d (latent-force: [L, I], [B, 0],
        blockdiag(chol K, E_sqrtm), zero noise factors) -- the conventions of `Filter`.  dtype "f32": the QR (work
        matrices, reflectors, trailing updates) in fp32; the state, the mean path and all scalars stay fp64."""
        self.ctx = ctx
        self.dtype = dtype
        d, ds = L.shape
        nB = 0 if B is None else B.shape[0]
        self._keep = [_f64(L, (d, ds)), _f64(B if nB else np.zeros((0, ds))), _f64(E_sqrtm, (d, d)),
                      _f64(R_sqrtm if nB else np.zeros((0, 0))), _f64(Gamma, (ds, ds))]
        desc = FilterDesc(d=d, num_derivatives=int(num_derivatives), nB=nB, L=_dp(self._keep[0]),
                          B=_dp(self._keep[1]) if nB else None, E_sqrtm=_dp(self._keep[2]),
                          R_sqrtm=_dp(self._keep[3]) if nB else None, Gamma=_dp(self._keep[4]), d_state=ds,
                          dtype=Filter.DTYPES[dtype])
        h = _vp()
        ctx.check(ctx.lib.pnmol_sqrt_filter_create(ctx.handle, ctypes.byref(desc), ctypes.byref(h)),
                  "pnmol_sqrt_filter_create")
        self.handle, self.d, self.ds, self.n, self.m = h, d, ds, int(num_derivatives) + 1, d + nB

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.ctx.lib.pnmol_sqrt_filter_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def set_state(self, t, mean, cov_sqrtm):
        D = self.n * self.ds
        mean, C = _f64(mean, (self.n, self.ds)), _f64(cov_sqrtm, (D, D))
        self.ctx.check(self.ctx.lib.pnmol_sqrt_filter_set_state(self.handle, float(t), _dp(mean), _dp(C)),
                       "pnmol_sqrt_filter_set_state")

    def get_state(self, *, factor=True):
        D = self.n * self.ds
        t, mean = ctypes.c_double(0), np.empty((self.n, self.ds))
        C = np.empty((D, D)) if factor else None
        self.ctx.check(self.ctx.lib.pnmol_sqrt_filter_get_state(self.handle, ctypes.byref(t), _dp(mean),
                                                                 _dp(C) if factor else None), "pnmol_sqrt_filter_get_state")
        return t.value, mean, C

    def predict_mean(self, dt):
        out = np.empty(self.d)
        self.ctx.check(self.ctx.lib.pnmol_sqrt_filter_predict_mean(self.handle, float(dt), _dp(out)),
                       "pnmol_sqrt_filter_predict_mean")
        return out

    def set_operator(self, M, shift=None):
        M = _f64(M, (self.d, self.ds))
        sh = None if shift is None else _f64(shift, (self.d,))
        self.ctx.check(self.ctx.lib.pnmol_sqrt_filter_set_operator(self.handle, _dp(M), None if sh is None else _dp(sh)),
                       "pnmol_sqrt_filter_set_operator")

    def prepare_error_model(self, dt):
        self.ctx.check(self.ctx.lib.pnmol_sqrt_filter_prepare_error_model(self.handle, float(dt)),
                       "pnmol_sqrt_filter_prepare_error_model")

    def step(self, dt, want_error=False):
        info = StepOut()
        err = np.empty(self.d) if want_error else None
        self.ctx.check(self.ctx.lib.pnmol_sqrt_filter_step(self.handle, float(dt), ctypes.byref(info),
                                                           _dp(err) if want_error else None), "pnmol_sqrt_filter_step")
        return (info, err) if want_error else info

    def steps(self, k, dt):
        means, stds = np.empty((k, self.ds)), np.empty((k, self.ds))
        infos = (StepOut * k)()
        self.ctx.check(self.ctx.lib.pnmol_sqrt_filter_steps(self.handle, int(k), float(dt), _dp(means), _dp(stds), infos),
                       "pnmol_sqrt_filter_steps")
        return means, stds, list(infos)

    def last_steps_ms(self):
        ms = ctypes.c_float(0)
        self.ctx.lib.pnmol_sqrt_filter_last_steps_ms(self.handle, ctypes.byref(ms))
        return ms.value
