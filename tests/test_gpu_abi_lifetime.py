"""Lifetime rule of the C ABI (include/pnmol_hip.h, "Lifetimes"), exercised through raw ctypes calls -- the way a binding
other than pnmol/_hip.py would hit it.  Round 2 had a use-after-free here (pnmol_state_destroy dereferenced a filter that
had been freed); the rule now is: a destroy call on a parent with live children returns -1 and frees nothing.

No reference counterpart (the reference has no FFI): this pins the build's own boundary.
"""

import ctypes
import gc

import numpy as np
import pytest

from pnmol import _hip

pytestmark = pytest.mark.gpu


def _desc(d=12, nu=2, dtype=0):
    L = np.diag(np.full(d, -2.0)) + np.diag(np.ones(d - 1), 1) + np.diag(np.ones(d - 1), -1)
    B = np.zeros((2, d))
    B[0, 0] = B[1, -1] = 1.0
    keep = [np.ascontiguousarray(a, dtype=np.float64) for a in (L, B, 1e-3 * np.eye(d), np.zeros((2, 2)), np.eye(d))]
    desc = _hip.FilterDesc(d, nu, 2, *[_hip._dp(a) for a in keep], d, dtype)
    return desc, keep


def test_out_of_order_destroys_are_refused_not_crashes():
    lib = _hip.load_library()
    ctx = _hip._vp()
    assert lib.pnmol_ctx_create(0, ctypes.byref(ctx)) == 0
    desc, keep = _desc()
    flt = _hip._vp()
    assert lib.pnmol_filter_create(ctx, ctypes.byref(desc), ctypes.byref(flt)) == 0
    s1, s2 = _hip._vp(), _hip._vp()
    assert lib.pnmol_state_create(flt, ctypes.byref(s1)) == 0
    assert lib.pnmol_state_clone(s1, ctypes.byref(s2)) == 0

    # ctx before filter, filter before states: refused, with a message, and everything still works afterwards
    assert lib.pnmol_ctx_destroy(ctx) == -1
    assert b"still alive" in lib.pnmol_last_error(ctx)
    assert lib.pnmol_filter_destroy(flt) == -1
    assert b"2 state(s)" in lib.pnmol_last_error(ctx)
    t = ctypes.c_double(-1.0)
    assert lib.pnmol_state_get_time(s1, ctypes.byref(t)) == 0 and t.value == 0.0

    assert lib.pnmol_state_destroy(s1) == 0
    assert lib.pnmol_filter_destroy(flt) == -1          # one state left
    assert lib.pnmol_state_destroy(s2) == 0
    assert lib.pnmol_filter_destroy(flt) == 0
    # the square-root filter is a child of the ctx as well
    sq = _hip._vp()
    assert lib.pnmol_sqrt_filter_create(ctx, ctypes.byref(desc), ctypes.byref(sq)) == 0
    assert lib.pnmol_ctx_destroy(ctx) == -1
    assert lib.pnmol_sqrt_filter_destroy(sq) == 0
    assert lib.pnmol_ctx_destroy(ctx) == 0


def test_descriptor_dtype_is_validated():
    lib = _hip.load_library()
    ctx = _hip.Context.default(0)
    for bad in (2, -1, 7):
        desc, keep = _desc(dtype=bad)
        h = _hip._vp()
        assert lib.pnmol_filter_create(ctx.handle, ctypes.byref(desc), ctypes.byref(h)) == -1
        assert b"dtype" in lib.pnmol_last_error(ctx.handle)
    for bad in (2, -1, 7):        # the QR form: 0 = fp64, 1 = fp32 QR (include/pnmol_sqrt.h); anything else is refused
        desc, keep = _desc(dtype=bad)
        h = _hip._vp()
        assert lib.pnmol_sqrt_filter_create(ctx.handle, ctypes.byref(desc), ctypes.byref(h)) == -1
        assert b"dtype" in lib.pnmol_last_error(ctx.handle)
    desc, keep = _desc(dtype=1)
    h = _hip._vp()
    assert lib.pnmol_sqrt_filter_create(ctx.handle, ctypes.byref(desc), ctypes.byref(h)) == 0
    assert lib.pnmol_sqrt_filter_destroy(h) == 0


def test_python_mirror_frees_states_whatever_the_finalisation_order(hip_ctx):
    """A filter and its states that die in one cyclic-GC pass: no crash, no refused destroy left behind (the mirror keeps
    the raw state handles on the filter, so the filter's finaliser can destroy them first)."""
    desc, keep = _desc()
    L, B, E, R, G = keep
    flt = _hip.Filter(hip_ctx, L=L, B=B, E_sqrtm=E, R_sqrtm=R, Gamma=G, num_derivatives=2)
    states = [flt.new_state() for _ in range(3)]
    cyc = {"f": flt, "s": states}
    cyc["self"] = cyc            # unreachable cycle holding the filter and its states
    handles = [int(ctypes.cast(s.handle, ctypes.c_void_p).value) for s in states]
    assert len(flt._live) == 3 and all(handles)
    del flt, states, cyc
    gc.collect()
    # the ctx has no children left: a fresh ctx-level object count is not exposed, so check through a second filter whose
    # destroy must succeed immediately and through the explicit path State._destroy -> no entry -> no double free
    flt2 = _hip.Filter(hip_ctx, L=L, B=B, E_sqrtm=E, R_sqrtm=R, Gamma=G, num_derivatives=2)
    st = flt2.new_state()
    flt2.__del__()               # filter first: destroys the state's handle, then the filter
    assert flt2.handle is None and not flt2._live
    st._destroy()                # finds its entry gone: must not touch the library
    assert st.handle is None
