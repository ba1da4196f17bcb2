"""PDE filter driver (reference: src/pnmol/pdefilter.py).

Host-side loop shell: time stepping, accept/reject, info counters, solution stacking.  The
per-step arithmetic is behind `attempt_step` (see pnmol.white).  Differences to the reference:
`PDESolution.cov_sqrtm` is derived lazily from the device-resident covariances (the reference
stacks (T+1) dense D x D factors eagerly, pdefilter.py:87,100), and `marginal_std` gives the
quantity experiments/figure1.py:76-80 reads out without forming any factor.
"""

from abc import ABC, abstractmethod
from collections import namedtuple
from typing import Iterable

import numpy as np

from . import kernels
from .odetools import step


class PDEFilterState(namedtuple("_", "t y error_estimate reference_state diffusion_squared_local")):
    """PDE filter state (pdefilter.py:17-22)."""


class PDESolution:
    """t (T+1,), mean (T+1,n,d), cov_sqrtm (T+1,D,D) [lazy], info, diffusion_squared_calibrated
    (pdefilter.py:25-31)."""

    def __init__(self, t, mean, ys, info, diffusion_squared_calibrated):
        self.t, self.mean, self.info = t, mean, info
        self.diffusion_squared_calibrated = diffusion_squared_calibrated
        self._ys = ys
        self._cov_sqrtm = None

    @property
    def cov_sqrtm(self):
        if self._cov_sqrtm is None:
            self._cov_sqrtm = np.stack([y.cov_sqrtm for y in self._ys])
        return self._cov_sqrtm

    @property
    def marginal_std(self):
        """sqrt(diag(cov)) as (T+1, n, d), uncalibrated."""
        out = []
        for y in self._ys:
            var = y.marginal_var if hasattr(y, "marginal_var") else \
                np.einsum("ij,ij->i", y.cov_sqrtm, y.cov_sqrtm).reshape(y.mean.shape, order="F")
            out.append(np.sqrt(np.maximum(var, 0.0)))
        return np.stack(out)


class PDEFilter(ABC):
    """Interface of the filtering-based PDE solvers (pdefilter.py:34-235)."""

    def __init__(self, *, steprule=None, num_derivatives=2, spatial_kernel=None, diffuse_prior_scale=1e0):
        self.steprule = steprule or step.Adaptive()
        self.num_derivatives = num_derivatives
        self.iwp = None
        self.spatial_kernel = spatial_kernel or kernels.Matern52() + kernels.WhiteNoise()
        self.E0 = None
        self.E1 = None
        self.diffuse_prior_scale = diffuse_prior_scale

    def __repr__(self):
        return (f"{self.__class__.__name__}(num_derivatives={self.num_derivatives}, steprule={self.steprule}, "
                f"spatial_kernel={self.spatial_kernel})")

    @staticmethod
    def _collect(diffusion_squared_list, state):
        if isinstance(state.diffusion_squared_local, list):
            diffusion_squared_list.extend(state.diffusion_squared_local)
        else:
            diffusion_squared_list.append(state.diffusion_squared_local)

    def solve(self, *args, **kwargs):
        times, means, ys, info, d2 = [], [], [], dict(), []
        for state, info in self.solution_generator(*args, **kwargs):
            times.append(state.t)
            means.append(state.y.mean)
            ys.append(state.y)
            self._collect(d2, state)
        return PDESolution(t=np.stack(times), mean=np.stack(means), ys=ys, info=info,
                           diffusion_squared_calibrated=np.mean(np.array(d2)))

    def simulate_final_state(self, *args, **kwargs):
        state, info, d2 = None, None, []
        for state, info in self.solution_generator(*args, **kwargs):
            self._collect(d2, state)
        cov_sqrtm_new = state.y.cov_sqrtm * np.sqrt(np.mean(np.array(d2)))
        return state._replace(y=state.y._replace(cov_sqrtm=cov_sqrtm_new)), info

    def solution_generator(self, pde, /, *, stop_at=None, progressbar=False):
        """Generate solver steps, starting with the initial state (pdefilter.py:118-165)."""
        time_stopper = _TimeStopper(stop_at) if stop_at is not None else None
        state = self.initialize(pde)
        info = dict(num_f_evaluations=0, num_df_evaluations=0, num_df_diagonal_evaluations=0, num_steps=0,
                    num_attempted_steps=0)
        yield state, info
        dt = self.steprule.first_dt(pde)
        pbar = None
        if progressbar:
            from tqdm import tqdm
            pbar = tqdm(total=100)
            threshold = increment = pde.tmax / 100
        while state.t < pde.tmax:
            if pbar is not None:
                while state.t + dt >= threshold:
                    pbar.update()
                    threshold += increment
                pbar.set_description(f"t={state.t:.4f}, dt={dt:.2E}")
            if time_stopper is not None:
                dt = time_stopper.adjust_dt_to_time_stops(state.t, dt)
            state, dt, step_info = self.perform_full_step(state, dt, pde)
            info["num_steps"] += 1
            for key in ("num_f_evaluations", "num_df_evaluations", "num_df_diagonal_evaluations",
                        "num_attempted_steps"):
                info[key] += step_info[key]
            yield state, info
        if pbar is not None:
            pbar.update()
            pbar.close()

    def perform_full_step(self, state, initial_dt, pde):
        """One accepted step incl. the accept/reject loop of the step rule (pdefilter.py:177-227)."""
        dt, accepted, proposed = initial_dt, False, None
        step_info = dict(num_f_evaluations=0, num_df_evaluations=0, num_df_diagonal_evaluations=0,
                         num_attempted_steps=0)
        while not accepted:
            proposed, attempt_info = self.attempt_step(state, dt, pde)
            step_info["num_attempted_steps"] += 1
            for key in ("num_f_evaluations", "num_df_evaluations", "num_df_diagonal_evaluations"):
                step_info[key] += attempt_info.get(key, 0)
            # NB the reference multiplies the (already dt-scaled) estimate by dt again (pdefilter.py:210)
            internal_norm = self.steprule.scale_error_estimate(
                unscaled_error_estimate=dt * proposed.error_estimate if proposed.error_estimate is not None else None,
                reference_state=proposed.reference_state)
            accepted = self.steprule.is_accepted(internal_norm)
            suggested_dt = self.steprule.suggest(dt, internal_norm, local_convergence_rate=self.num_derivatives + 1)
            dt = min(suggested_dt, pde.tmax - (proposed.t if accepted else state.t))
            assert dt >= 0, f"Invalid step size: dt={dt}"
        return proposed, dt, step_info

    @abstractmethod
    def initialize(self, pde):
        raise NotImplementedError

    @abstractmethod
    def attempt_step(self, state, dt, pde):
        raise NotImplementedError


class _TimeStopper:
    """Make the solver stop at specified time-points (pdefilter.py:238-256)."""

    def __init__(self, locations: Iterable):
        self._locations = iter(locations)
        self._next_location = next(self._locations)

    def adjust_dt_to_time_stops(self, t, dt):
        if t >= self._next_location:
            try:
                self._next_location = next(self._locations)
            except StopIteration:
                self._next_location = np.inf
        if t + dt > self._next_location:
            dt = self._next_location - t
        return dt
