"""Step-size selection rules (reference: src/pnmol/odetools/step.py). Host-side scalars."""

import abc

import numpy as np


class StepRule(abc.ABC):
    @abc.abstractmethod
    def suggest(self, previous_dt, scaled_error_estimate, local_convergence_rate=None):
        raise NotImplementedError

    @abc.abstractmethod
    def is_accepted(self, scaled_error_estimate):
        raise NotImplementedError

    def scale_error_estimate(self, unscaled_error_estimate, reference_state):
        raise NotImplementedError

    def first_dt(self, discretized_pde):
        raise NotImplementedError


class Constant(StepRule):
    """Constant steps: always accept, error estimate unused (step.py:30-55)."""

    def __init__(self, dt):
        self.dt = dt
        self.min_step, self.max_step = 1e-15, 1e15

    def __repr__(self):
        return f"{self.__class__.__name__}(dt={self.dt})"

    def suggest(self, previous_dt, scaled_error_estimate, local_convergence_rate=None):
        return self.dt

    def is_accepted(self, scaled_error_estimate):
        return True

    def scale_error_estimate(self, unscaled_error_estimate, reference_state):
        return None

    def first_dt(self, discretized_pde):
        return self.dt


class Adaptive(StepRule):
    """Proportional control on the scaled error norm (step.py:58-119)."""

    def __init__(self, abstol=1e-4, reltol=1e-2, max_changes=(0.2, 10.0), safety_scale=0.95, min_step=1e-15,
                 max_step=1e15):
        self.abstol, self.reltol, self.max_changes, self.safety_scale = abstol, reltol, max_changes, safety_scale
        self.min_step, self.max_step = min_step, max_step

    def __repr__(self):
        return f"{self.__class__.__name__}(abstol={self.abstol}, reltol={self.reltol})"

    def suggest(self, previous_dt, scaled_error_estimate, local_convergence_rate=None):
        if local_convergence_rate is None:
            raise ValueError("Please provide a local convergence rate.")
        small, large = self.max_changes
        change = self.safety_scale * (1.0 / scaled_error_estimate) ** (1.0 / local_convergence_rate)
        return float(np.maximum(small, np.minimum(change, large))) * previous_dt

    def is_accepted(self, scaled_error_estimate):
        return scaled_error_estimate < 1

    def scale_error_estimate(self, unscaled_error_estimate, reference_state):
        err, ref = np.asarray(unscaled_error_estimate), np.asarray(reference_state)
        if err.ndim > 0 and err.shape != ref.shape:
            raise ValueError("Unscaled error estimate needs same shape as reference state.")
        ratio = np.atleast_1d(err / (self.abstol + self.reltol * ref))
        return np.linalg.norm(ratio) / np.sqrt(ratio.shape[0])

    def first_dt(self, discretized_pde):
        f = getattr(discretized_pde, "f", None)
        y0 = discretized_pde.y0
        dy0 = discretized_pde.L @ y0 if f is None else f(discretized_pde.t0, y0)
        return 0.01 * np.linalg.norm(y0) / np.linalg.norm(dy0)
