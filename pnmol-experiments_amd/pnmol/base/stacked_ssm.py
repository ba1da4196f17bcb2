"""Stack of independent state-space models (reference: src/pnmol/base/stacked_ssm.py:7-80).

Host-side bookkeeping only: the device treats the stack [u; eps] of the latent-force filter as ONE IWP with
2d spatial components (same A, Q per component; the diffusions sit block-diagonally in the Gram), so none of
these dense block-diagonal matrices is formed on the hot path."""

import numpy as np
import scipy.linalg


class StackedSSM:
    def __init__(self, processes):
        self.processes = tuple(processes)
        self._dims = tuple(p.state_dimension for p in self.processes)

    @property
    def state_dimension(self):
        return sum(self._dims)

    def _stack(self, pairs):
        firsts, seconds = zip(*pairs)
        return scipy.linalg.block_diag(*firsts), scipy.linalg.block_diag(*seconds)

    @property
    def preconditioned_discretize(self):
        return self._stack([p.preconditioned_discretize for p in self.processes])

    def non_preconditioned_discretize(self, dt):
        return self._stack([p.non_preconditioned_discretize(dt) for p in self.processes])

    def nordsieck_preconditioner(self, dt):
        return self._stack([p.nordsieck_preconditioner(dt) for p in self.processes])

    def projection_to_process(self, process_to_project_onto):
        start = sum(self._dims[:process_to_project_onto])
        return np.eye(self.state_dimension)[start:start + self._dims[process_to_project_onto], :]

    def projection_matrix(self, derivative_to_project_onto, process_to_project_onto=None):
        if process_to_project_onto is None:
            return scipy.linalg.block_diag(*[p.projection_matrix(derivative_to_project_onto)
                                             for p in self.processes])
        assert isinstance(process_to_project_onto, int)
        proc = self.processes[process_to_project_onto]
        return proc.projection_matrix(derivative_to_project_onto) @ self.projection_to_process(process_to_project_onto)
