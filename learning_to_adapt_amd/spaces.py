"""Minimal ``Box`` space: the planner only needs ``low``, ``high`` and ``shape``.

Stand-in for ``learning_to_adapt/spaces/box.py:5-73`` (the reference planner reads
``action_space.low/high/shape`` at ``policies/mpc_controller.py:68,76`` and the
dynamics models read ``observation_space.shape`` / ``action_space.shape`` at
``dynamics/mlp_dynamics.py:54-55``).  Any object with these three attributes works.
"""

import numpy as np


class Box(object):
    def __init__(self, low, high, shape=None):
        if shape is None:
            low = np.asarray(low, dtype=np.float64)
            high = np.asarray(high, dtype=np.float64)
            assert low.shape == high.shape
            self.low, self.high = low, high
        else:
            assert np.isscalar(low) and np.isscalar(high)
            self.low = low + np.zeros(shape)
            self.high = high + np.zeros(shape)

    @property
    def shape(self):
        return self.low.shape

    @property
    def flat_dim(self):
        return int(np.prod(self.low.shape))

    @property
    def bounds(self):
        return self.low, self.high

    def sample(self):
        return np.random.uniform(low=self.low, high=self.high, size=self.low.shape)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool((x >= self.low).all() and (x <= self.high).all())

    def __repr__(self):
        return "Box%s" % (self.shape,)

    def __eq__(self, other):
        return isinstance(other, Box) and np.allclose(self.low, other.low) and np.allclose(self.high, other.high)

    def __hash__(self):
        return hash((self.low.tobytes(), self.high.tobytes()))
