"""``Policy`` base - API surface of ``learning_to_adapt/policies/base.py:4-70``."""

from ..utils.serializable import Serializable


class Policy(Serializable):
    def __init__(self, env):
        Serializable.quick_init(self, locals())
        self.env = env
        while hasattr(self.env, 'wrapped_env'):
            self.env = self.env.wrapped_env

    def get_action(self, observation):
        raise NotImplementedError

    def get_actions(self, observations):
        raise NotImplementedError

    def reset(self, dones=None):
        pass

    @property
    def vectorized(self):
        """True when ``get_actions`` handles a batch of observations (one per env)."""
        return False

    @property
    def observation_space(self):
        return self.env.observation_space

    @property
    def action_space(self):
        return self.env.action_space

    @property
    def recurrent(self):
        return False

    def log_diagnostics(self, paths, prefix=''):
        pass

    @property
    def state_info_keys(self):
        return [k for k, _ in self.state_info_specs]

    @property
    def state_info_specs(self):
        return list()

    def terminate(self):
        pass
