"""Minimal policy interface used by the planner (API surface of the reference's
``policies/base.py:4-70``: what ``Sampler``, ``rollout`` and ``Trainer`` touch)."""

from ..utils.serializable import Serializable


def innermost_env(env):
    """Follow ``wrapped_env`` links (``policies/base.py:8-9``; note that the reference's
    ``NormalizedEnv`` keeps its env in ``_wrapped_env`` and is therefore NOT unwrapped - SURVEY.md N1)."""
    while hasattr(env, "wrapped_env"):
        env = env.wrapped_env
    return env


class Policy(Serializable):
    #: True when ``get_actions`` takes one observation per env (the planner overrides this)
    vectorized = False
    recurrent = False
    state_info_specs = ()

    def __init__(self, env):
        Serializable.quick_init(self, locals())
        self.env = innermost_env(env)

    # -- what a concrete policy implements ---------------------------------------------------
    def get_action(self, observation):
        raise NotImplementedError

    def get_actions(self, observations):
        raise NotImplementedError

    # -- hooks the sampler / trainer call; no-ops by default ----------------------------------
    def reset(self, dones=None):
        return None

    def log_diagnostics(self, paths, prefix=""):
        return None

    def terminate(self):
        return None

    # -- spaces --------------------------------------------------------------------------------
    observation_space = property(lambda self: self.env.observation_space)
    action_space = property(lambda self: self.env.action_space)
    state_info_keys = property(lambda self: [key for key, _ in self.state_info_specs])
