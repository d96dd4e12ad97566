from .reward_spec import RewardSpec, reward_spec_for_env  # noqa: F401
from .synthetic_env import SyntheticEnv  # noqa: F401
