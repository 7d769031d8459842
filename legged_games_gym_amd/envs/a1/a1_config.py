"""Import-path mirror of reference ``envs/a1/a1_config.py``."""
from ..configs import A1RoughCfg, A1RoughCfgPPO  # noqa: F401
