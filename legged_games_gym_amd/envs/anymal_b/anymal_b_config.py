"""Import-path mirror of reference ``envs/anymal_b/anymal_b_config.py``."""
from ..configs import AnymalBRoughCfg, AnymalBRoughCfgPPO  # noqa: F401
