"""Import-path mirror of reference ``envs/anymal_c/mixed_terrains/anymal_c_rough_config.py``."""
from ...configs import AnymalCRoughCfg, AnymalCRoughCfgPPO  # noqa: F401
