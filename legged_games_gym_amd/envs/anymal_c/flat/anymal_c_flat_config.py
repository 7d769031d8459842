"""Import-path mirror of reference ``envs/anymal_c/flat/anymal_c_flat_config.py``."""
from ...configs import AnymalCFlatCfg, AnymalCFlatCfgPPO  # noqa: F401
