"""Import-path mirror of reference ``envs/cassie/cassie_config.py``."""
from ..configs import CassieRoughCfg, CassieRoughCfgPPO  # noqa: F401
