"""Nested-class configuration base.

Behavioural mirror of reference ``legged_gym/envs/base/base_config.py:33-54``:
constructing a config object replaces every nested *class* attribute by an
*instance* of that class, recursively, so ``cfg.env.num_envs = 7`` mutates the
instance and not the class shared by every other config.
"""
import inspect


class BaseConfig:
    def __init__(self) -> None:
        BaseConfig.init_member_classes(self)

    @staticmethod
    def init_member_classes(obj) -> None:
        # dir() (alphabetical) is what the reference walks; the order is not
        # observable here but we keep it for exactness.
        for name in dir(obj):
            if name == "__class__":
                continue
            member = getattr(obj, name)
            if inspect.isclass(member):
                inst = member()
                setattr(obj, name, inst)
                BaseConfig.init_member_classes(inst)
