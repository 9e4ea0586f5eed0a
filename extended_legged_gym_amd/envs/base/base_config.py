"""Nested-class configuration base (API of the reference's `envs/base/base_config.py:33-54`).

A config is a class whose attributes are plain values or further nested classes.  Instantiating the outer class
replaces every nested class by an instance of it, recursively, so `cfg.env.num_envs = 64` edits one config object and
never the shared class.
"""
import inspect


class BaseConfig:
    def __init__(self) -> None:
        self.init_member_classes(self)

    @staticmethod
    def init_member_classes(obj):
        for name in dir(obj):
            if name == "__class__":
                continue
            member = getattr(obj, name)
            if inspect.isclass(member):
                inst = member()
                setattr(obj, name, inst)
                BaseConfig.init_member_classes(inst)
