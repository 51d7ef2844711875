"""Nested-class configuration objects (reference envs/base/base_config.py:34-55): instantiating a config turns
every nested class attribute into an instance, recursively, so `cfg.env.num_envs` is writable per object."""
import inspect


def namespace(section_, base_=None, /, **fields):
    """Build one nested config class; `base_` lets a task override a parent section field-by-field."""
    return type(section_, (base_,) if base_ is not None else (), dict(fields))


class BaseConfig:
    def __init__(self):
        self._instantiate_sections(self)

    @staticmethod
    def _instantiate_sections(node):
        for key in dir(node):
            if key == "__class__":
                continue
            value = getattr(node, key)
            if inspect.isclass(value):
                inst = value()
                setattr(node, key, inst)
                BaseConfig._instantiate_sections(inst)

    # the reference spells it init_member_classes; keep the name callable for user code
    init_member_classes = _instantiate_sections
