"""Helper mirroring segmentation_pipeline/models/utils.py:4-9 (reference)."""
import inspect


def filter_kwargs(constructor, **kwargs):
    """Keep only the keyword arguments `constructor` accepts (used by ModularUNet to
    inject in_channels / out_channels / channels into plug-in up/down-samplers)."""
    accepted = inspect.signature(constructor).parameters
    return {name: value for name, value in kwargs.items() if name in accepted}


def is_sequence(x):
    """segmentation_pipeline/utils/utils.py:19-20 (reference)."""
    from collections.abc import Sequence
    return isinstance(x, Sequence) and not isinstance(x, str)
