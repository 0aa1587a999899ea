"""
Generator of record for the compat tier (what ``safelife/random.py`` is to the reference): the numpy
Generator whose PCG64 stream the GPU ``speedups`` functions consume when no game-specific generator is
in force.  ``set_rng(gen)`` takes effect immediately and, used as a ``with`` block, steps back to the
previous generator on exit; every change is forwarded to ``speedups.set_bit_generator`` (which copies the
stream's state to the device for a call and writes it back advanced).
"""
import numpy as np

from . import speedups

_generators = [np.random.default_rng()]         # innermost last


def _activate():
    speedups.set_bit_generator(_generators[-1].bit_generator)


def get_rng():
    """The generator currently in force."""
    return _generators[-1]


class set_rng(object):
    """``set_rng(gen)`` or ``with set_rng(gen): ...``"""

    def __init__(self, generator):
        _generators.append(generator)
        self._level = len(_generators) - 1
        _activate()

    def __enter__(self):
        return _generators[-1]

    def __exit__(self, exc_type, exc, tb):
        del _generators[self._level:]
        _activate()
        return False


def coinflip(p, n=None):
    """Bernoulli(p) draw(s) from the generator in force: a bool, or an array of shape ``n``."""
    return get_rng().random(n) < p
