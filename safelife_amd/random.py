"""
Process-global random generator of the compat tier -- the counterpart of the reference's
``safelife/random.py:12-32``: ``set_rng(gen)`` installs a numpy Generator for the duration of a
``with`` block and hands its BitGenerator to ``speedups`` (which copies its PCG64 state to the
device and back, see speedups.py).
"""
import numpy as np

from . import speedups

random_gen = np.random.default_rng()


def get_rng():
    return random_gen


class set_rng(object):
    def __init__(self, new_rng):
        global random_gen
        self.old_rng = random_gen
        random_gen = new_rng
        speedups.set_bit_generator(random_gen.bit_generator)

    def __enter__(self):
        pass

    def __exit__(self, *args):
        global random_gen
        random_gen = self.old_rng
        speedups.set_bit_generator(random_gen.bit_generator)


def coinflip(p, n=None):
    return random_gen.random(n) < p
