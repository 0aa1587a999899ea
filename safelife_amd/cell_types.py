"""
Cell bit layout of a SafeLife board (``uint16`` per cell).

Identical, bit for bit, to the reference's ``CellTypes`` (safelife/safelife_game.py:75-123) and to
the C enum in safelife/speedups_src/constants.h:4-33 -- boards, level files and observations are
interchangeable with the reference.
"""
import numpy as np

_u16 = np.uint16


class CellTypes(object):
    alive_bit = 0
    agent_bit = 1
    pushable_bit = 2
    destructible_bit = 3
    frozen_bit = 4
    preserving_bit = 5
    inhibiting_bit = 6
    spawning_bit = 7
    exit_bit = 8
    color_bit = 9          # three bits: red, green, blue
    orientation_bit = 12   # two bits
    pullable_bit = 15      # bit 14 is unused

    alive = _u16(1 << alive_bit)
    agent = _u16(1 << agent_bit)
    pushable = _u16(1 << pushable_bit)
    pullable = _u16(1 << pullable_bit)
    destructible = _u16(1 << destructible_bit)
    frozen = _u16(1 << frozen_bit)
    preserving = _u16(1 << preserving_bit)
    inhibiting = _u16(1 << inhibiting_bit)
    spawning = _u16(1 << spawning_bit)
    exit = _u16(1 << exit_bit)
    color_r = _u16(1 << color_bit)
    color_g = _u16(2 << color_bit)
    color_b = _u16(4 << color_bit)
    orientation_mask = _u16(3 << orientation_bit)

    # composites
    empty = _u16(0)
    freezing = inhibiting | preserving
    movable = pushable | pullable
    player = agent | freezing | frozen | destructible     # 122
    wall = frozen
    crate = frozen | movable
    spawner = frozen | spawning | destructible
    hard_spawner = frozen | spawning
    level_exit = frozen | exit
    life = alive | destructible
    colors = (color_r, color_g, color_b)
    rainbow_color = color_r | color_g | color_b
    ice_cube = frozen | freezing | movable
    plant = frozen | alive | movable
    tree = frozen | alive
    fountain = preserving | frozen
    parasite = inhibiting | alive | pushable | frozen
    weed = preserving | alive | pushable | frozen
    powers = alive | freezing | spawning


#: default points table, rows = goal colour (k r g y b m c w), columns = cell colour + "empty"
#: (safelife_game.py:595-605)
DEFAULT_POINTS_TABLE = np.array([
    [0, -1, 0, 0, 0, 0, 0, 0, 0],
    [-3, 3, -3, 0, -3, 0, -3, -3, 0],
    [0, -3, 5, 0, 0, 0, 3, 0, 0],
    [-3, 0, 0, 3, 0, 0, 0, 0, 0],
    [3, -3, 3, 0, 5, 3, 3, 3, 0],
    [-3, 3, -3, 0, -3, 5, -3, -3, 0],
    [3, -3, 3, 0, 3, 0, 5, 3, 0],
    [0, -1, 0, 0, 0, 0, 0, 0, 0],
], dtype=np.int64)
DEFAULT_POINTS_TABLE.setflags(write=False)
