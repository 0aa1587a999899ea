"""CPU suite, part 2: the C-ABI library loads and exports what include/safelife_hip.h declares
(no compute calls without a GPU), and the host-side logic (level files, level pool constants)."""
import ctypes as C
import os
import subprocess
import re

import numpy as np
import pytest

from safelife_amd import _hip, levels
from safelife_amd.cell_types import CellTypes
from tests import util

REPO = util.REPO


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(REPO, "include", "safelife_hip.h")).read()
    declared = set(re.findall(r"\b(slhip_\w+)\s*\(", header))
    assert declared == set(_hip.EXPORTS), declared ^ set(_hip.EXPORTS)
    lib = _hip.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.slhip_abi_version() == _hip.SL_ABI_VERSION == 13


def _ctypes_layout(struct, prefix=""):
    out = []
    for name, ctype in struct._fields_:
        off = getattr(struct, name).offset
        if isinstance(ctype, type) and issubclass(ctype, C.Structure):
            out += [(prefix + name + "." + n, off + o, sz) for n, o, sz in _ctypes_layout(ctype)]
        else:
            out.append((prefix + name, off, C.sizeof(ctype)))
    return out


def test_env_struct_layout_matches_header(tmp_path):
    """ctypes mirrors vs the C structs as gcc lays them out: offset and size of every field."""
    structs = {"sl_env_batch": _hip.EnvBatch, "sl_wrappers": _hip.Wrappers, "sl_wrap_state": _hip.WrapState,
               "sl_pcg64": _hip.Pcg64, "sl_episode_record": _hip.EpisodeRecord, "sl_episode_queue": _hip.EpisodeQueue,
               "sl_agent_state": _hip.AgentState, "sl_multi_agent": _hip.MultiAgent}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "safelife_hip.h"', 'int main(void) {']
    want = []
    for cname, st in structs.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        want.append("%s %d" % (cname, C.sizeof(st)))
        for name, off, sz in _ctypes_layout(st):
            lines.append('printf("%s.%s %%zu %%zu\\n", offsetof(%s, %s), sizeof(((%s *)0)->%s));'
                         % (cname, name, cname, name, cname, name))
            want.append("%s.%s %d %d" % (cname, name, off, sz))
    lines += ['printf("scalars %zu level %zu out %zu\\n", sizeof(sl_env_scalars), sizeof(sl_level_scalars), '
              'sizeof(sl_step_out));', "return 0; }"]
    want.append("scalars 64 level 32 out 16")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = str(tmp_path / "layout")
    subprocess.check_call(["gcc", "-I", os.path.join(REPO, "include"), str(src), "-o", exe])
    got = subprocess.check_output([exe]).decode().split("\n")
    assert [g for g in got if g] == want
    # every field the header declares in sl_env_batch is mirrored (names, in order)
    header = open(os.path.join(REPO, "include", "safelife_hip.h")).read()
    start = header.index("typedef struct sl_env_batch {") + len("typedef struct sl_env_batch {")
    body = re.sub(r"/\*.*?\*/", "", header[start:header.index("} sl_env_batch;")], flags=re.S)
    names = []
    for stmt in body.split(";"):
        stmt = " ".join(stmt.split())
        if stmt:
            rest = re.match(r"^(const )?(\w+) ?(\*?)(.*)$", stmt).group(4)
            names += [re.sub(r"\[\w+\]|[\*\s]", "", part) for part in rest.split(",")]
    assert names == [f[0] for f in _hip.EnvBatch._fields_]


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from safelife_amd import speedups
    with pytest.raises(_hip.SafeLifeHipError):
        speedups.advance_board(np.zeros((5, 5), np.uint16))


def test_product_never_imports_oracle():
    pkg = os.path.join(REPO, "safelife_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(root, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "sl_oracle" not in text, f


def test_level_from_legacy_keys(tmp_path):
    board = np.zeros((7, 9), np.uint16)
    board[2, 5] = CellTypes.player
    board[4, 1] = CellTypes.level_exit
    board[0, 0] = CellTypes.level_exit | CellTypes.color_r
    goals = np.zeros_like(board)
    p = tmp_path / "lvl.npz"
    np.savez(p, board=board, goals=goals, agent_loc=np.array([5, 2]), orientation=np.array(3),
             spawn_prob=np.array(0.25), min_performance=np.array(0.5),
             **{"class": np.array("safelife.game_physics.SafeLifeGame")})
    (lv,) = levels.load_levels(str(p))
    assert lv.agent_locs.tolist() == [[2, 5]]                       # (x, y) -> (row, col)
    assert lv.board[2, 5] == CellTypes.player | (3 << 12)           # orientation applied
    assert lv.exit_locs.tolist() == [0, 4 * 9 + 1]                  # row-major, agent excluded
    assert lv.spawn_prob == 0.25 and lv.min_performance == 0.5


def test_level_archive(tmp_path):
    dt = np.dtype([("name", "U8"), ("board", np.uint16, (5, 5)), ("goals", np.uint16, (5, 5)),
                   ("agent_loc", np.int64, (2,)), ("orientation", np.int64), ("spawn_prob", float),
                   ("min_performance", float), ("class", "U40")])
    arr = np.zeros(3, dt)
    for i in range(3):
        arr[i]["name"] = "lv%d" % i
        arr[i]["board"][1, i] = CellTypes.player
        arr[i]["agent_loc"] = (i, 1)
    p = tmp_path / "arch.npz"
    np.savez(p, levels=arr)
    lvls = levels.load_levels(str(p))
    assert [lv.agent_locs.tolist() for lv in lvls] == [[[1, 0]], [[1, 1]], [[1, 2]]]
    assert lvls[2].name.endswith(os.path.join("arch", "lv2"))


@pytest.mark.parametrize("name", ["prune_still_25", "append_spawn_25", "append_still_26", "navigation_64"])
def test_pool_constants_match_reference(name):
    """required_points / initial_available_points of the reference's own procgen levels."""
    pool, ref = util.pool_from_fixture(name, util.oracle_counts)
    assert np.array_equal(pool.pool_required_reset, ref["required_points"])
    assert np.array_equal(pool.pool_required_step, ref["required_points"])
    for k, lv in enumerate(pool.levels):
        table = pool.points_table[pool.pool_table_idx[k]].astype(np.int64)
        avail = levels.available_points(table, pool.initial_counts[k], levels.initial_colors(lv.board))
        assert avail == ref["initial_available_points"][k]
    half, _ = util.pool_from_fixture(name, util.oracle_counts, n=8, min_performance_fraction=0.5)
    assert (half.pool_required_step <= half.pool_required_reset).all()


def test_pool_rng_follows_level_iterator_seeding():
    lv = [levels.Level(np.zeros((5, 5), np.uint16)) for _ in range(3)]
    pool = levels.LevelPool(lv, seed=np.random.SeedSequence(42), counts_fn=util.oracle_counts)
    kids = np.random.SeedSequence(42).spawn(3)
    for k in range(3):
        st = np.random.default_rng(kids[k]).bit_generator.state["state"]
        assert int(pool.pool_rng[k][1]) == st["state"] & ((1 << 64) - 1)


def test_emd_restatement_known_answers():
    """side_effects.earth_mover_distance (LP restatement of pyemd's EMD-hat; parity unpinned, see the
    module docstring) on cases with closed-form answers."""
    from safelife_amd.side_effects import earth_mover_distance as emd
    a, b = np.zeros((6, 7)), np.zeros((6, 7))
    assert emd(a, b) == 0.0
    a[1, 1] = b[1, 1] = 0.7
    assert emd(a, b) == 0.0
    b[:] = 0
    b[1, 3] = 0.7                                    # all mass two columns over
    assert abs(emd(a, b) - 0.7 * np.tanh(2 / 5)) < 1e-9
    assert abs(emd(a, b) - emd(b, a)) < 1e-12
    b[:] = 0
    b[1, 6] = 0.7                                    # torus: 5 columns right == 2 columns left ...
    # ... but the reference wraps only positive coordinate differences (min(dx, W - dx) with a signed
    # dx, side_effects.py:47-50), so its ground distance is direction dependent; restated as is
    assert abs(emd(b, a) - 0.7 * np.tanh(2 / 5)) < 1e-9
    assert abs(emd(a, b) - 0.7 * np.tanh(5 / 5)) < 1e-9
    assert abs(emd(b, a, wrap_x=False) - 0.7 * np.tanh(5 / 5)) < 1e-9
    a[4, 4] = 0.25                                   # extra mass: +penalty * |difference|
    assert abs(emd(b, a) - (0.7 * np.tanh(2 / 5) + 0.25)) < 1e-9
    assert abs(emd(b, a, metric="euclidean", tanh_scale=0) - (0.7 * 2 + 0.25)) < 1e-9


def test_cell_names_round_trip():
    from safelife_amd.side_effects import cell_name, name_to_cell
    assert cell_name(CellTypes.life | CellTypes.color_g) == "life-green"
    assert cell_name(CellTypes.spawner | CellTypes.color_r | CellTypes.color_g) == "spawner-yellow"
    assert cell_name(CellTypes.player) == "agent-gray"
    for name in ("life-green", "spawner-yellow", "crate-gray", "tree-white", "exit-red"):
        assert cell_name(name_to_cell(name)) == name
