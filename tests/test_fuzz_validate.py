"""Seeded fuzzing of the host side of `rtr_upload_scene` (validation + scene compilation, no GPU):
scene files come from disk, so a corrupted record must end in an error code or in an accepted scene,
never in an out-of-bounds read, an endless loop or a crash of the caller's process.

The reference has no loader to compare with (its scenes are C++ builders); what is pinned
here is the boundary's own promise in include/rtr_hip.h ("validate + upload")."""
import numpy as np
import pytest

import _golden as G

A = G.A
rtr = G.rtr

INT_VALUES = [-2 ** 31, -7, -1, 0, 1, 2, 3, 5, 11, 31, 32, 33, 255, 4096, 10 ** 6, 2 ** 31 - 1]
FLOAT_VALUES = [0.0, -0.0, -1.0, 1e-300, -1e-300, 1e300, -1e300, np.inf, -np.inf, np.nan]


def _pick(rng, values):
    return values[int(rng.integers(0, len(values)))]


def _copy(sid):
    return rtr.Scene.from_bytes(G.scene(sid).to_bytes())


def _mutate(sc, rng):
    """One random field of one random record; returns a short description for the failure message."""
    tables = [("nodes", sc.nodes), ("materials", sc.materials), ("textures", sc.textures), ("lights", sc.lights),
              ("images", sc.images), ("perlin", sc.perlin)]
    tables = [(n, t) for n, t in tables if len(t)]
    pick = int(rng.integers(0, len(tables) + 2))
    if pick == len(tables):
        if len(sc.list_children):
            k = int(rng.integers(0, len(sc.list_children)))
            v = _pick(rng, INT_VALUES + [int(rng.integers(0, len(sc.nodes)))])
            sc.list_children[k] = np.int32(v)
            return "list_children[%d] = %d" % (k, v)
        pick = 0
    if pick == len(tables) + 1:
        v = _pick(rng, INT_VALUES + [int(rng.integers(0, len(sc.nodes)))])
        sc.root = v
        return "root = %d" % v
    name, table = tables[pick]
    k = int(rng.integers(0, len(table)))
    field = _pick(rng, list(table.dtype.names))
    col = table[field]
    slot = (k,) + tuple(int(rng.integers(0, d)) for d in col.shape[1:])
    if col.dtype.kind == "f":
        v = _pick(rng, FLOAT_VALUES)
    elif col.dtype.kind == "u":
        v = _pick(rng, [0, 1, 7, 2 ** 32, 2 ** 63, 2 ** 64 - 1, 2 ** 64 - 12])
    else:
        n_other = max(len(sc.nodes), len(sc.materials), len(sc.textures), 1)
        v = _pick(rng, INT_VALUES + [int(rng.integers(0, n_other))])
    col[slot] = col.dtype.type(v)
    return "%s[%s].%s = %r" % (name, slot, field, v)


@pytest.mark.parametrize("sid", [21, 23, 9, 35, 1, 30, 26])
def test_random_field_corruption_never_crashes_the_host(sid):
    rng = np.random.default_rng(1000 + sid)
    rounds = 60 if sid == 9 else 250  # scene 9 compiles 3 400 references per accepted mutation
    accepted = 0
    for it in range(rounds):
        sc = _copy(sid)
        what = [_mutate(sc, rng) for _ in range(int(rng.integers(1, 4)))]
        try:
            info = rtr.native.validate_scene(sc)
        except rtr.native.RtrError as e:
            assert e.code in (A.RTR_ERR_INVALID, A.RTR_ERR_UNSUPPORTED), (what, e.code, e.message)
            assert e.message, what
            continue
        accepted += 1
        assert info["stack_words"] >= 1, what
    assert 0 < accepted < rounds  # both outcomes occur: the mutations are neither all harmless nor all fatal


def test_truncated_and_garbled_scene_files_raise():
    blob = G.scene(21).to_bytes()
    rng = np.random.default_rng(5)
    for cut in [0, 7, 8, 47, 48, 200, len(blob) // 2, len(blob) - 1]:
        with pytest.raises(ValueError):
            rtr.Scene.from_bytes(blob[:cut])
    with pytest.raises(ValueError):
        rtr.Scene.from_bytes(blob + b"\0")
    for _ in range(200):  # flip bytes of the header (counts, root): parse error or a scene that validation judges
        b = bytearray(blob)
        for k in rng.integers(8, 48, size=int(rng.integers(1, 4))):
            b[int(k)] = int(rng.integers(0, 256))
        try:
            sc = rtr.Scene.from_bytes(bytes(b))
        except (ValueError, MemoryError, OverflowError):
            continue
        try:
            rtr.native.validate_scene(sc)
        except rtr.native.RtrError:
            pass
