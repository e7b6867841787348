"""The C++ oracle against the committed golden primary-hit fixtures (tests/golden/hits_*.npz).

The fixtures were computed by oracle/pyoracle.py from a Python tree mirroring the reference's
Geometry enum; here the same scene is built through the host library, flattened, and traced by
rr_oracle.cpp.  Agreement therefore checks the builder, the flattening (primitive ids) and the
traversal against an independent implementation.  CPU only.
"""
import numpy as np
import pytest

import golden_scenes as G
import oracle_lib as O


def compare_hits(got, gold, exact_t=True):
    assert np.array_equal(got["prim_id"], gold["prim_id"]), f"{int((got['prim_id'] != gold['prim_id']).sum())} ids differ"
    hit = gold["prim_id"] >= 0
    assert np.array_equal(got["front_face"][hit], gold["front_face"][hit])
    if exact_t:
        # equal_nan: a degenerate ray (zero direction) is "hit" at t = NaN by the reference's comparisons
        assert np.array_equal(got["t"][hit], gold["t"][hit], equal_nan=True)
        assert np.array_equal(got["point"][hit], gold["point"][hit], equal_nan=True)
    else:
        assert np.allclose(got["t"][hit], gold["t"][hit], rtol=1e-5, atol=0, equal_nan=True)
        ok = np.isfinite(gold["point"][hit]).all(axis=1)
        dp = np.linalg.norm(got["point"][hit][ok] - gold["point"][hit][ok], axis=1)
        assert (dp <= 1e-5 * np.maximum(np.linalg.norm(gold["point"][hit][ok], axis=1), 1.0)).all(), dp.max()
    assert np.allclose(got["normal"][hit], gold["normal"][hit], rtol=1e-5, atol=1e-7, equal_nan=True)
    # u, v of spheres go through acos / atan2: last-bit differences between math libraries
    assert np.allclose(got["u"][hit], gold["u"][hit], atol=1e-6, equal_nan=True)
    assert np.allclose(got["v"][hit], gold["v"][hit], atol=1e-6, equal_nan=True)


@pytest.mark.parametrize("name", sorted(G.SCENES))
def test_oracle_reproduces_golden_hits(built, name):
    fx = np.load(f"{G.GOLDEN_DIR}/hits_{name}.npz")
    A, flat = G.build_host(name)
    assert flat.n_spheres + flat.n_moving_spheres + flat.n_rects + flat.n_triangles + flat.n_media == int(fx["n_prims"])
    got = O.primary_hits(flat, fx["rays"], seed=int(fx["media_seed"]))
    # media free paths go through log10: t within 1e-5 relative instead of bit-exact
    compare_hits(got, fx["hits"], exact_t=(name != "media"))
    assert (fx["hits"]["prim_id"] >= 0).sum() > 50
