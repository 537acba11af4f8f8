"""Pin the oracle against the reference's own known-answer vectors
(UnitTests/TriangleHitTest.cpp:173-340) -- bit-exact, both arithmetic modes, both epsilons."""
import numpy as np
import pytest


def bits(x):
    return int(np.float32(x).view(np.uint32))


@pytest.mark.parametrize("contract", [0, 1])
@pytest.mark.parametrize("eps_mode", [0, 1])
def test_triangle_hit_kats(orc, kats, contract, eps_mode):
    for c in kats:
        ray = orc.ray_make(c["origin"], c["dir"], True, contract)      # rt::Ray( o, d, true )
        hit, t, u, v = orc.hit_triangle(ray, c["a"], c["b"], c["c"], contract, eps_mode)
        assert hit == c["hit"], c["name"]
        if not hit:
            continue
        assert bits(t) == int(c["t_bits"], 16), c["name"]
        assert bits(u) == bits(c["u"]) and bits(v) == bits(c["v"]), c["name"]
        n = orc.triangle_normal(c["a"], c["b"], c["c"], contract)      # ASSERT_EQ( n, ... )
        assert np.array_equal(n, np.array(c["normal"], np.float32)), c["name"]
        p = orc.ray_point(ray, t, contract)                            # EXPECT_EQ( hitpoint, ... )
        assert np.array_equal(p, np.array(c["hitpoint"], np.float32)), c["name"]


def test_kat_count(kats):
    assert len(kats) == 8 and sum(c["hit"] for c in kats) == 3
