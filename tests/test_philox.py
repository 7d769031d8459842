"""The numpy Philox4x32-10 (tests/philox_np.py) against Random123's known-answer vectors (kat_vectors, philox4x32 10 rounds),
and the oracle's C generator against the numpy one through a reset: the uniforms behind the reference-executed reset /
resample / push / noise fixtures are exactly the ones the oracle and the kernels draw."""
import numpy as np

from tests import philox_np as ph

KAT = [  # counter, key, expected (Random123 kat_vectors: "philox4x32 10 ...")
    ((0x00000000,) * 4, (0x00000000,) * 2, (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_known_answer_vectors():
    for ctr, key, want in KAT:
        got = ph.philox4x32_10(np.array(ctr, np.uint64), np.array(key, np.uint64))
        assert tuple(int(x) for x in got) == want, (ctr, [hex(int(x)) for x in got])
    # vectorised call = element-wise calls
    c = np.array([k[0] for k in KAT], np.uint64)
    k = np.array([k[1] for k in KAT], np.uint64)
    assert np.array_equal(ph.philox4x32_10(c, k), np.array([k[2] for k in KAT], np.uint32))


def test_uniform_resolution_and_lane_addressing():
    u = ph.uniforms(1, np.arange(1000), 7, ph.DOF, 2)
    assert u.dtype == np.float32 and u.min() >= 0.0 and u.max() < 1.0
    assert np.all(u * 16777216.0 == np.floor(u * 16777216.0))          # 24-bit grid, like torch.rand
    assert abs(u.mean() - 0.5) < 0.02
    l = ph.lanes(1, [3, 9], 7, ph.DOF, 2, 7)                           # lanes 2..8 = block 0 words 2,3; block 1; block 2 word 0
    for r, e in enumerate((3, 9)):
        b0, b1, b2 = (ph.uniforms(1, e, 7, ph.DOF, b) for b in range(3))
        assert np.array_equal(l[r], np.concatenate((b0[2:], b1, b2[:1])))


def test_oracle_generator_is_this_philox(oracle_lib):
    """_reset_dofs through the oracle: dof_pos = q0 * (0.5 + u) with u = lanes 0..11 of purpose DOF (legged_robot.py:397-412)."""
    from tests.common import make_setup
    from oracle.oracle import OracleSim
    N, step = 37, 12345
    cfg, robot, p, names, model, w = make_setup("anymal_c_flat", N, seed=0x1234567890)
    o = OracleSim(p, model, robot, w)
    ids = np.arange(N, dtype=np.int32)
    o.reset_idx(ids, step)
    q0 = np.array(list(p.default_dof_pos)[:12], np.float32)
    u = ph.lanes(0x1234567890, ids, step, ph.DOF, 0, 12)
    want = q0 * ((np.float32(1.5) - np.float32(0.5)) * u + np.float32(0.5))
    np.testing.assert_allclose(o.dof_pos, want, rtol=0, atol=1e-7)
