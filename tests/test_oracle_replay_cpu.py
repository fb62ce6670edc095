"""CPU: the decision-replay mechanism of the oracle (the basis of the tie-proof GPU parity tests, tests/parity_util.py).
A free-running refinement that records its own decisions, replayed with those decisions, must reproduce itself exactly;
edited decisions must change the result (the replay really is in control)."""
import numpy as np
import pytest

import parity_util as PU


@pytest.mark.parametrize("kw", [dict(n_iters=3), dict(solver=1, lambda0=1e-3, n_iters=5), dict(nparam=7, n_iters=3, w_dc=0.15),
                                dict(automask=0, n_iters=2, param=1)],
                         ids=["gn", "lm", "pose+scale+dc", "euler,noautomask"])
def test_replay_of_own_decisions_is_the_identity(oracle64, kw):
    from oracle.oracle import default_opts
    from tightly_coupled_sfm_amd import synth
    b = synth.make_batch(1, 24, 40, seed0=3)
    a = (b["tgt"][0], b["src"][0], b["depth_t"][0, 0], b["depth_s"][0, 0], b["pose_init"][0], b["K"][0])
    p0, l0, s0 = oracle64.refine(*a, default_opts(**kw), log_scale=0.02)
    p1, l1, s1, bits, dec = oracle64.refine_record(*a, default_opts(**kw), log_scale=0.02)
    assert np.array_equal(p0, p1) and np.array_equal(s0, s1)
    oracle64.flip_stats_reset()
    p2, l2, s2 = oracle64.refine(*a, default_opts(**kw), log_scale=0.02, bits=bits, decide=dec)
    assert np.array_equal(p0, p2) and l0 == l2 and np.array_equal(s0, s2)
    fn, fh = oracle64.flip_stats(kw["n_iters"])
    assert not fn.any() and not fh.any()          # replaying its own decisions, the oracle would have decided the same at EVERY linearisation
    n_it = kw["n_iters"]
    assert np.array_equal((bits[:n_it] & 1).sum((1, 2)), s0[:n_it, 2])           # bit 0 is the mask the cost was taken over
    assert bits.shape[0] == n_it + (1 if kw.get("solver") == 1 else 0)
    # the first-linearisation check of parity_util accepts the oracle's own decisions with zero flips
    o = type("O", (), dict(w_l1=0.15, w_ssim=0.85))
    ph = oracle64.photometric(*a[:4], a[4], a[5], log_scale=0.02 if kw.get("nparam") == 7 else 0.0)
    assert PU.check_first_masks(bits[0], ph, kw.get("automask", 1)) == 0


def _with(p, i, z):
    p = np.array(p, np.float64); p[i] = z
    return p


def test_replay_obeys_edited_decisions(oracle64):
    from oracle.oracle import default_opts
    from tightly_coupled_sfm_amd import synth
    b = synth.make_batch(1, 24, 40, seed0=3)
    a = (b["tgt"][0], b["src"][0], b["depth_t"][0, 0], b["depth_s"][0, 0], b["pose_init"][0], b["K"][0])
    kw = dict(solver=1, lambda0=1e-3, n_iters=5)
    p1, _, s1, bits, dec = oracle64.refine_record(*a, default_opts(**kw))
    drop = bits.copy(); drop[:, 5:12, 5:30] &= 0xFFFE                                  # take a block of pixels out of every mask
    oracle64.flip_stats_reset()
    p2, _, s2 = oracle64.refine(*a, default_opts(**kw), bits=drop, decide=dec)
    assert s2[0, 2] < s1[0, 2] and np.abs(p2 - p1).max() > 1e-6
    fn, fh = oracle64.flip_stats(5)               # ... which the flip statistics report at every linearisation, as hard (non-tie) flips:
    assert np.all(fn == s1[0, 2] - s2[0, 2]) or np.all(fn > 0.5 * (s1[0, 2] - s2[0, 2]))      # (later iterates: the masks have moved a little)
    assert np.all(fh > 0.8 * fn)                  # what a kernel that corrupts masks after the first pose update would look like
    inval = bits.copy(); inval[:, 5:12, 5:30] = 0                                 # ... and declare them invalid: their samples become zero,
    p3, _, s3 = oracle64.refine(*a, default_opts(**kw), bits=inval, decide=dec)   # which the SSIM windows of their neighbours see
    assert s3[0, 2] == s2[0, 2] and abs(s3[0, 0] - s2[0, 0]) > 1e-9
    rej = dec.copy(); rej[2] = 0                                                  # force a rejection at the third linearisation
    p4, _, s4 = oracle64.refine(*a, default_opts(**kw), bits=bits, decide=rej)
    assert s4[3, 3] > s1[3, 3] and np.abs(p4 - p1).max() > 1e-7                   # lambda went up instead of down
    # bilinear-cell parity bits only act on samples within 1e-4 px of a texel boundary: scrambling them everywhere else is a no-op,
    # flipping them AT such a sample moves that sample to the neighbouring cell (same value, other gradient)
    ix, iy = oracle64.sample_positions(a[2], a[4], a[5])
    near = (np.abs(ix - np.round(ix)) < 1e-4) | (np.abs(iy - np.round(iy)) < 1e-4)
    scr = bits.copy(); scr[0][~near] ^= 12
    p5, _, s5 = oracle64.refine(*a, default_opts(**kw), bits=scr, decide=dec)
    if not near.any():
        assert np.array_equal(p5, p1) and np.array_equal(s5, s1)
    # an exact tie by construction: bisect the x translation until the sample of one pixel sits on a texel boundary
    v0, u0 = 12, 20
    pose = np.asarray(a[4], np.float64).copy()
    f = lambda tx: oracle64.sample_positions(a[2], _with(pose, 0, tx), a[5])[0][v0, u0]
    lo, hi = -0.03, 0.03
    t_lo, t_hi = f(lo), f(hi)
    target = np.round((t_lo + t_hi) / 2)
    assert min(t_lo, t_hi) < target < max(t_lo, t_hi)
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        if (f(mid) - target) * (t_lo - target) > 0: lo = mid
        else: hi = mid
    pose = _with(pose, 0, 0.5 * (lo + hi))
    a2 = (a[0], a[1], a[2], a[3], pose, a[5])
    q1, _, r1, bt, dc = oracle64.refine_record(*a2, default_opts(n_iters=1))
    ixt = oracle64.sample_positions(a[2], pose, a[5])[0][v0, u0]
    assert abs(ixt - np.round(ixt)) < 1e-9
    fl = bt.copy(); fl[0, v0, u0] ^= 4
    q2, _, r2 = oracle64.refine(*a2, default_opts(n_iters=1), bits=fl, decide=dc)
    assert abs(r2[0, 0] - r1[0, 0]) < 1e-9 * r1[0, 0]          # the cost is continuous across the boundary ...
    assert np.abs(q2 - q1).max() > 1e-9                        # ... the step is not (another image gradient)
    far = bt.copy(); far[0, v0 + 3, u0 + 3] ^= 4               # a sample that is NOT on a boundary ignores the bit
    ixf = oracle64.sample_positions(a[2], pose, a[5])[0][v0 + 3, u0 + 3]
    if abs(ixf - np.round(ixf)) > 1e-3:
        q3, _, _ = oracle64.refine(*a2, default_opts(n_iters=1), bits=far, decide=dc)
        assert np.array_equal(q3, q1)
    want = PU.check_lm_decisions(s1, dec, default_opts(**kw))
    assert want == 0                                                              # the oracle's own decisions are what its costs imply
