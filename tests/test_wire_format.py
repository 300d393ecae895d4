"""DVMW map wire format (include/dvmslam_wire.h): layout, host assembly / validation through the C ABI,
round trips, rejection of damaged blocks.  No GPU needed: these entry points are format logic, not compute."""
import numpy as np
import pytest

from wire_scene import assert_equal_delta, make_delta


@pytest.fixture(scope="module")
def wire():
    from dvm_slam_amd import wire as w
    return w


def test_layout_is_aligned_and_ordered(wire, capi):
    h = np.zeros(1, wire.HEADER)
    h["n_keyframes"], h["n_mappoints"], h["n_keypoints"], h["n_bow"], h["n_fv_nodes"], h["n_fv_feats"], h["n_links"], h["n_obs"] = 3, 7, 101, 33, 9, 40, 5, 11
    off, nbytes, total = wire.layout(h)
    assert nbytes == [64, 3 * 192, 7 * 160, 101 * 28, 101 * 32, 101 * 16, 33 * 4, 33 * 8, 9 * 8, 40 * 4, 5 * 24, 11 * 24]
    assert all(o % 64 == 0 for o in off) and total % 64 == 0
    for s in range(11):
        assert off[s + 1] >= off[s] + nbytes[s] and off[s + 1] - (off[s] + nbytes[s]) < 64
    assert total >= off[11] + nbytes[11]


@pytest.mark.parametrize("seed,n_kf,n_mp", [(0, 3, 40), (1, 1, 0), (2, 0, 5), (3, 12, 300), (4, 0, 0)])
def test_round_trip(wire, capi, seed, n_kf, n_mp):
    kfs, mps = make_delta(wire, capi, seed, n_kf, n_mp)
    blk = wire.build(kfs, mps, sender_agent=2)
    assert blk.size % 64 == 0
    parsed = wire.parse(blk)
    assert int(parsed[0]["sender_agent"]) == 2 and int(parsed[0]["total_bytes"]) == blk.size
    assert_equal_delta(wire, kfs, mps, parsed)
    # the head-only form is the prefix of the full block (what the sender uploads before the device gather)
    head = wire.build(kfs, mps, sender_agent=2, head_only=True)
    assert np.array_equal(head, blk[:head.size])


def test_damaged_blocks_are_rejected(wire, capi):
    kfs, mps = make_delta(wire, capi, 5, 3, 20)
    blk = wire.build(kfs, mps)
    wire.validate(blk)
    S = wire.sections(blk)

    def broken(mut):
        b = blk.copy()
        mut(wire.sections(b) if mut.__code__.co_argcount == 1 and mut.__name__ != "raw" else b)
        with pytest.raises(capi.DvmError):
            wire.validate(b)
    b = blk.copy(); b[0] ^= 1
    with pytest.raises(capi.DvmError):
        wire.validate(b)                                   # magic
    with pytest.raises(capi.DvmError):
        wire.validate(blk[:-64])                           # truncated
    with pytest.raises(capi.DvmError):
        wire.validate(blk[:32])
    b = blk.copy(); b[:64].view(wire.HEADER)["version"] = 9
    with pytest.raises(capi.DvmError):
        wire.validate(b)
    off, _, _ = wire.layout(blk[:64].view(wire.HEADER))
    for field, val in (("kp_off", 10 ** 6), ("n_kp", 10 ** 6), ("bow_off", 2 ** 31), ("n_links", 999), ("n_fv_nodes", 10 ** 5), ("n_levels", 1000)):
        b = blk.copy()
        b[off[1]:off[1] + 192].view(wire.KEYFRAME)[field] = val
        with pytest.raises(capi.DvmError):
            wire.validate(b)
    b = blk.copy()
    b[off[2]:off[2] + 160].view(wire.MAPPOINT)["n_obs"] = 10 ** 6
    with pytest.raises(capi.DvmError):
        wire.validate(b)
    kf = [k for k in range(len(kfs)) if len(kfs[k]["bow_ids"]) >= 2][0]
    b = blk.copy()
    ids = wire.sections(b)[6]
    o = int(S[1][kf]["bow_off"]); ids[o], ids[o + 1] = ids[o + 1], ids[o]        # BoW ids not ascending any more
    with pytest.raises(capi.DvmError):
        wire.validate(b)
    kf = [k for k in range(len(kfs)) if len(kfs[k]["fv"]["fv_feat"])][0]
    b = blk.copy()
    wire.sections(b)[9][int(S[1][kf]["fv_feat_off"])] = 10 ** 6                   # feature index beyond the keypoints
    with pytest.raises(capi.DvmError):
        wire.validate(b)
    # observations index the observing keyframe's keypoints: negative, or past the end of a keyframe travelling in the block
    obs = wire.sections(blk)[11]
    assert len(obs) > 0
    uu = {bytes(S[1][k]["uuid"].tobytes()): int(S[1][k]["n_kp"]) for k in range(len(kfs))}
    j = [i for i in range(len(obs)) if bytes(obs[i]["kf_uuid"].tobytes()) in uu][0]
    for val in (-1, uu[bytes(obs[j]["kf_uuid"].tobytes())], 10 ** 6):
        b = blk.copy()
        wire.sections(b)[11][j]["index"] = val
        with pytest.raises(capi.DvmError):
            wire.validate(b)
    b = blk.copy()
    wire.sections(b)[11][j]["index_right"] = -2
    with pytest.raises(capi.DvmError):
        wire.validate(b)
    # builder-side checks
    bad = [dict(k) for k in kfs]; bad[0]["bow_ids"] = bad[0]["bow_ids"][::-1].copy() if len(bad[0]["bow_ids"]) > 1 else np.array([5, 3], np.int32)
    bad[0]["bow_vals"] = np.zeros(len(bad[0]["bow_ids"]))
    with pytest.raises(capi.DvmError):
        wire.build(bad, mps)
