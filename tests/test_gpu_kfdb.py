"""GPU parity of the KeyFrameDatabase place-recognition path (second half): dvm_bowdb_query against the
brute-force definition, and the host mirror dvm_host::KeyFrameDatabase (device word intersection + scores, host replay of
the reference's bookkeeping) against the inverted-file oracle through long mixed operation sequences.  Exact."""
import numpy as np
import pytest

from kfdb_scene import fill, make_db_scene

pytestmark = pytest.mark.gpu


def test_bowdb_query_raw(capi, oracle):
    kfs = make_db_scene(3, n_maps=2, kf_per_map=70)
    kfs.append(dict(ids=np.zeros(0, np.int32), vals=np.zeros(0, np.float64)))           # empty BowVector
    kfs.append(dict(ids=np.arange(0, 3000, 2, dtype=np.int32), vals=np.full(1500, 1 / 1500.0)))  # > 64 words per chunk, many hits
    q = kfs[7]
    common, first, score = capi.bowdb_query_raw([(k["ids"], k["vals"]) for k in kfs], q["ids"], q["vals"], erase=(3, 20))
    for j, k in enumerate(kfs):
        if j in (3, 20):
            assert common[j] == -1
            continue
        inter = np.intersect1d(q["ids"], k["ids"])
        assert common[j] == len(inter)
        assert first[j] == (inter[0] if len(inter) else -1)
        assert score[j] == np.float32(oracle.bow_score(q["ids"], q["vals"], k["ids"], k["vals"]))
    # empty query
    common, first, score = capi.bowdb_query_raw([(k["ids"], k["vals"]) for k in kfs[:5]], np.zeros(0, np.int32), np.zeros(0))
    assert np.all(common == 0) and np.all(first == -1) and np.all(score == 0)


def test_bowdb_culling_reclaims_words_and_keeps_answers(capi, oracle):
    """Keyframe culling: erased slots keep their numbers but stop costing query work, their words are reclaimed by
    a repack once they outnumber the live ones, and every answer stays what a store holding only the live keyframes gives."""
    rng = np.random.default_rng(5)
    db = capi.BowDb()
    bows = []
    for k in range(600):
        ids = np.sort(rng.choice(20000, 400, replace=False)).astype(np.int32)
        vals = rng.random(400); vals /= vals.sum()
        bows.append((ids, vals))
        assert db.add(ids, vals) == k
    q = bows[17]
    before = db.query(*q)
    assert db.stats() == dict(slots=600, live=600, words=240000, capacity=db.stats()["capacity"])
    dead = [k for k in range(600) if k % 4 != 1]
    for k in dead:
        db.erase(k)
    mid = db.query(*q)
    assert db.stats()["words"] == 240000 and db.stats()["live"] == 150          # erase alone reclaims nothing
    ids = np.sort(rng.choice(20000, 400, replace=False)).astype(np.int32); vals = np.full(400, 1 / 400.0)
    assert db.add(ids, vals) == 600                                               # ... the next add repacks
    st = db.stats()
    assert st["slots"] == 601 and st["live"] == 151 and st["words"] == 151 * 400 and st["capacity"] < 240000
    after = db.query(*q)
    live = np.array([k for k in range(600) if k % 4 == 1])
    for res in (mid, after):
        assert np.all(res[0][dead] == -1) and np.all(res[1][dead] == -1) and np.all(res[2][dead] == 0)
        for a, b in zip(res, before):
            assert np.array_equal(a[live], b[live])
    inter = np.intersect1d(q[0], ids)
    assert after[0][600] == len(inter) and after[2][600] == np.float32(oracle.bow_score(q[0], q[1], ids, vals))
    with pytest.raises(RuntimeError):
        db.query(q[0][::-1], q[1])                                                 # query ids must ascend (binary search)
    db.close()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_keyframe_database_sequences(capi, oracle, seed):
    kfs = make_db_scene(seed + 10)
    dbo = oracle.KeyFrameDatabase()
    dbg = capi.HostKeyFrameDatabase()
    fill(dbo, kfs); fill(dbg, kfs)
    rng = np.random.default_rng(seed)
    n = len(kfs)
    alive = set(range(n))
    hits = reloc_hits = 0
    for step in range(160):
        op = rng.random()
        j = int(rng.choice(sorted(alive)))
        q = kfs[j]
        if op < 0.35:
            m = int(rng.integers(0, 3))
            r_o = dbo.detect_merge_possibility(q["ids"], q["vals"], q["uuid"], m)
            r_g = dbg.detect_merge_possibility(q["ids"], q["vals"], q["uuid"], m)
            assert r_o == r_g, (step, r_o, r_g)
            hits += r_o[1] >= 0
        elif op < 0.55:
            m = int(rng.integers(0, 3))
            s0 = float(rng.choice([0.0, 0.5]))
            assert dbo.merge_score(q["ids"], q["vals"], q["uuid"], m, s0) == dbg.merge_score(q["ids"], q["vals"], q["uuid"], m, s0)
        elif op < 0.42 + 0.35:      # Tracking::Relocalization's query: a frame (any BoW vector, small id range -> repeated and zero ids occur)
            m = int(rng.integers(0, 3))
            fid = int(rng.integers(0, 12))
            co = dbo.detect_reloc(q["ids"], q["vals"], fid, m)
            cg = dbg.detect_reloc(q["ids"], q["vals"], fid, m)
            assert np.array_equal(co, cg), (step, fid, co, cg)
            reloc_hits += len(co) > 0
        elif op < 0.85:
            lo, mo = dbo.detect_n_best(j, 3)
            lg, mg = dbg.detect_n_best(j, 3)
            assert np.array_equal(lo, lg) and np.array_equal(mo, mg), (step, lo, lg, mo, mg)
            hits += len(lo) + len(mo) > 0
        elif op < 0.92 and len(alive) > 30:
            dbo.erase(j); dbg.erase(j); alive.discard(j)
        elif op < 0.96:
            b = bool(rng.integers(0, 2))
            dbo.set_bad(j, b); dbg.set_bad(j, b)
        else:   # a new keyframe arrives
            k = dict(kfs[j]); k["uuid"] = int(rng.integers(1, 2**62)); k["mn_id"] = len(kfs) + 1
            so = dbo.add(k["ids"], k["vals"], k["map_id"], k["uuid"], k["mn_id"]); sg = dbg.add(k["ids"], k["vals"], k["map_id"], k["uuid"], k["mn_id"])
            assert so == sg == len(kfs)
            for db in (dbo, dbg):
                db.set_neighbours(so, kfs[j]["neigh"]); db.set_connected(so, kfs[j]["connected"])
            k["neigh"], k["connected"] = kfs[j]["neigh"], kfs[j]["connected"]
            kfs.append(k); alive.add(so)
        if step % 20 == 0:
            for s in range(len(kfs)):
                assert dbo.state(s) == dbg.state(s), (step, s)
                assert dbo.reloc_state(s) == dbg.reloc_state(s), (step, s)
    assert hits > 20 and reloc_hits > 5
    for s in range(len(kfs)):
        assert dbo.state(s) == dbg.state(s) and dbo.reloc_state(s) == dbg.reloc_state(s)
