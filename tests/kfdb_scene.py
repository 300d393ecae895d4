"""Synthetic multi-map keyframe database for the place-recognition tests: keyframes along a trajectory share words with
their neighbours (overlapping word windows), two maps revisit the same places (merge candidates)."""
import numpy as np


def make_db_scene(seed=0, n_maps=3, kf_per_map=50, n_words=3000, words_per_kf=160):
    rng = np.random.default_rng(seed)
    kfs = []
    for m in range(n_maps):
        start = rng.integers(0, 200) if m else 0       # the maps revisit the same places
        for k in range(kf_per_map):
            centre = (start + 23 * k) % n_words
            local = (centre + rng.integers(-220, 220, words_per_kf)) % n_words
            noise = rng.integers(0, n_words, words_per_kf // 5)
            ids, cnt = np.unique(np.concatenate([local, noise]), return_counts=True)
            vals = cnt * rng.uniform(0.5, 2.0, len(ids))
            vals = vals / vals.sum()          # L1-normalised TF-IDF-like weights
            kfs.append(dict(ids=ids.astype(np.int32), vals=vals.astype(np.float64), map_id=m, uuid=int(rng.integers(1, 2**62)),
                            mn_id=len(kfs) + 1, k=k))
    for i, kf in enumerate(kfs):
        same = [j for j, o in enumerate(kfs) if o["map_id"] == kf["map_id"] and j != i]
        same.sort(key=lambda j: (abs(kfs[j]["k"] - kf["k"]), j))
        kf["neigh"] = np.array(same[:10], np.int32)
        kf["connected"] = np.array(same[:14], np.int32)
    return kfs


def fill(db, kfs):
    for kf in kfs:
        s = db.add(kf["ids"], kf["vals"], kf["map_id"], kf["uuid"], kf["mn_id"])
        assert s == kf["mn_id"] - 1
    for s, kf in enumerate(kfs):
        db.set_neighbours(s, kf["neigh"])
        db.set_connected(s, kf["connected"])
