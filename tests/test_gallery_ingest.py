"""SQLite FACE.EMBEDDING blob ingest (the format src/db.cpp writes) - host logic only."""
import os
import sys

import numpy as np

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_sqlite_gallery_roundtrip(synth, tmp_path):
    import gallery_sqlite as gs
    emb = synth.make_gallery(37)
    names = ["u%d" % (i % 5) for i in range(37)]   # several faces per user, like the reference's FACE table
    db = str(tmp_path / "test.db")
    gs.write_gallery(db, names, emb)
    n2, e2 = gs.load_gallery(db)
    assert n2 == names and np.array_equal(e2, emb)   # rowid order, raw little-endian float32[512]
