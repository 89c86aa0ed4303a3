"""Gallery ingest from the reference's SQLite database.

Schema (``/root/reference/src/db.cpp:58-65``): ``USER(USR_ID, USR_NM)``, ``FACE(IMG_ID, USR_ID, IMG_PATH, EMBEDDING BLOB)``; the blob is the
raw little-endian ``float32[512]`` written by ``insertFace`` (``db.cpp:146``).  ``Database::getEmbeddings`` (``db.cpp:316-346``) reads
``SELECT * FROM FACE`` in rowid order and calls ``addEmbedding(USR_ID, blob)`` per row - this does the same in bulk.

    names, emb = load_gallery("test.db"); recognizer.setGallery(emb, names); recognizer.initMatMul()
"""
import sqlite3

import numpy as np


def load_gallery(db_path, dim=512):
    con = sqlite3.connect(db_path)
    try:
        rows = con.execute("SELECT USR_ID, EMBEDDING FROM FACE").fetchall()
    finally:
        con.close()
    names = [str(r[0]) for r in rows]
    emb = np.empty((len(rows), dim), np.float32)
    for i, r in enumerate(rows):
        v = np.frombuffer(r[1], dtype="<f4")
        if v.size != dim:
            raise ValueError("FACE row %d: embedding blob has %d floats, expected %d" % (i, v.size, dim))
        emb[i] = v
    return names, emb


def write_gallery(db_path, names, emb):
    """Create a database with the reference's schema (tests / synthetic galleries)."""
    con = sqlite3.connect(db_path)
    con.execute("CREATE TABLE IF NOT EXISTS USER (USR_ID TEXT PRIMARY KEY NOT NULL, USR_NM TEXT NOT NULL)")
    con.execute("CREATE TABLE IF NOT EXISTS FACE (IMG_ID INTEGER PRIMARY KEY AUTOINCREMENT, USR_ID TEXT NOT NULL, IMG_PATH TEXT NOT NULL, EMBEDDING BLOB NOT NULL)")
    for n in sorted(set(names)):
        con.execute("INSERT OR IGNORE INTO USER VALUES (?, ?)", (n, "user " + n))
    con.executemany("INSERT INTO FACE (USR_ID, IMG_PATH, EMBEDDING) VALUES (?, ?, ?)",
                    [(n, "img%d.jpg" % i, np.asarray(e, "<f4").tobytes()) for i, (n, e) in enumerate(zip(names, emb))])
    con.commit()
    con.close()
