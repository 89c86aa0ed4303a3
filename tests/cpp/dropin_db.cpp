// See dropin_db.h.  SQLite is a runtime library in this image (libsqlite3.so.0, no header under /usr/include), so the handful of
// C API entry points used here are declared by hand; the schema is the reference's (src/db.cpp:58-65): FACE(IMG_ID, USR_ID,
// IMG_PATH, EMBEDDING BLOB = raw little-endian float32[dim]).
#include "dropin_db.h"

#include <iostream>

extern "C" {
struct sqlite3_stmt;
int sqlite3_open(const char *, sqlite3 **);
int sqlite3_close(sqlite3 *);
int sqlite3_prepare_v2(sqlite3 *, const char *, int, sqlite3_stmt **, const char **);
int sqlite3_step(sqlite3_stmt *);
int sqlite3_finalize(sqlite3_stmt *);
int sqlite3_column_int(sqlite3_stmt *, int);
int sqlite3_column_bytes(sqlite3_stmt *, int);
const unsigned char *sqlite3_column_text(sqlite3_stmt *, int);
const void *sqlite3_column_blob(sqlite3_stmt *, int);
const char *sqlite3_errmsg(sqlite3 *);
}
namespace {
const int kSqliteOk = 0, kSqliteRow = 100, kSqliteDone = 101;
}

Database::Database(const std::string &path, int embedDim) : m_db(nullptr), m_dim(embedDim) {
    if (sqlite3_open(path.c_str(), &m_db) != kSqliteOk) throw std::logic_error("Can't open database");
}

Database::~Database() { sqlite3_close(m_db); }

int Database::getNumEmbeddings() {
    sqlite3_stmt *stmt = nullptr;
    if (sqlite3_prepare_v2(m_db, "SELECT COUNT(*) FROM FACE;", -1, &stmt, nullptr) != kSqliteOk) {
        std::cout << "SQL error: " << sqlite3_errmsg(m_db);
        return -1;
    }
    int n = -2;
    if (sqlite3_step(stmt) == kSqliteRow) n = sqlite3_column_int(stmt, 0);
    sqlite3_finalize(stmt);
    return n;
}

int Database::getEmbeddings(ArcFaceIR50 &recognizer) {
    const int numEmbeds = getNumEmbeddings();
    std::cout << "[INFO] There are " << numEmbeds << " embeddings in database\n";
    if (numEmbeds < 0) return -1;
    recognizer.initKnownEmbeds(numEmbeds);
    sqlite3_stmt *stmt = nullptr;
    if (sqlite3_prepare_v2(m_db, "SELECT * FROM FACE;", -1, &stmt, nullptr) != kSqliteOk) {
        std::cout << "SQL error: " << sqlite3_errmsg(m_db);
        return -1;
    }
    int rc;
    while ((rc = sqlite3_step(stmt)) == kSqliteRow) {
        const std::string userId(reinterpret_cast<const char *>(sqlite3_column_text(stmt, 1)));
        if (sqlite3_column_bytes(stmt, 3) != m_dim * (int)sizeof(float)) {
            sqlite3_finalize(stmt);
            return -3;
        }
        // the pointer is into SQLite's own buffer and dies at the next step(): addEmbedding must copy before returning
        float *embedding = const_cast<float *>(static_cast<const float *>(sqlite3_column_blob(stmt, 3)));
        recognizer.addEmbedding(userId, embedding);
    }
    sqlite3_finalize(stmt);
    return rc == kSqliteDone ? 0 : -2;
}

int Database::classCountSeenFromDbTU() { return ArcFaceIR50::classCount; }
