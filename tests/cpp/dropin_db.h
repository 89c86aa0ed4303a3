// Second translation unit of the drop-in link test: plays the role of /root/reference/src/db.h + db.cpp, which include arcface.h
// (src/db.h:8) next to src/app.cpp (src/app.cpp:3) - two TUs that both see `static int ArcFaceIR50::classCount`.  Only the reading
// half the hot path needs is here (Database::getNumEmbeddings / getEmbeddings, src/db.cpp:283-346), written against the SQLite C API.
#ifndef DROPIN_DB_H
#define DROPIN_DB_H

#include <string>

#include "frt/arcface.h"

struct sqlite3;

class Database {
  public:
    Database(const std::string &path, int embedDim);
    ~Database();
    int getNumEmbeddings();                      // src/db.cpp:283-314
    int getEmbeddings(ArcFaceIR50 &recognizer);  // src/db.cpp:316-346
    static int classCountSeenFromDbTU();         // ODR check: must be the same object as in the main TU

  private:
    sqlite3 *m_db;
    int m_dim;
};

#endif
