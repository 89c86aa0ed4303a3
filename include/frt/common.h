// Drop-in for /root/reference/src/common.h (+ common.cpp) without NvInfer.h / cublasLt.h: same names, same behaviour.
//   Bbox, Paths                 src/common.h:13-21
//   fileExists, getFilePaths    src/common.cpp:3-41
//   checkCudaStatus / checkCublasStatus -> checkFrtStatus (status codes of libfrt; same throw: std::logic_error)
//   TRTLogger                   src/common.h:28-53 (kept so that constructor signatures stay source compatible)
#ifndef FRT_COMMON_H
#define FRT_COMMON_H

#include <dirent.h>

#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../frt.h"

struct Bbox {
    int x1, y1, x2, y2;  // x = row, y = column (src/retinaface.cpp:165)
    float score;
};
static_assert(sizeof(Bbox) == sizeof(frt_bbox) && sizeof(Bbox) == 20, "Bbox must be layout-identical to frt_bbox");

struct Paths {
    std::string absPath;
    std::string className;
};

inline bool fileExists(const std::string &name) {
    std::ifstream f(name.c_str());
    return f.good();
}

// <root>/<class>/<file>.jpg walk used by the offline gallery generator (src/app.cpp "gen" mode); not on the hot path.
inline void getFilePaths(std::string rootPath, std::vector<struct Paths> &paths) {
    DIR *root = opendir(rootPath.c_str());
    if (!root) return;
    const std::string ext = ".jpg";
    while (struct dirent *cls = readdir(root)) {
        const std::string classDir = rootPath + "/" + cls->d_name;
        DIR *cd = opendir(classDir.c_str());
        if (!cd) continue;
        while (struct dirent *fe = readdir(cd)) {
            const std::string fn(fe->d_name);
            if (fe->d_type == DT_DIR || fn.size() < ext.size() || fn.compare(fn.size() - ext.size(), ext.size(), ext) != 0) continue;
            Paths p;
            p.className = cls->d_name;
            p.absPath = classDir + "/" + fn;
            paths.push_back(p);
        }
        closedir(cd);
    }
    closedir(root);
}

// The reference prints to std::cerr and throws std::logic_error (src/common.cpp:43-55); a missing engine file throws
// std::logic_error("Cant find engine file") without the prefix (src/retinaface.cpp:53).
inline void checkFrtStatus(int status) {
    if (status == FRT_OK) return;
    const std::string msg = frt_last_error();
    if (status == FRT_ERR_NOT_FOUND) throw std::logic_error(msg);
    std::cerr << "FRT API failed with status " << status << ": " << msg << std::endl;
    throw std::logic_error("FRT API failed");
}

namespace nvinfer1 {  // only the logger vocabulary the reference's application code mentions
class ILogger {
  public:
    enum class Severity { kINTERNAL_ERROR = 0, kERROR = 1, kWARNING = 2, kINFO = 3, kVERBOSE = 4 };
    virtual void log(Severity severity, const char *msg) noexcept = 0;
    virtual ~ILogger() {}
};
}  // namespace nvinfer1

class TRTLogger : public nvinfer1::ILogger {
  public:
    void log(nvinfer1::ILogger::Severity severity, const char *msg) noexcept override {
        static const char *tag[] = {"INTERNAL_ERROR: ", "ERROR: ", "WARNING: ", "INFO: ", "VERBOSE: "};
        const int s = static_cast<int>(severity);
        std::cerr << ((s >= 0 && s <= 4) ? tag[s] : "UNKNOWN: ") << msg << std::endl;
    }
};

#endif  // FRT_COMMON_H
