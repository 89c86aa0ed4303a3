// The reference's /inference call sequence (src/app.cpp:304-310) written against the drop-in shells, compiled with
// g++ -std=c++11 exactly like reference application code would be; dropin_db.cpp is the second translation unit (the reference
// links app.cpp + db.cpp, both of which include arcface.h).  Usage:
//   dropin_demo <det.frtw> <rec.frtw> <frame.bin (u8 BGR HWC)> <rows> <cols> <gallery.bin (fp32 [n][512]) | gallery.db (SQLite)> <n>
// Prints one line per face: x1 y1 x2 y2 score argmax sim   (argmax = gallery row of the best match)
// With a single argument "--selftest" it only exercises the no-GPU error paths.
#include <cstdio>
#include <cstdlib>
#include <fstream>

#include "dropin_db.h"
#include "frt/arcface.h"
#include "frt/retinaface.h"

static std::vector<char> slurp(const char *p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char **argv) {
    TRTLogger gLogger;
    if (argc == 2 && std::string(argv[1]) == "--selftest") {
        int ok = 0;
        try {
            RetinaFace det(gLogger, "/nonexistent.engine", 640, 480, "input_det", {"output_det0", "output_det1"}, {3, 288, 320}, 1, 4, 0.4f, 0.6f);
        } catch (const std::logic_error &e) {
            ok += std::string(e.what()) == "Cant find engine file";  // src/retinaface.cpp:53
        }
        try {
            ArcFaceIR50 rec(gLogger, "/nonexistent.engine", 640, 480, "input", "output", {3, 112, 112}, 512, 1, 4, 0.65f);
        } catch (const std::logic_error &e) {
            ok += std::string(e.what()) == "Cant find engine file";  // src/arcface.cpp:67
        }
        static_assert(sizeof(Bbox) == 20, "Bbox layout");
        // one process-wide classCount, shared by both translation units (src/arcface.h:39, src/arcface.cpp:19)
        ArcFaceIR50::classCount = 41;
        ok += Database::classCountSeenFromDbTU() == 41;
        ArcFaceIR50::classCount = 0;
        std::printf("selftest %s\n", ok == 3 ? "ok" : "FAILED");
        return ok == 3 ? 0 : 1;
    }
    if (argc == 8 && std::string(argv[1]) == "--recognize") {
        // POST /recognize, src/app.cpp:243-287: ONE face image already at the recogniser's input size, a Bbox that covers the whole frame
        // (x2 / y2 touch the far corner), rec_maxBatchSize 1, the recogniser constructed for the VIDEO frame size (src/app.cpp:55-57)
        //   dropin_demo --recognize <rec.frtw> <face.bin (u8 BGR 112x112x3)> <videoFrameWidth> <videoFrameHeight> <gallery.bin> <n>
        const int n = std::atoi(argv[7]);
        std::vector<char> fb = slurp(argv[3]), gb = slurp(argv[6]);
        std::vector<int> recInputShape = {3, 112, 112};
        ArcFaceIR50 recognizer(gLogger, argv[2], std::atoi(argv[4]), std::atoi(argv[5]), "input", "output", recInputShape, 512, 1, 4, 0.65f);
        recognizer.initKnownEmbeds(n);
        const float *g = reinterpret_cast<const float *>(gb.data());
        for (int i = 0; i < n; ++i) recognizer.addEmbedding(std::to_string(i), const_cast<float *>(g + (size_t)i * 512));
        recognizer.initMatMul();
        cv::Mat frame(112, 112, CV_8UC3, fb.data());
        std::vector<struct Bbox> outputBbox;
        for (int round = 0; round < 2; ++round) {  // twice: the handler clears its vectors and is called again
            Bbox bbox;
            bbox.x1 = 0;
            bbox.y1 = 0;
            bbox.x2 = recInputShape[1];
            bbox.y2 = recInputShape[2];
            bbox.score = 1;
            outputBbox.push_back(bbox);
            recognizer.forward(frame, outputBbox);
            float *output_sims = recognizer.featureMatching();
            std::vector<std::string> names;
            std::vector<float> sims;
            std::tie(names, sims) = recognizer.getOutputs(output_sims);
            if (names.size() != 1) return 7;
            std::printf("recognize %s %.9g %.9g\n", names[0].c_str(), sims[0], output_sims[std::atoi(names[0].c_str())]);
            outputBbox.clear();
        }
        return 0;
    }
    if (argc != 8) return 2;
    const int rows = std::atoi(argv[4]), cols = std::atoi(argv[5]), n = std::atoi(argv[7]);
    std::vector<char> fb = slurp(argv[3]), gb = slurp(argv[6]);
    cv::Mat frame(rows, cols, CV_8UC3, fb.data());
    // construction as in src/app.cpp:52-57
    RetinaFace detector(gLogger, argv[1], cols, rows, "input_det", {"output_det0", "output_det1"}, {3, rows, cols}, 1, 4, 0.4f, 0.6f);
    ArcFaceIR50 recognizer(gLogger, argv[2], cols, rows, "input", "output", {3, 112, 112}, 512, 1, 4, 0.65f);
    const std::string gpath(argv[6]);
    if (gpath.size() > 3 && gpath.compare(gpath.size() - 3, 3, ".db") == 0) {
        // start-up as in src/app.cpp:62,101-105: the database TU feeds the recogniser row by row from SQLite's blob buffer
        Database db(gpath, 512);
        if (db.getEmbeddings(recognizer) != 0) return 5;
        if (ArcFaceIR50::classCount != n || Database::classCountSeenFromDbTU() != n) return 6;
        recognizer.initMatMul();
        // /reload as in src/app.cpp:354-365 (must not leak or break the matcher)
        recognizer.resetEmbeddings();
        if (db.getEmbeddings(recognizer) != 0) return 5;
        recognizer.initMatMul();
    } else {
        // gallery load as in src/db.cpp:326-340
        recognizer.initKnownEmbeds(n);
        const float *g = reinterpret_cast<const float *>(gb.data());
        for (int i = 0; i < n; ++i) recognizer.addEmbedding(std::to_string(i), const_cast<float *>(g + (size_t)i * 512));
        recognizer.initMatMul();
    }
    // src/app.cpp:304-310
    std::vector<struct Bbox> outputBbox = detector.findFace(frame);
    if (outputBbox.empty()) {
        std::printf("No faces found\n");
        return 0;
    }
    recognizer.forward(frame, outputBbox);
    float *output_sims = recognizer.featureMatching();
    std::vector<std::string> names;
    std::vector<float> sims;
    std::tie(names, sims) = recognizer.getOutputs(output_sims);
    std::vector<std::string> names2;
    std::vector<float> sims2;
    std::tie(names2, sims2) = recognizer.matchTop1();
    for (size_t i = 0; i < outputBbox.size(); ++i) {
        if (names[i] != names2[i]) return 3;
        std::printf("%d %d %d %d %.9g %s %.9g\n", outputBbox[i].x1, outputBbox[i].y1, outputBbox[i].x2, outputBbox[i].y2, outputBbox[i].score,
                    names[i].c_str(), sims[i]);
    }
    try {  // empty gallery -> throw const char* (src/arcface.cpp:198)
        recognizer.resetEmbeddings();
        recognizer.featureMatching();
        return 4;
    } catch (const char *) {
    }
    return 0;
}
