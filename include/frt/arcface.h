// Drop-in for /root/reference/src/arcface.h: getCroppedFaces, struct CroppedFace and class ArcFaceIR50 with the reference's
// public surface (src/arcface.h:11-39).  Differences, all deliberate:
//   * featureMatching() returns an object-owned, reused buffer (the reference leaks new float[F*N] per call, arcface.cpp:194);
//   * forward() with rec_maxBatchSize >= 2 implements the evident intent (the reference's batched branch overflows m_embed);
//   * initKnownEmbeds() frees the previous gallery (the reference leaks it on every /reload);
//   * crop + resize + normalise run in one HIP kernel; CroppedFace.face still holds the u8 BGR 112x112 crop (app.cpp:329)
//     and CroppedFace.faceMat the fp32 planar RGB tensor, like after the reference's preprocessFaces().
#ifndef FRT_ARCFACE_H
#define FRT_ARCFACE_H

#include <algorithm>
#include <array>
#include <cassert>
#include <cstring>
#include <iterator>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include <tuple>

#include "coalesce.h"
#include "common.h"
#include "cvlite.h"
#include "matmul.h"

struct CroppedFace {
    cv::Mat face;     // u8 BGR crop
    cv::Mat faceMat;  // getCroppedFaces: same u8 crop; after ArcFaceIR50::forward: CV_32FC1 [3*112 x 112] planar RGB
    int x1, y1, x2, y2;
};

// src/arcface.cpp:3-17 (ROI = cols [y1,y2) x rows [x1,x2); INTER_CUBIC to resize_w x resize_h)
inline void getCroppedFaces(cv::Mat frame, std::vector<struct Bbox> &outputBbox, int resize_w, int resize_h, std::vector<struct CroppedFace> &croppedFaces) {
    croppedFaces.clear();
    const int n = (int)outputBbox.size();
    if (!n) return;
    std::vector<unsigned char> crops((size_t)n * resize_h * resize_w * 3);
    checkFrtStatus(frt_crop_faces(frame.data, frame.rows, frame.cols, (size_t)frame.step, reinterpret_cast<const frt_bbox *>(outputBbox.data()), n,
                                  resize_w, resize_h, crops.data(), -1));
    for (int i = 0; i < n; ++i) {
        CroppedFace c;
        c.faceMat = cv::Mat(resize_h, resize_w, CV_8UC3, &crops[(size_t)i * resize_h * resize_w * 3]).clone();
        c.face = c.faceMat.clone();
        c.x1 = outputBbox[i].x1;
        c.y1 = outputBbox[i].y1;
        c.x2 = outputBbox[i].x2;
        c.y2 = outputBbox[i].y2;
        croppedFaces.push_back(c);
    }
}

// `static int classCount` of the reference (src/arcface.h:39, defined once in src/arcface.cpp:19).  This shell is header-only and
// the reference includes arcface.h from two translation units (src/app.cpp:3 and src/db.cpp via src/db.h:8), so the definition
// has to be ODR-safe in C++11 (no inline variables): a static data member of a class TEMPLATE may be defined in a header, and
// ArcFaceIR50 inherits it - `ArcFaceIR50::classCount` / `recognizer.classCount` keep working unchanged.
template <class Tag = void>
struct ArcFaceIR50Statics {
    static int classCount;  // process-wide, as in the reference
};
template <class Tag>
int ArcFaceIR50Statics<Tag>::classCount = 0;

class ArcFaceIR50 : public ArcFaceIR50Statics<> {
  public:
    ArcFaceIR50(TRTLogger gLogger, const std::string engineFile, int frameWidth, int frameHeight, std::string inputName, std::string outputName,
                std::vector<int> inputShape, int outputDim, int maxBatchSize, int maxFacesPerScene, float knownPersonThreshold, int device = 0)
        : croppedFaces(this), h_(nullptr), m_id(frtdetail::nextObjectId()), m_frameWidth(frameWidth), m_frameHeight(frameHeight), m_OUTPUT_D(outputDim),
          m_maxBatchSize(maxBatchSize), m_maxFacesPerScene(maxFacesPerScene), m_knownPersonThresh(knownPersonThreshold), matmul(device) {
        (void)gLogger;
        (void)inputName;
        (void)outputName;
        assert(inputShape.size() == 3);  // src/arcface.cpp:25
        m_INPUT_C = inputShape[0];
        m_INPUT_H = inputShape[1];
        m_INPUT_W = inputShape[2];
        // FRT_COALESCE=<frames>: the recogniser must be able to take the faces of a coalesced batch in one pass
        const int envFrames = frtdetail::coalesceEnv().frames;
        const int devBatch = std::max(maxBatchSize, envFrames * maxFacesPerScene);
        checkFrtStatus(frt_embedder_create(engineFile.c_str(), m_INPUT_C, m_INPUT_H, m_INPUT_W, outputDim, devBatch, device, &h_));
        std::cout << "[INFO] Loading ArcFace Engine...\n";
        croppedFaces.reserve((size_t)maxFacesPerScene);
        m_device = device;
        m_devBatch = devBatch;
        if (envFrames > 0) {
            m_auto = true;
            try {
                frtdetail::autoLink(false, frtdetail::Pending{device, frameWidth, frameHeight, devBatch, nullptr, h_, matmul.handle(), &m_link});
            } catch (...) {
                frt_embedder_destroy(h_);
                throw;
            }
        }
    }
    ~ArcFaceIR50() {
        frtdetail::autoUnlink(false, &m_link);
        if (std::shared_ptr<frtdetail::CoalesceLink> l = link()) l->shutdown();  // the coalescer borrows this object's embedder and matcher (waits for calls inside it)
        frt_embedder_destroy(h_);
        frtdetail::retireObject(m_id);  // every thread drops its per-thread state of this object (page-locked similarity buffers) on its next access
    }
    // Opt-in request coalescing (include/frt/coalesce.h): findFace() calls of concurrent request threads on `detector` share one device
    // batch - detector, crop, recogniser, top-1 - and this object's forward() / featureMatching() / getOutputs() on the same thread, frame
    // and boxes answer from what that batch computed.  maxFrames 0: the detector's maxBatchSize (construct both objects with room:
    // detector maxBatchSize >= maxFrames, recogniser maxBatchSize >= maxFrames * maxFacesPerScene).  FRT_COALESCE=<frames>[:<window_us>]
    // in the environment does all of this without a code change.
    template <class Detector>
    void coalesceWith(Detector &detector, int maxFrames = 0, int windowUs = 100) {
        std::shared_ptr<frtdetail::CoalesceLink> l = frtdetail::makeLink(detector.handle(), h_, matmul.handle(), maxFrames, windowUs);
        if (std::shared_ptr<frtdetail::CoalesceLink> old = link()) old->shutdown();
        std::atomic_store(&m_link, l);
        detector.attachCoalescer(l);
    }
    // Extension (round 5): fp32 end-to-end recogniser (BASELINE configs[1]'s "fp32"; see frt_embedder_set_precision).  Default: fp16 MFMA.
    void setPrecisionFp32(bool on) { checkFrtStatus(frt_embedder_set_precision(h_, on ? 1 : 0)); }
    ArcFaceIR50(const ArcFaceIR50 &) = delete;
    ArcFaceIR50 &operator=(const ArcFaceIR50 &) = delete;

    // src/arcface.cpp:105-114: BGR->RGB, (x-127.5)*0.0078125, planar; output becomes CV_32FC1 [3*H x W]
    void preprocessFace(cv::Mat &face, cv::Mat &output) {
        cv::Mat tight = face.isContinuous() ? face : face.clone();
        output = cv::Mat(3 * m_INPUT_H, m_INPUT_W, CV_32FC1);
        checkFrtStatus(frt_embedder_preprocess_face(h_, tight.data, output.ptr<float>(0)));
    }
    void doInference(float *input, float *output) { checkFrtStatus(frt_embedder_infer(h_, input, 1, output)); }                          // arcface.cpp:131-137
    void doInference(float *input, float *output, int batchSize) { checkFrtStatus(frt_embedder_infer(h_, input, batchSize, output)); }  // arcface.cpp:139-148

    // Gallery management (src/arcface.cpp:150-164, :233-236).  The reference keeps a host array m_knownEmbeds (new float[num*D], never
    // freed) and uploads it in initMatMul; here every row goes straight into libfrt's pinned staging chunks and on to the device
    // while the caller (Database::getEmbeddings, src/db.cpp:316-346) fetches the next one - no second 2 GB host copy, no leak on
    // /reload, and the previous gallery stays searchable until initMatMul() swaps the new one in.
    void addEmbedding(const std::string className, float embedding[]) {  // arcface.cpp:150-154 (copies immediately: db.cpp:339 passes a blob pointer)
        matmul.galleryAppend(embedding, 1);
        classNames.push_back(className);
        classCount++;
    }
    void addEmbedding(const std::string className, std::vector<float> embedding) {  // arcface.cpp:156-160
        assert((int)embedding.size() == m_OUTPUT_D);
        matmul.galleryAppend(embedding.data(), 1);
        classNames.push_back(className);
        classCount++;
    }
    // Extension: bulk enrolment (one call for n rows; db.cpp:339's loop collapses to this)
    void addEmbeddings(const std::vector<std::string> &names, const float *embeddings) {
        matmul.galleryAppend(embeddings, (int)names.size());
        classNames.insert(classNames.end(), names.begin(), names.end());
        classCount += (int)names.size();
    }
    void initKnownEmbeds(int num) { matmul.galleryBegin(num, m_OUTPUT_D); }  // arcface.cpp:162
    void initMatMul() { matmul.galleryCommit(); }                            // arcface.cpp:164
    void resetEmbeddings() {                                                  // arcface.cpp:233-236
        classCount = 0;
        classNames.clear();
    }

    // src/arcface.cpp:166-187
    void forward(cv::Mat image, std::vector<struct Bbox> outputBbox) {
        const int n = (int)outputBbox.size();
        State &s = st();
        s.croppedFaces.clear();
        s.top_valid = s.from_record = false;
        s.embeds.assign((size_t)std::max(n, 1) * m_OUTPUT_D, 0.f);
        if (!n) return;
        if (forwardFromRecord(s, image, outputBbox)) return;  // a coalesced findFace() of this thread already computed everything
        std::vector<unsigned char> crops((size_t)n * m_INPUT_H * m_INPUT_W * 3);
        checkFrtStatus(frt_embedder_forward(h_, image.data, image.rows, image.cols, (size_t)image.step, reinterpret_cast<const frt_bbox *>(outputBbox.data()), n,
                                            s.embeds.data(), crops.data()));
        for (int i = 0; i < n; ++i) {
            CroppedFace c;
            c.face = cv::Mat(m_INPUT_H, m_INPUT_W, CV_8UC3, &crops[(size_t)i * m_INPUT_H * m_INPUT_W * 3]).clone();
            preprocessFace(c.face, c.faceMat);
            c.x1 = outputBbox[i].x1;
            c.y1 = outputBbox[i].y1;
            c.x2 = outputBbox[i].x2;
            c.y2 = outputBbox[i].y2;
            croppedFaces.push_back(c);
        }
    }
    // Optional alignment mode (no reference counterpart): forward() with the 5-point similarity warp instead of the bbox crop.
    void forwardAligned(cv::Mat image, std::vector<struct Bbox> outputBbox, const std::vector<std::array<float, 10>> &landmarks) {
        const int n = (int)landmarks.size();
        State &s = st();
        s.croppedFaces.clear();
        s.top_valid = s.from_record = false;
        s.embeds.assign((size_t)std::max(n, 1) * m_OUTPUT_D, 0.f);
        if (!n) return;
        std::vector<unsigned char> crops((size_t)n * m_INPUT_H * m_INPUT_W * 3);
        checkFrtStatus(frt_embedder_forward_aligned(h_, image.data, image.rows, image.cols, (size_t)image.step, landmarks[0].data(), n, s.embeds.data(),
                                                    crops.data()));
        for (int i = 0; i < n; ++i) {
            CroppedFace c;
            c.face = cv::Mat(m_INPUT_H, m_INPUT_W, CV_8UC3, &crops[(size_t)i * m_INPUT_H * m_INPUT_W * 3]).clone();
            preprocessFace(c.face, c.faceMat);
            c.x1 = c.y1 = c.x2 = c.y2 = 0;
            if ((size_t)i < outputBbox.size()) {
                c.x1 = outputBbox[(size_t)i].x1;
                c.y1 = outputBbox[(size_t)i].y1;
                c.x2 = outputBbox[(size_t)i].x2;
                c.y2 = outputBbox[(size_t)i].y2;
            }
            croppedFaces.push_back(c);
        }
    }
    // src/arcface.cpp:189-201.  Throws a const char* exactly like the reference (handlers catch const char*, app.cpp:276,341).
    // The [F x N] matrix lands in a page-locked buffer (one DMA instead of a staged pageable copy: 4 MB per face at N = 1M), and the
    // row-wise first maximum is computed on the device in the same call - the same accumulators, bit for bit - so that getOutputs()
    // on THIS buffer is O(F) instead of the reference's O(F*N) host scan.  setMaterializeSimilarities(false) skips the matrix
    // altogether for callers that, like src/app.cpp:309-310, only ever hand the pointer on to getOutputs (the buffer then holds
    // stale values; default: on, the reference's contract).
    float *featureMatching() {
        State &s = st();
        if (classNames.size() > 0 && s.croppedFaces.size() > 0) {
            const size_t n = s.croppedFaces.size();
            const size_t need = std::max<size_t>(m_materialize ? n * (size_t)classCount * sizeof(float) : 0, sizeof(float));
            if (need > s.out_cap) {
                frt_pinned_free(s.out);
                s.out = nullptr;
                s.out_cap = 0;
                void *p = nullptr;
                checkFrtStatus(frt_pinned_alloc(need, matmul.device(), &p));
                s.out = static_cast<float *>(p);
                s.out_cap = need;
            }
            // coalesced request, matrix not wanted: the batch's match stage already holds this frame's first maxima
            if (s.from_record && s.top_valid && !m_materialize) return s.out;
            s.top_idx.resize(n);
            s.top_sim.resize(n);
            s.top_valid = false;
            matmul.calculateTop1(s.embeds.data(), (int)n, m_materialize ? s.out : nullptr, s.top_idx.data(), s.top_sim.data());
            s.top_valid = true;
        } else {
            throw "Feature matching: No faces in database or no faces found";
        }
        return s.out;
    }
    void setMaterializeSimilarities(bool on) { m_materialize = on; }
    // src/arcface.cpp:203-217: first maximum per row (std::max_element), no threshold here
    std::tuple<std::vector<std::string>, std::vector<float>> getOutputs(float *output_sims) {
        std::vector<std::string> names;
        std::vector<float> sims;
        State &s = st();
        float *const m_out = s.out;
        const std::vector<int> &m_top_idx = s.top_idx;
        const std::vector<float> &m_top_sim = s.top_sim;
        bool fast = output_sims == m_out && s.top_valid && m_top_idx.size() == croppedFaces.size();  // the matrix featureMatching() just produced
        // the device maxima are only a shortcut for LOCAL, in-range rows: "no row wins" (-1: a row of NaNs - std::max_element answers 0 there)
        // and indices shifted by a shard's row offset go through the host scan below (or answer row 0 when no matrix was materialised)
        for (size_t i = 0; fast && i < m_top_idx.size(); ++i)
            if (m_top_idx[i] < 0 || (size_t)m_top_idx[i] >= classNames.size() || m_top_idx[i] >= classCount) fast = false;
        if (fast) {
            for (size_t i = 0; i < croppedFaces.size(); ++i) {
                names.push_back(classNames[(size_t)m_top_idx[i]]);
                sims.push_back(m_top_sim[i]);
            }
            return std::make_tuple(names, sims);
        }
        if (output_sims == m_out && !m_materialize) {  // nothing to scan: the reference's answer for a row without a maximum is element 0
            for (size_t i = 0; i < croppedFaces.size(); ++i) {
                const int k = i < m_top_idx.size() ? m_top_idx[i] : -1;
                const bool ok = k >= 0 && (size_t)k < classNames.size();
                names.push_back(classNames[ok ? (size_t)k : 0]);
                sims.push_back(i < m_top_sim.size() ? m_top_sim[i] : 0.f);
            }
            return std::make_tuple(names, sims);
        }
        for (size_t i = 0; i < croppedFaces.size(); ++i) {
            const float *row = output_sims + i * (size_t)classCount;
            const int argmax = (int)std::distance(row, std::max_element(row, row + classCount));
            names.push_back(classNames[(size_t)argmax]);
            sims.push_back(row[argmax]);
        }
        return std::make_tuple(names, sims);
    }
    // Extension: featureMatching + getOutputs fused on the device (no F x N matrix, no host scan).
    std::tuple<std::vector<std::string>, std::vector<float>> matchTop1() {
        State &s = st();
        if (classNames.empty() || s.croppedFaces.empty()) throw "Feature matching: No faces in database or no faces found";
        const int n = (int)s.croppedFaces.size();
        std::vector<int> idx((size_t)n);
        std::vector<float> sims((size_t)n);
        if (s.from_record && s.top_valid && (int)s.top_idx.size() == n) {  // coalesced request: computed by the batch's match stage
            idx = s.top_idx;
            sims = s.top_sim;
        } else {
            matmul.top1(s.embeds.data(), n, idx.data(), sims.data());
        }
        std::vector<std::string> names;
        for (int i = 0; i < n; ++i) {
            const int k = idx[(size_t)i];  // -1: no row won (NaN similarities); the reference's max_element answers element 0
            names.push_back(classNames[(k >= 0 && (size_t)k < classNames.size()) ? (size_t)k : 0]);
        }
        return std::make_tuple(names, sims);
    }
    // src/arcface.cpp:219-231: drawing only, not on the hot path (not even called by app.cpp); real OpenCV required.
    void visualize(cv::Mat &image, std::vector<std::string> names, std::vector<float> sims) {
#ifdef FRT_HAVE_OPENCV
        for (size_t i = 0; i < croppedFaces.size(); ++i) {
            const CroppedFace &c = croppedFaces[i];
            const float fontScaler = static_cast<float>(c.x2 - c.x1) / static_cast<float>(m_frameWidth);
            const cv::Scalar color = sims[i] >= m_knownPersonThresh ? cv::Scalar(0, 255, 0) : cv::Scalar(0, 0, 255);
            cv::rectangle(image, cv::Point(c.y1, c.x1), cv::Point(c.y2, c.x2), color, 2, 8, 0);
            cv::putText(image, names[i] + " " + std::to_string(sims[i]), cv::Point(c.y1 + 2, c.x2 - 3), cv::FONT_HERSHEY_DUPLEX, 0.1 + 2 * fontScaler, color, 1);
        }
#else
        (void)image;
        (void)names;
        (void)sims;
#endif
    }
    const float *embeddings() const { return st().embeds.data(); }
    frt_embedder *handle() { return h_; }
    MatMul &matcher() { return matmul; }
    bool coalescing() const {
        std::shared_ptr<frtdetail::CoalesceLink> l = link();
        return l && l->alive();
    }

  private:
    // What a call leaves behind for the next call of the same request - per calling THREAD (include/frt/coalesce.h): the reference keeps
    // it in the object and shares the object between its server's threads.
    struct State {
        std::vector<struct CroppedFace> croppedFaces;
        std::vector<float> embeds;
        float *out = nullptr;  // page-locked [F x N] similarity matrix (reused; the reference leaks new float[F*N] per call)
        size_t out_cap = 0;
        bool top_valid = false, from_record = false;
        std::vector<int> top_idx;
        std::vector<float> top_sim;
        ~State() { frt_pinned_free(out); }
    };
    State &st() const { return frtdetail::perThread<State>(m_id); }
    static std::vector<struct CroppedFace> &croppedFacesOf(const ArcFaceIR50 *o) { return o->st().croppedFaces; }

  public:
    // `std::vector<struct CroppedFace> croppedFaces` of the reference (src/arcface.h:37), one per calling thread
    frtdetail::PerThreadVector<struct CroppedFace, ArcFaceIR50, &ArcFaceIR50::croppedFacesOf> croppedFaces;
    // static int classCount: inherited from ArcFaceIR50Statics<> above (src/arcface.h:39, arcface.cpp:19)

  private:
    // forward() of a frame this thread's coalesced findFace() has just analysed: same frame bytes (pointer, size, sampled fingerprint),
    // same boxes, every ROI non-empty -> embeddings, crops and first maxima come from the record
    std::shared_ptr<frtdetail::CoalesceLink> link() const { return std::atomic_load(&m_link); }
    // FRT_COALESCE: the detector of an auto-linked pair was destroyed (its destructor shut the coalescer down) - wait for a new partner
    void relinkIfOrphaned() {
        if (!m_auto) return;
        std::shared_ptr<frtdetail::CoalesceLink> l = link();
        if (!l || l->alive()) return;
        std::lock_guard<std::mutex> lk(m_relink);
        l = link();
        if (!l || l->alive()) return;
        std::atomic_store(&m_link, std::shared_ptr<frtdetail::CoalesceLink>());
        try {
            frtdetail::autoLink(false, frtdetail::Pending{m_device, m_frameWidth, m_frameHeight, m_devBatch, nullptr, h_, matmul.handle(), &m_link});
        } catch (...) {  // (a partner exists but the coalescer could not be built: stay on the plain path)
        }
    }
    bool forwardFromRecord(State &s, const cv::Mat &image, const std::vector<struct Bbox> &boxes) {
        relinkIfOrphaned();
        std::shared_ptr<frtdetail::CoalesceLink> lnk = link();
        if (!lnk) return false;
        frtdetail::FrameRecord &fr = frtdetail::frameRecord();
        const size_t n = boxes.size();
        if (fr.link != lnk.get() || fr.data != image.data || fr.rows != image.rows || fr.cols != image.cols || fr.boxes.size() != n ||
            m_INPUT_H != 112 || m_INPUT_W != 112 || m_OUTPUT_D != 512)
            return false;
        for (size_t i = 0; i < n; ++i)
            if (std::memcmp(&fr.boxes[i], &boxes[i], sizeof(Bbox)) != 0 || !fr.res[i].valid) return false;
        if (fr.print != frtdetail::framePrint(image.data, image.rows, image.cols, (size_t)image.step)) return false;
        s.embeds.assign(fr.embeds.begin(), fr.embeds.begin() + (long)(n * 512));
        bool matched = true;
        s.top_idx.resize(n);
        s.top_sim.resize(n);
        for (size_t i = 0; i < n; ++i) {
            CroppedFace c;
            c.face = cv::Mat(112, 112, CV_8UC3, &fr.crops[i * 112 * 112 * 3]).clone();
            // faceMat = the recogniser's input tensor as preprocessFace() leaves it (src/arcface.cpp:105-114): planar RGB, (x - 127.5) / 128 -
            // both steps are exact in binary floating point, so this host loop IS the device kernel's output, bit for bit
            c.faceMat = cv::Mat(3 * 112, 112, CV_32FC1);
            float *t = c.faceMat.ptr<float>(0);
            const unsigned char *px = c.face.data;
            for (int k = 0; k < 112 * 112; ++k)
                for (int ch = 0; ch < 3; ++ch) t[ch * 112 * 112 + k] = ((float)px[k * 3 + (2 - ch)] - 127.5f) * 0.0078125f;
            c.x1 = boxes[i].x1;
            c.y1 = boxes[i].y1;
            c.x2 = boxes[i].x2;
            c.y2 = boxes[i].y2;
            s.croppedFaces.push_back(c);
            s.top_idx[i] = fr.res[i].match_idx;
            s.top_sim[i] = fr.res[i].match_sim;
            if (fr.res[i].match_idx < 0) matched = false;  // the batch ran without a gallery
        }
        // the batch's (row index, similarity) pairs name rows of the gallery it ran against: after addEmbedding / initKnownEmbeds / initMatMul
        // they would index a different classNames - featureMatching / getOutputs then take the device path with these embeddings
        if (!fr.galleryGen || fr.galleryGen != frt_matcher_generation(matmul.handle())) matched = false;
        s.from_record = true;
        s.top_valid = matched;
        return true;
    }
    frt_embedder *h_;
    uint64_t m_id;
    int m_frameWidth, m_frameHeight, m_INPUT_C, m_INPUT_H, m_INPUT_W, m_OUTPUT_D, m_maxBatchSize, m_maxFacesPerScene;
    float m_knownPersonThresh;
    bool m_materialize = true;
    bool m_auto = false;
    int m_device = 0, m_devBatch = 0;
    std::mutex m_relink;
    std::vector<std::string> classNames;
    MatMul matmul;
    std::shared_ptr<frtdetail::CoalesceLink> m_link;
};

#endif  // FRT_ARCFACE_H
