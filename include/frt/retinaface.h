// Drop-in for /root/reference/src/retinaface.h: class RetinaFace with the reference's constructor and findFace signatures
// (src/retinaface.h:18-23).  engineFile is an FRTW weight blob instead of a TensorRT engine; inputName / outputNames are
// accepted and ignored (the reference only uses them for getBindingIndex, src/retinaface.cpp:93-95).
#ifndef FRT_RETINAFACE_H
#define FRT_RETINAFACE_H

#include <algorithm>
#include <array>
#include <cassert>
#include <cstring>

#include "coalesce.h"
#include "common.h"
#include "cvlite.h"

#ifndef CLIP
#define CLIP(a, min, max) (MAX(MIN(a, max), min))
#endif

struct anchorBox {
    float cx, cy, sx, sy;
};

class RetinaFace {
  public:
    RetinaFace(TRTLogger gLogger, const std::string engineFile, int frameWidth, int frameHeight, std::string inputName,
               std::vector<std::string> outputNames, std::vector<int> inputShape, int maxBatchSize, int maxFacesPerScene,
               float nms_threshold, float bbox_threshold, int device = 0)
        : h_(nullptr), m_maxFacesPerScene(maxFacesPerScene), m_maxBatchSize(maxBatchSize), m_frameWidth(frameWidth), m_frameHeight(frameHeight) {
        (void)gLogger;
        (void)inputName;
        assert(inputShape.size() == 3);   // src/retinaface.cpp:8
        assert(outputNames.size() == 2);  // src/retinaface.cpp:86
        const int envFrames = frtdetail::coalesceEnv().frames;  // FRT_COALESCE=<frames>: room for a coalesced batch (include/frt/coalesce.h)
        const int devBatch = std::max(maxBatchSize, envFrames);
        checkFrtStatus(frt_detector_create(engineFile.c_str(), frameWidth, frameHeight, inputShape[0], inputShape[1], inputShape[2],
                                           devBatch, maxFacesPerScene, nms_threshold, bbox_threshold, device, &h_));
        std::cout << "[INFO] Loading RetinaFace Engine...\n";
        m_device = device;
        m_devBatch = devBatch;
        if (envFrames > 0) {
            m_auto = true;
            try {
                frtdetail::autoLink(true, frtdetail::Pending{device, frameWidth, frameHeight, devBatch, h_, nullptr, nullptr, &m_link});
            } catch (...) {
                frt_detector_destroy(h_);
                throw;
            }
        }
    }
    ~RetinaFace() {
        frtdetail::autoUnlink(true, &m_link);
        if (std::shared_ptr<frtdetail::CoalesceLink> l = link()) l->shutdown();  // the coalescer borrows this object's detector (waits for calls inside it)
        frt_detector_destroy(h_);
    }
    // set by ArcFaceIR50::coalesceWith(detector)
    void attachCoalescer(const std::shared_ptr<frtdetail::CoalesceLink> &l) {
        std::shared_ptr<frtdetail::CoalesceLink> old = link();
        if (old && old != l) old->shutdown();
        std::atomic_store(&m_link, l);
    }
    bool coalescing() const {
        std::shared_ptr<frtdetail::CoalesceLink> l = link();
        return l && l->alive();
    }
    // batches submitted / frames carried by the coalescer so far (frames / batches = the batch size the load produced)
    bool coalesceStats(long &batches, long &frames) const {
        batches = frames = 0;
        std::shared_ptr<frtdetail::CoalesceLink> l = link();
        frtdetail::CoalesceUse use(l.get());
        return use.c && frt_coalescer_stats(use.c, &batches, &frames) == FRT_OK;
    }
    RetinaFace(const RetinaFace &) = delete;
    RetinaFace &operator=(const RetinaFace &) = delete;

    // src/retinaface.cpp:147-152.  img must be CV_8UC3 of frameWidth x frameHeight (the caller resizes, src/app.cpp:301).
    std::vector<struct Bbox> findFace(cv::Mat &img) {
        relinkIfOrphaned();
        std::shared_ptr<frtdetail::CoalesceLink> lnk = link();
        frtdetail::CoalesceUse use(img.rows == m_frameHeight && img.cols == m_frameWidth ? lnk.get() : nullptr);
        if (use.c) {
            // coalesced: this frame joins the batch that the requests being served right now form together; the batch runs detector, crop,
            // recogniser and top-1 in one device pass and this thread keeps its frame's share for forward() / featureMatching()
            frtdetail::FrameRecord &fr = frtdetail::frameRecord();
            const size_t K = (size_t)m_maxFacesPerScene;
            fr.link = nullptr;
            fr.res.resize(K);
            fr.embeds.resize(K * 512);
            fr.crops.resize(K * 112 * 112 * 3);
            int n = 0;
            const unsigned gen0 = lnk->mat ? frt_matcher_generation(lnk->mat) : 0u;
            checkFrtStatus(frt_coalescer_infer_crops(use.c, img.data, img.rows, img.cols, (size_t)img.step, fr.res.data(), fr.embeds.data(), fr.crops.data(), &n));
            fr.boxes.resize((size_t)n);
            for (int i = 0; i < n; ++i) std::memcpy(&fr.boxes[(size_t)i], &fr.res[(size_t)i].box, sizeof(Bbox));
            fr.data = img.data;
            fr.rows = img.rows;
            fr.cols = img.cols;
            fr.print = frtdetail::framePrint(img.data, img.rows, img.cols, (size_t)img.step);
            const unsigned gen1 = lnk->mat ? frt_matcher_generation(lnk->mat) : 0u;
            fr.galleryGen = gen0 == gen1 ? gen1 : 0u;  // a reload while the batch ran: its row indices may name rows of either gallery
            fr.link = lnk.get();
            return fr.boxes;
        }
        std::vector<struct Bbox> out((size_t)m_maxFacesPerScene);
        int n = 0;
        checkFrtStatus(frt_detector_find_faces(h_, img.data, img.rows, img.cols, (size_t)img.step, reinterpret_cast<frt_bbox *>(out.data()), &n));
        out.resize((size_t)n);
        return out;
    }
    // New surface: frames.size() <= maxBatchSize frames of identical size in one device pass.
    std::vector<std::vector<struct Bbox>> findFaceBatch(std::vector<cv::Mat> &frames) {
        const int nf = (int)frames.size();
        std::vector<std::vector<struct Bbox>> res((size_t)nf);
        if (!nf) return res;
        const int rows = frames[0].rows, cols = frames[0].cols;
        std::vector<unsigned char> packed((size_t)nf * rows * cols * 3);
        for (int f = 0; f < nf; ++f)
            for (int r = 0; r < rows; ++r)
                std::memcpy(&packed[((size_t)f * rows + r) * cols * 3], frames[f].data + (size_t)r * frames[f].step, (size_t)cols * 3);
        std::vector<frt_bbox> out((size_t)nf * m_maxFacesPerScene);
        std::vector<int> n((size_t)nf);
        checkFrtStatus(frt_detector_find_faces_batch(h_, packed.data(), nf, rows, cols, (size_t)cols * 3, (size_t)rows * cols * 3, out.data(), n.data()));
        for (int f = 0; f < nf; ++f) {
            const Bbox *b = reinterpret_cast<const Bbox *>(&out[(size_t)f * m_maxFacesPerScene]);
            res[(size_t)f].assign(b, b + n[(size_t)f]);
        }
        return res;
    }
    // Optional alignment mode (no reference counterpart; needs an engine file exported WITH LandmarkHead, see include/frt.h).
    // landmarks[i] = (x0,y0,...,x4,y4), x = column, y = row, frame pixels.
    bool hasLandmarks() const { return frt_detector_has_landmarks(h_) != 0; }
    std::vector<struct Bbox> findFaceLandmarks(cv::Mat &img, std::vector<std::array<float, 10>> &landmarks) {
        std::vector<struct Bbox> out((size_t)m_maxFacesPerScene);
        landmarks.assign((size_t)m_maxFacesPerScene, std::array<float, 10>());
        int n = 0;
        checkFrtStatus(frt_detector_find_faces_landmarks(h_, img.data, img.rows, img.cols, (size_t)img.step, reinterpret_cast<frt_bbox *>(out.data()),
                                                         landmarks.empty() ? nullptr : landmarks[0].data(), &n));
        out.resize((size_t)n);
        landmarks.resize((size_t)n);
        return out;
    }
    frt_detector *handle() { return h_; }

  private:
    // FRT_COALESCE: the partner of an auto-linked pair was destroyed (its destructor shut the coalescer down) - wait for a new partner
    std::shared_ptr<frtdetail::CoalesceLink> link() const { return std::atomic_load(&m_link); }
    void relinkIfOrphaned() {
        if (!m_auto) return;
        std::shared_ptr<frtdetail::CoalesceLink> l = link();
        if (!l || l->alive()) return;
        std::lock_guard<std::mutex> lk(m_relink);
        l = link();
        if (!l || l->alive()) return;
        std::atomic_store(&m_link, std::shared_ptr<frtdetail::CoalesceLink>());
        try {
            frtdetail::autoLink(true, frtdetail::Pending{m_device, m_frameWidth, m_frameHeight, m_devBatch, h_, nullptr, nullptr, &m_link});
        } catch (...) {  // (a partner exists but the coalescer could not be built: stay on the plain path)
        }
    }
    frt_detector *h_;
    bool m_auto = false;
    int m_device = 0, m_devBatch = 0;
    std::mutex m_relink;
    int m_maxFacesPerScene, m_maxBatchSize, m_frameWidth, m_frameHeight;
    std::shared_ptr<frtdetail::CoalesceLink> m_link;
};

#endif  // FRT_RETINAFACE_H
