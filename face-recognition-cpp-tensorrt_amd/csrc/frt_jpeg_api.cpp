// C ABI of the JPEG frame-ingest / reply steps (include/frt.h, section "Frame ingest" and "Reply step"): host threads do the
// entropy coding, the device does the transforms (kernels_jpeg.hip).  No CPU pixel path: without a HIP device every entry point
// except frt_jpeg_info / frt_base64_encode fails with FRT_ERR_DEVICE.
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "frt_host.hpp"
#include "frt_jpeg.hpp"
#include "frt_jpeg_dev.h"
#include "frt_kernels.h"

using frthost::guarded;
using frthost::raise;
using frthost::use_device;

namespace {

// fixed pool of worker threads; parallel_for blocks the caller until every index has run
class Pool {
  public:
    explicit Pool(int n) {
        for (int i = 0; i < n; ++i) th_.emplace_back([this] { loop(); });
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (std::thread &t : th_) t.join();
    }
    void parallel_for(int n, const std::function<void(int)> &fn) {
        if (n <= 0) return;
        if (th_.empty() || n == 1) {
            for (int i = 0; i < n; ++i) fn(i);
            return;
        }
        std::unique_lock<std::mutex> lk(mu_);
        fn_ = &fn;
        next_ = 0;
        total_ = n;
        done_ = 0;
        ++epoch_;
        cv_.notify_all();
        done_cv_.wait(lk, [&] { return done_ == total_; });
        fn_ = nullptr;
        if (err_) {  // a task threw (e.g. bad_alloc while a JFIF stream grows): rethrow here, where guarded() turns it into a status
            std::exception_ptr e = err_;
            err_ = nullptr;
            std::rethrow_exception(e);
        }
    }

  private:
    void loop() {
        unsigned seen = 0;
        std::unique_lock<std::mutex> lk(mu_);
        while (true) {
            cv_.wait(lk, [&] { return stop_ || (epoch_ != seen && fn_); });
            if (stop_) return;
            seen = epoch_;
            while (fn_ && next_ < total_) {
                const int i = next_++;
                const std::function<void(int)> *f = fn_;
                lk.unlock();
                std::exception_ptr ep;
                try {
                    (*f)(i);
                } catch (...) {  // never let an exception leave a worker thread (std::terminate)
                    ep = std::current_exception();
                }
                lk.lock();
                if (ep && !err_) err_ = ep;
                if (++done_ == total_) done_cv_.notify_all();
            }
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    const std::function<void(int)> *fn_ = nullptr;
    int next_ = 0, total_ = 0, done_ = 0;
    unsigned epoch_ = 0;
    bool stop_ = false;
    std::exception_ptr err_;
};

}  // namespace

struct frt_jpeg_decoder {
    int device = 0, max_images = 0, max_w = 0, max_h = 0;
    std::mutex mu;
    std::unique_ptr<Pool> pool;
    hipStream_t own_stream = nullptr;
    // two staging sets so that the host can entropy-decode batch b+1 while batch b is still being copied / transformed
    static constexpr int NSET = 2;
    struct Set {
        int16_t *h_coef = nullptr, *d_coef = nullptr;
        JpegImageDesc *h_desc = nullptr, *d_desc = nullptr;
        uint8_t *d_planes = nullptr, *d_full = nullptr;  // component planes; decoded full-size frames (only when a resize follows)
        hipEvent_t ev = nullptr;
        bool pending = false;
    } set[NSET];
    int cur = 0;
    size_t blocks_per_image = 0, plane_bytes_per_image = 0;
    // encoder scratch (reply step)
    uint16_t *d_q = nullptr;
    int q_quality = -1;
    // A private stream only when a caller really passes NULL: an extra non-blocking stream created before a pipeline's stage streams
    // disturbs ROCm's stream -> hardware-queue mapping (see frt_matcher::load_begin); callers that feed a pipeline pass their producer stream.
    hipStream_t ensure_stream() {
        if (!own_stream) HIPCHK(hipStreamCreate(&own_stream));
        return own_stream;
    }
    frtjpeg::EncTables enc{};
    uint8_t *d_crops = nullptr;
    int16_t *d_eblocks = nullptr, *h_eblocks = nullptr;
    size_t ecap_pixels = 0, ecap_blocks = 0;
};

namespace {

size_t blocks_for(int w, int h) {  // upper bound over the supported samplings: whole 16x16 MCUs, 4:4:4 density
    const size_t mx = (size_t)(w + 15) / 16, my = (size_t)(h + 15) / 16;
    return mx * my * 4 * 3;
}

void fill_desc(const frtjpeg::Header &h, JpegImageDesc &d, uint64_t coef_block0, uint64_t plane0, uint64_t out_off) {
    d.width = h.width;
    d.height = h.height;
    d.ncomp = h.ncomp;
    d.hmax = h.hmax;
    d.vmax = h.vmax;
    uint64_t po = plane0;
    for (int i = 0; i < 3; ++i) {
        const frtjpeg::Component &c = h.c[i < h.ncomp ? i : 0];
        d.h[i] = c.h;
        d.v[i] = c.v;
        d.bw[i] = c.bw;
        d.bh[i] = c.bh;
        d.dw[i] = c.dw;
        d.dh[i] = c.dh;
        d.pitch[i] = c.bw * 8;
        d.block0[i] = (uint32_t)c.block0;
        d.plane_off[i] = po;
        if (i < h.ncomp) {
            po += (uint64_t)c.bw * 8 * c.bh * 8;
            std::memcpy(d.q[i], h.q[c.tq], sizeof(d.q[i]));
        }
    }
    d.coef_block0 = coef_block0;
    d.out_off = out_off;
}

void check_supported(const frtjpeg::Header &h) {
    const bool ok = h.ncomp == 1 || (h.hmax == 1 && h.vmax == 1) || (h.hmax == 2 && h.vmax == 1) || (h.hmax == 2 && h.vmax == 2);
    if (!ok) raise(FRT_ERR_FORMAT, "jpeg: unsupported chroma subsampling (4:4:4, 4:2:2 and 4:2:0 are supported)");
}

}  // namespace

extern "C" {

int frt_jpeg_info(const uint8_t *data, size_t size, int *width, int *height, int *components) {
    return guarded([&] {
        if (!data) raise(FRT_ERR_INVALID, "null argument");
        frtjpeg::Parsed p;
        std::string err;
        if (frtjpeg::parse(data, size, p, err)) raise(FRT_ERR_FORMAT, err);
        if (width) *width = p.h.width;
        if (height) *height = p.h.height;
        if (components) *components = p.h.ncomp;
    });
}

int frt_jpeg_read_coefficients(const uint8_t *data, size_t size, int16_t *coef_out, size_t coef_capacity_blocks, int32_t *geometry_out) {
    return guarded([&] {
        if (!data || !geometry_out) raise(FRT_ERR_INVALID, "null argument");
        frtjpeg::Parsed p;
        std::string err;
        if (frtjpeg::parse(data, size, p, err)) raise(FRT_ERR_FORMAT, err);
        const frtjpeg::Header &h = p.h;
        int32_t *g = geometry_out;  // width, height, ncomp, hmax, vmax, then per component: h, v, bw, bh, dw, dh, block0
        g[0] = h.width; g[1] = h.height; g[2] = h.ncomp; g[3] = h.hmax; g[4] = h.vmax;
        for (int i = 0; i < 3; ++i) {
            const frtjpeg::Component &c = h.c[i];
            int32_t *q = g + 5 + 7 * i;
            q[0] = c.h; q[1] = c.v; q[2] = c.bw; q[3] = c.bh; q[4] = c.dw; q[5] = c.dh; q[6] = (int32_t)c.block0;
        }
        g[26] = (int32_t)h.total_blocks;
        for (int i = 0; i < 3; ++i)
            for (int k = 0; k < 64; ++k) g[27 + 64 * i + k] = h.q[h.c[i < h.ncomp ? i : 0].tq][k];
        if (!coef_out) return;
        if (coef_capacity_blocks < h.total_blocks) raise(FRT_ERR_CAPACITY, "jpeg: coefficient buffer too small");
        std::memset(coef_out, 0, h.total_blocks * 128);
        if (frtjpeg::decode_coefficients(data, size, p, coef_out, err)) raise(FRT_ERR_FORMAT, err);
    });
}

int frt_jpeg_decoder_create(int max_images, int max_width, int max_height, int n_threads, int device, frt_jpeg_decoder **out) {
    return guarded([&] {
        if (!out) raise(FRT_ERR_INVALID, "null argument");
        *out = nullptr;
        if (max_images < 1 || max_width < 1 || max_height < 1 || max_width > 65535 || max_height > 65535) raise(FRT_ERR_INVALID, "jpeg decoder: bad limits");
        // staging is sized for max_images x max_width x max_height (about 15 bytes per pixel over both sets, 6 of them pinned host
        // memory): 2^31 pixels per batch = 256 8K frames is far beyond any frame ingest and keeps a typo from pinning tens of GB
        if ((uint64_t)max_images * (uint64_t)max_width * (uint64_t)max_height > (1ull << 31))
            raise(FRT_ERR_CAPACITY, "jpeg decoder: max_images * max_width * max_height exceeds 2^31 pixels per batch");
        use_device(device);
        std::unique_ptr<frt_jpeg_decoder> d(new frt_jpeg_decoder);
        d->device = device;
        d->max_images = max_images;
        d->max_w = max_width;
        d->max_h = max_height;
        if (n_threads <= 0) {
            n_threads = (int)std::thread::hardware_concurrency();
            if (n_threads > max_images) n_threads = max_images;
            if (n_threads > 64) n_threads = 64;
        }
        d->pool.reset(new Pool(n_threads > 1 ? n_threads : 0));
        d->blocks_per_image = blocks_for(max_width, max_height);
        d->plane_bytes_per_image = d->blocks_per_image * 64;
        // (own_stream is created on first use with a NULL caller stream: see ensure_stream)
        for (frt_jpeg_decoder::Set &s : d->set) {
            const size_t cb = d->blocks_per_image * max_images * 128;
            HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&s.h_coef), cb, hipHostMallocDefault));
            HIPCHK(hipMalloc(reinterpret_cast<void **>(&s.d_coef), cb));
            HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&s.h_desc), sizeof(JpegImageDesc) * max_images, hipHostMallocDefault));
            HIPCHK(hipMalloc(reinterpret_cast<void **>(&s.d_desc), sizeof(JpegImageDesc) * max_images));
            HIPCHK(hipMalloc(reinterpret_cast<void **>(&s.d_planes), d->plane_bytes_per_image * max_images));
            HIPCHK(hipMalloc(reinterpret_cast<void **>(&s.d_full), (size_t)max_width * max_height * 3 * max_images));
            HIPCHK(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
        }
        *out = d.release();
    });
}

void frt_jpeg_decoder_destroy(frt_jpeg_decoder *d) {
    if (!d) return;
    (void)hipSetDevice(d->device);
    (void)hipDeviceSynchronize();
    for (frt_jpeg_decoder::Set &s : d->set) {
        if (s.h_coef) (void)hipHostFree(s.h_coef);
        if (s.h_desc) (void)hipHostFree(s.h_desc);
        for (void *p : {(void *)s.d_coef, (void *)s.d_desc, (void *)s.d_planes, (void *)s.d_full})
            if (p) (void)hipFree(p);
        if (s.ev) (void)hipEventDestroy(s.ev);
    }
    for (void *p : {(void *)d->d_q, (void *)d->d_crops, (void *)d->d_eblocks})
        if (p) (void)hipFree(p);
    if (d->h_eblocks) (void)hipHostFree(d->h_eblocks);
    if (d->own_stream) (void)hipStreamDestroy(d->own_stream);
    delete d;
}

// n JPEG byte strings -> n u8 BGR frames [out_h][out_w][3] on the device (frames_dev), asynchronous on hip_stream.
int frt_jpeg_decode_batch_dev(frt_jpeg_decoder *d, const uint8_t *const *data, const size_t *sizes, int n, void *frames_dev, int out_h, int out_w,
                              void *hip_stream) {
    return guarded([&] {
        if (!d || !data || !sizes || !frames_dev || n < 0 || out_h < 1 || out_w < 1) raise(FRT_ERR_INVALID, "jpeg decode: bad argument");
        if (n == 0) return;
        if (n > d->max_images) raise(FRT_ERR_CAPACITY, "jpeg decode: more images than the decoder was created for");
        std::lock_guard<std::mutex> lk(d->mu);
        use_device(d->device);
        hipStream_t st = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : d->ensure_stream();
        frt_jpeg_decoder::Set &s = d->set[d->cur];
        d->cur = (d->cur + 1) % frt_jpeg_decoder::NSET;
        if (s.pending) {  // the staging set's previous batch must have left the host buffers
            HIPCHK(hipEventSynchronize(s.ev));
            s.pending = false;
        }
        // ---- host: headers (serial, cheap), then one entropy-decoding task per image
        std::vector<frtjpeg::Parsed> parsed((size_t)n);
        std::string err;
        bool any_resize = false;
        uint64_t blk = 0, pl = 0;
        int max_bpc = 0, max_w = 0, max_h = 0;
        for (int i = 0; i < n; ++i) {
            if (!data[i]) raise(FRT_ERR_INVALID, "jpeg decode: null image");
            if (frtjpeg::parse(data[i], sizes[i], parsed[(size_t)i], err)) raise(FRT_ERR_FORMAT, err);
            const frtjpeg::Header &h = parsed[(size_t)i].h;
            check_supported(h);
            if (h.width > d->max_w || h.height > d->max_h || h.total_blocks > d->blocks_per_image)
                raise(FRT_ERR_CAPACITY, "jpeg decode: image larger than the decoder's max_width x max_height");
            const bool resize = h.width != out_w || h.height != out_h;
            any_resize = any_resize || resize;
            // images that already have the output size are written straight into frames_dev
            const uint64_t out_off = resize ? (uint64_t)i * d->max_w * d->max_h * 3 : (uint64_t)i * out_w * out_h * 3;
            fill_desc(h, s.h_desc[i], blk, pl, out_off);
            blk += h.total_blocks;
            pl += h.total_blocks * 64;
            for (int c = 0; c < h.ncomp; ++c) max_bpc = std::max(max_bpc, h.c[c].bw * h.c[c].bh);
            max_w = std::max(max_w, h.width);
            max_h = std::max(max_h, h.height);
        }
        std::vector<std::string> errs((size_t)n);
        std::atomic<int> bad{0};
        d->pool->parallel_for(n, [&](int i) {
            const frtjpeg::Header &h = parsed[(size_t)i].h;
            int16_t *c = s.h_coef + s.h_desc[i].coef_block0 * 64;
            std::memset(c, 0, h.total_blocks * 128);
            if (frtjpeg::decode_coefficients(data[i], sizes[i], parsed[(size_t)i], c, errs[(size_t)i])) bad.fetch_add(1);
        });
        if (bad.load())
            for (const std::string &e : errs)
                if (!e.empty()) raise(FRT_ERR_FORMAT, e);
        // ---- device: coefficients + descriptors up, transforms, optional resize
        HIPCHK(hipMemcpyAsync(s.d_coef, s.h_coef, blk * 128, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(s.d_desc, s.h_desc, sizeof(JpegImageDesc) * n, hipMemcpyHostToDevice, st));
        uint8_t *frames = reinterpret_cast<uint8_t *>(frames_dev);
        if (!any_resize) {
            launch_jpeg_decode(s.d_coef, s.d_desc, n, max_bpc, max_w, max_h, s.d_planes, frames, st);
        } else {
            // mixed batch: two destination buffers, so the descriptors of same-size images must point into frames_dev as well; simplest
            // correct form: decode everything into d_full, then resize / copy every image into its slot
            for (int i = 0; i < n; ++i) s.h_desc[i].out_off = (uint64_t)i * d->max_w * d->max_h * 3;
            HIPCHK(hipMemcpyAsync(s.d_desc, s.h_desc, sizeof(JpegImageDesc) * n, hipMemcpyHostToDevice, st));
            launch_jpeg_decode(s.d_coef, s.d_desc, n, max_bpc, max_w, max_h, s.d_planes, s.d_full, st);
            for (int i = 0; i < n; ++i) {
                const frtjpeg::Header &h = parsed[(size_t)i].h;
                const uint8_t *src = s.d_full + (size_t)i * d->max_w * d->max_h * 3;
                uint8_t *dst = frames + (size_t)i * out_w * out_h * 3;
                if (h.width == out_w && h.height == out_h)
                    HIPCHK(hipMemcpyAsync(dst, src, (size_t)out_w * out_h * 3, hipMemcpyDeviceToDevice, st));
                else  // cv::resize(rawInput, frame, Size(frameW, frameH)) - src/app.cpp:301, default INTER_LINEAR
                    launch_resize_linear(src, 1, h.height, h.width, (size_t)h.width * 3, 0, dst, out_h, out_w, (size_t)out_w * 3, 0, st);
            }
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(s.ev, st));
        s.pending = true;
    });
}

// one image, host in / host out (tests, tools): decoded at its own size
int frt_jpeg_decode(frt_jpeg_decoder *d, const uint8_t *data, size_t size, uint8_t *bgr_out, size_t capacity, int *width, int *height) {
    return guarded([&] {
        if (!d || !data || !bgr_out) raise(FRT_ERR_INVALID, "null argument");
        int w = 0, h = 0, c = 0;
        if (frt_jpeg_info(data, size, &w, &h, &c) != FRT_OK) raise(FRT_ERR_FORMAT, frthost::last_error());
        if ((size_t)w * h * 3 > capacity) raise(FRT_ERR_CAPACITY, "jpeg decode: output buffer too small");
        use_device(d->device);
        uint8_t *dev = nullptr;
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&dev), (size_t)w * h * 3));
        const uint8_t *one[1] = {data};
        const size_t sz[1] = {size};
        const int rc = frt_jpeg_decode_batch_dev(d, one, sz, 1, dev, h, w, nullptr);
        std::string msg = frthost::last_error();
        hipError_t e = hipSuccess;
        if (rc == FRT_OK) {
            e = hipStreamSynchronize(d->ensure_stream());
            if (e == hipSuccess) e = hipMemcpy(bgr_out, dev, (size_t)w * h * 3, hipMemcpyDeviceToHost);
        }
        (void)hipFree(dev);
        if (rc != FRT_OK) raise(rc, msg);
        HIPCHK(e);
        if (width) *width = w;
        if (height) *height = h;
    });
}

// ---------------------------------------------------------------------------------------------------------------- reply step
static void encode_setup(frt_jpeg_decoder *d, int quality, int n, int rows, int cols) {
    if (d->q_quality != quality) {
        frtjpeg::make_enc_tables(quality, d->enc);
        if (!d->d_q) HIPCHK(hipMalloc(reinterpret_cast<void **>(&d->d_q), sizeof(uint16_t) * 128));
        HIPCHK(hipMemcpy(d->d_q, d->enc.q, sizeof(uint16_t) * 128, hipMemcpyHostToDevice));
        d->q_quality = quality;
    }
    const size_t px = (size_t)n * rows * cols * 3, blocks = (size_t)n * 6 * ((cols + 15) / 16) * ((rows + 15) / 16);
    if (px > d->ecap_pixels) {
        if (d->d_crops) (void)hipFree(d->d_crops);
        d->d_crops = nullptr;
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&d->d_crops), px));
        d->ecap_pixels = px;
    }
    if (blocks > d->ecap_blocks) {
        if (d->d_eblocks) (void)hipFree(d->d_eblocks);
        if (d->h_eblocks) (void)hipHostFree(d->h_eblocks);
        d->d_eblocks = d->h_eblocks = nullptr;
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&d->d_eblocks), blocks * 128));
        HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&d->h_eblocks), blocks * 128, hipHostMallocDefault));
        d->ecap_blocks = blocks;
    }
}

// cv::imencode(".jpg", crop) (src/app.cpp:328; quality 95, 4:2:0, standard tables) for n equally sized u8 BGR images.
// bgr: host (device_input = 0) or device (1) pointer, tight [n][rows][cols][3].  out: n slots of out_stride bytes; out_sizes[n].
int frt_jpeg_encode_batch(frt_jpeg_decoder *d, const void *bgr, int device_input, int n, int rows, int cols, int quality, uint8_t *out, size_t out_stride,
                          size_t *out_sizes) {
    return frt_jpeg_encode_batch_after(d, bgr, device_input, n, rows, cols, quality, out, out_stride, out_sizes, nullptr);
}

// Same, ordered behind the producer of a DEVICE input: ready_event (hipEvent_t as void*, may be NULL) was recorded by the caller after the
// work that writes `bgr` (a pipeline stage, frt_embedder_forward on another stream ...); the codec's stream waits for it on the device.
int frt_jpeg_encode_batch_after(frt_jpeg_decoder *d, const void *bgr, int device_input, int n, int rows, int cols, int quality, uint8_t *out,
                                size_t out_stride, size_t *out_sizes, void *ready_event) {
    return guarded([&] {
        if (!d || !bgr || !out || !out_sizes || n < 0 || rows < 1 || cols < 1 || rows > 65535 || cols > 65535) raise(FRT_ERR_INVALID, "jpeg encode: bad argument");
        if (n == 0) return;
        std::lock_guard<std::mutex> lk(d->mu);
        use_device(d->device);
        encode_setup(d, quality, n, rows, cols);
        hipStream_t st = d->ensure_stream();
        if (ready_event) HIPCHK(hipStreamWaitEvent(st, reinterpret_cast<hipEvent_t>(ready_event), 0));
        const uint8_t *src = reinterpret_cast<const uint8_t *>(bgr);
        if (!device_input) {
            HIPCHK(hipMemcpyAsync(d->d_crops, bgr, (size_t)n * rows * cols * 3, hipMemcpyHostToDevice, st));
            src = d->d_crops;
        }
        launch_jpeg_encode_blocks(src, n, rows, cols, d->d_q, d->d_eblocks, st);
        HIPCHK(hipGetLastError());
        const size_t per_img = (size_t)6 * ((cols + 15) / 16) * ((rows + 15) / 16);
        HIPCHK(hipMemcpyAsync(d->h_eblocks, d->d_eblocks, per_img * n * 128, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        std::vector<std::vector<uint8_t>> bytes((size_t)n);
        d->pool->parallel_for(n, [&](int i) { frtjpeg::write_jfif_420(d->enc, cols, rows, d->h_eblocks + (size_t)i * per_img * 64, bytes[(size_t)i]); });
        for (int i = 0; i < n; ++i) {
            if (bytes[(size_t)i].size() > out_stride) raise(FRT_ERR_CAPACITY, "jpeg encode: output slot too small");
            std::memcpy(out + (size_t)i * out_stride, bytes[(size_t)i].data(), bytes[(size_t)i].size());
            out_sizes[i] = bytes[(size_t)i].size();
        }
    });
}

// host half of the encoder alone: quantised zigzag blocks (layout of frt_jpeg.hpp:write_jfif_420) -> JFIF byte stream
int frt_jpeg_write_jfif(int quality, int width, int height, const int16_t *coef_zigzag, uint8_t *out, size_t capacity, size_t *size_out) {
    return guarded([&] {
        if (!coef_zigzag || !out || !size_out || width < 1 || height < 1 || width > 65535 || height > 65535) raise(FRT_ERR_INVALID, "jpeg write: bad argument");
        frtjpeg::EncTables t;
        frtjpeg::make_enc_tables(quality, t);
        std::vector<uint8_t> b;
        frtjpeg::write_jfif_420(t, width, height, coef_zigzag, b);
        if (b.size() > capacity) raise(FRT_ERR_CAPACITY, "jpeg write: output buffer too small");
        std::memcpy(out, b.data(), b.size());
        *size_out = b.size();
    });
}

size_t frt_base64_encode(const uint8_t *data, size_t size, char *out, size_t capacity) {
    const size_t need = (size + 2) / 3 * 4;
    if (!out || capacity < need + 1 || (!data && size)) return need + 1;
    const std::string s = frtjpeg::base64(data, size);
    std::memcpy(out, s.data(), s.size());
    out[s.size()] = 0;
    return s.size();
}

}  // extern "C"
