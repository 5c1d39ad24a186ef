// dcvc - standalone DCVC-UF encoder / decoder for 8-bit YUV420 files on an MI355X.
//
// The codec without the research harness: what test_video.py:166-399 (run_one_point_with_stream)
// does around the plugin - read YUV420 frames, code them picture by picture into the reference's
// stream container, decode the container again, write the reconstruction, log bits and PSNR - as a
// native tool on top of the C ABI only (include/dcvc_amd_codec.h, _ops.h, _stream.h). No Python, no
// torch: weights come from a .dcvw file (python -m dcvc_amd.export_weights), pictures travel as u8
// planes and are converted on the device (frame_io.hip).
//
//   dcvc encode --intra I.dcvw [--inter P.dcvw] -i in.yuv -W 1920 -H 1080 [-n frames] --qp-i 32 [--qp-p 32]
//               [--intra-period -1] [--reset-interval 32] -o out.bin
//   dcvc decode --intra I.dcvw [--inter P.dcvw] -i out.bin [-o rec.yuv] [-n frames] [--ref in.yuv --json log.json]
//               (the container carries no picture count: the last chunk of an 8-picture model is padded by repeating
//               the final picture, test_video.py:104-110 - give -n, or --ref whose length then trims the output, as the
//               reference's maximum_read = min(g_frame_delay, frame_num - decoded) does)
//
// Picture-type decisions, reset rule, chunk padding, container, PSNR ((6 Y + U + V) / 8 on the
// 0..255 planes) and the JSON log (what compare_bd_rate.py / dcvc_amd/bd_rate.py read) follow
// test_video.py:204-233, 95-110, 240-257, 32-45 and src/utils/common.py:46-116.
#include "dcvc_amd_codec.h"
#include "dcvc_amd_ops.h"
#include "dcvc_amd_rans.h"
#include "stream/container.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace {

[[noreturn]] void die(const std::string& msg)
{
    fprintf(stderr, "dcvc: %s\n", msg.c_str());
    exit(2);
}

void hip_ok(hipError_t e, const char* what)
{
    if (e != hipSuccess) die(std::string(what) + ": " + hipGetErrorString(e));
}

void abi_ok(long long rc, const char* what)
{
    if (rc < 0) die(std::string(what) + ": " + dcvc_last_error());
}

// ------------------------------------------------------------------------------------ .dcvw
struct WeightFile {
    int kind = -1;                 // 0 dmci, 1 ld, 2 hts, 3 htl
    float skip_thres = 0.f;
    std::vector<char> blob;
    std::vector<std::string> names;
    std::vector<const char*> name_ptrs;
    std::vector<const void*> data;
    std::vector<int> dtypes, ndims;
    std::vector<int64_t> dims;
};

WeightFile load_weights(const std::string& path)
{
    WeightFile w;
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) die("cannot open " + path);
    const std::streamsize size = f.tellg();
    f.seekg(0);
    if (size < 20) die(path + " is not a .dcvw file");
    w.blob.resize(static_cast<size_t>(size));
    if (!f.read(w.blob.data(), size)) die("cannot read " + path);
    const char* p = w.blob.data();
    const char* end = p + size;
    if (std::memcmp(p, "DCVW1\0\0\0", 8) != 0) die(path + " is not a .dcvw file");
    p += 8;
    uint32_t kind, count;
    std::memcpy(&kind, p, 4); std::memcpy(&w.skip_thres, p + 4, 4); std::memcpy(&count, p + 8, 4);
    p += 12;
    w.kind = static_cast<int>(kind);
    // every field is checked against the end of the file before it is read: the file comes from the user
    auto need = [&](uint64_t bytes, const char* what) {
        if (static_cast<uint64_t>(end - p) < bytes) die(path + ": truncated (" + what + ")");
    };
    for (uint32_t i = 0; i < count; ++i) {
        const char* rec = p;
        need(2, "name length");
        uint16_t nl;
        std::memcpy(&nl, p, 2); p += 2;
        need(static_cast<uint64_t>(nl) + 2, "name");
        w.names.emplace_back(p, nl); p += nl;
        const int dtype = static_cast<uint8_t>(p[0]), nd = static_cast<uint8_t>(p[1]); p += 2;
        if (dtype > 2 || nd > 8) die(path + ": bad record " + w.names.back());
        w.dtypes.push_back(dtype); w.ndims.push_back(nd);
        need(8ull * nd + 8, "dimensions");
        uint64_t elems = 1;
        for (int d = 0; d < nd; ++d) {
            int64_t v;
            std::memcpy(&v, p, 8); p += 8;
            if (v < 0 || (v > 0 && elems > (1ull << 40) / static_cast<uint64_t>(v))) die(path + ": bad shape of " + w.names.back());
            elems *= static_cast<uint64_t>(v);
            w.dims.push_back(v);
        }
        uint64_t nbytes;
        std::memcpy(&nbytes, p, 8); p += 8;
        if (nbytes != elems * (dtype == 0 ? 2u : 4u)) die(path + ": size of " + w.names.back() + " does not match its shape");
        need(nbytes, "tensor data");
        w.data.push_back(p);
        p += nbytes;
        const size_t pad = (8 - ((p - rec) & 7)) & 7;
        if (static_cast<size_t>(end - p) < pad && i + 1 < count) die(path + ": truncated (padding)");
        p += std::min<size_t>(pad, static_cast<size_t>(end - p));
    }
    for (const std::string& n : w.names) w.name_ptrs.push_back(n.c_str());
    return w;
}

// ------------------------------------------------------------------------------------ codecs
struct Codecs {
    dcvc_dmci* intra = nullptr;
    dcvc_dmcld* ld = nullptr;
    dcvc_dmcht* ht = nullptr;
    int frames_per_p = 1;          // g_frame_delay: 1 (LD), 8 (HT-S / HT-L)
    bool has_inter() const { return ld != nullptr || ht != nullptr; }
};

Codecs make_codecs(const std::string& intra_path, const std::string& inter_path)
{
    Codecs c;
    {
        const WeightFile w = load_weights(intra_path);
        if (w.kind != 0) die(intra_path + " does not hold an intra model");
        c.intra = dcvc_dmci_create();
        if (c.intra == nullptr) die(std::string("cannot create the intra codec: ") + dcvc_last_error());
        abi_ok(dcvc_dmci_set_param(c.intra, static_cast<int>(w.names.size()), w.name_ptrs.data(), w.data.data(),
                                   w.dtypes.data(), w.ndims.data(), w.dims.data(), w.skip_thres), "intra set_param");
    }
    if (!inter_path.empty()) {
        const WeightFile w = load_weights(inter_path);
        const int n = static_cast<int>(w.names.size());
        if (w.kind == 1) {
            c.ld = dcvc_dmcld_create();
            if (c.ld == nullptr) die(std::string("cannot create the inter codec: ") + dcvc_last_error());
            abi_ok(dcvc_dmcld_set_param(c.ld, n, w.name_ptrs.data(), w.data.data(), w.dtypes.data(), w.ndims.data(),
                                        w.dims.data(), w.skip_thres), "inter set_param");
        } else if (w.kind == 2 || w.kind == 3) {
            c.ht = dcvc_dmcht_create(w.kind == 2);
            if (c.ht == nullptr) die(std::string("cannot create the inter codec: ") + dcvc_last_error());
            c.frames_per_p = 8;
            abi_ok(dcvc_dmcht_set_param(c.ht, n, w.name_ptrs.data(), w.data.data(), w.dtypes.data(), w.ndims.data(),
                                        w.dims.data(), w.skip_thres), "inter set_param");
        } else {
            die(inter_path + " does not hold an inter model");
        }
    }
    return c;
}

// ------------------------------------------------------------------------------------ pictures
struct Geometry {
    int H = 0, W = 0, Hp = 0, Wp = 0;      // picture, padded to multiples of 16
    size_t y_bytes() const { return static_cast<size_t>(H) * W; }
    size_t uv_bytes() const { return static_cast<size_t>(H / 2) * (W / 2) * 2; }
    size_t frame_bytes() const { return y_bytes() + uv_bytes(); }
};

Geometry geometry(int H, int W)
{
    if (H <= 0 || W <= 0 || (H & 1) || (W & 1)) die("picture size must be positive and even (YUV420)");
    Geometry g;
    g.H = H; g.W = W; g.Hp = (H + 15) / 16 * 16; g.Wp = (W + 15) / 16 * 16;
    return g;
}

constexpr int kMaxPictureSide = 16384;      // sanity cap for sizes read from a stream (8K is 7680 x 4320)

struct DeviceBuffers {
    uint8_t* yuv8 = nullptr;       // staging for one u8 picture (planes)
    void* x = nullptr;             // fp16 [H][W][3 * frames]
    void* x_hat = nullptr;         // fp16 [frames][Hp][Wp][3]
    void* y16 = nullptr;           // fp16 planes for PSNR
    uint8_t* out8 = nullptr;       // u8 planes of a reconstruction
    uint8_t* h_yuv = nullptr;      // pinned
    uint16_t* h_p16 = nullptr;     // pinned fp16 planes
    hipStream_t st = nullptr;
};

DeviceBuffers make_buffers(const Geometry& g, int frames)
{
    DeviceBuffers b;
    hip_ok(hipStreamCreateWithFlags(&b.st, hipStreamNonBlocking), "hipStreamCreate");
    hip_ok(hipMalloc(&b.yuv8, g.frame_bytes()), "hipMalloc");
    hip_ok(hipMalloc(&b.x, static_cast<size_t>(g.H) * g.W * 3 * frames * 2), "hipMalloc");
    hip_ok(hipMalloc(&b.x_hat, static_cast<size_t>(frames) * g.Hp * g.Wp * 3 * 2), "hipMalloc");
    hip_ok(hipMalloc(&b.y16, g.frame_bytes() * 2), "hipMalloc");
    hip_ok(hipMalloc(&b.out8, g.frame_bytes()), "hipMalloc");
    hip_ok(hipHostMalloc(reinterpret_cast<void**>(&b.h_yuv), g.frame_bytes(), hipHostMallocDefault), "hipHostMalloc");
    hip_ok(hipHostMalloc(reinterpret_cast<void**>(&b.h_p16), g.frame_bytes() * 2, hipHostMallocDefault), "hipHostMalloc");
    return b;
}

void free_buffers(DeviceBuffers& b)
{
    if (b.st) hip_ok(hipStreamSynchronize(b.st), "sync");
    for (void* d : {static_cast<void*>(b.yuv8), b.x, b.x_hat, b.y16, static_cast<void*>(b.out8)}) {
        if (d) hip_ok(hipFree(d), "hipFree");
    }
    if (b.h_yuv) hip_ok(hipHostFree(b.h_yuv), "hipHostFree");
    if (b.h_p16) hip_ok(hipHostFree(b.h_p16), "hipHostFree");
    if (b.st) hip_ok(hipStreamDestroy(b.st), "hipStreamDestroy");
    b = DeviceBuffers{};
}

double psnr_plane(const uint8_t* src, const uint16_t* rec16, size_t n)
{
    // metrics.py:10-24 on float64; rec16 holds fp16 values in 0..255
    double se = 0;
    for (size_t i = 0; i < n; ++i) {
        _Float16 h;
        std::memcpy(&h, rec16 + i, 2);
        const double d = static_cast<double>(src[i]) - static_cast<double>(static_cast<float>(h));
        se += d * d;
    }
    const double mse = se / static_cast<double>(n);
    if (std::isnan(mse) || std::isinf(mse)) return -999.9;
    const double p = mse > 1e-10 ? 10.0 * std::log10(255.0 * 255.0 / mse) : 999.9;
    return p < 99.9 ? p : 99.9;
}

struct Args {
    std::map<std::string, std::string> kv;
    bool has(const std::string& k) const { return kv.count(k) != 0; }
    std::string str(const std::string& k, const std::string& def = "") const { return has(k) ? kv.at(k) : def; }
    int num(const std::string& k, int def) const { return has(k) ? atoi(kv.at(k).c_str()) : def; }
};

Args parse(int argc, char** argv)
{
    Args a;
    for (int i = 2; i < argc; ++i) {
        std::string k = argv[i];
        if (k.rfind("-", 0) != 0 || i + 1 >= argc) die("bad argument " + k);
        while (!k.empty() && k[0] == '-') k.erase(0, 1);
        a.kv[k] = argv[++i];
    }
    return a;
}

// test_video.py:204-213
bool is_intra_picture(int idx, int intra_period)
{
    if (idx == 0 || intra_period == 1) return true;
    return intra_period > 1 && idx != 1 && idx % intra_period == 1;
}

// ------------------------------------------------------------------------------------ encode
int encode(const Args& a)
{
    const Geometry g = geometry(a.num("H", 0), a.num("W", 0));
    Codecs c = make_codecs(a.str("intra"), a.str("inter"));
    const bool force_intra = !c.has_inter();
    const int intra_period = force_intra ? 1 : a.num("intra-period", -1);
    const int reset_interval = a.num("reset-interval", 32);
    const int qp_i = a.num("qp-i", 32), qp_p = a.num("qp-p", qp_i);
    const int delay = c.frames_per_p;
    if (intra_period > 1 && intra_period % delay != 0) die("intra period must be a multiple of the chunk size");
    FILE* in = fopen(a.str("i").c_str(), "rb");
    if (!in) die("cannot open " + a.str("i"));
    fseek(in, 0, SEEK_END);
    const long long total = ftell(in) / static_cast<long long>(g.frame_bytes());
    fseek(in, 0, SEEK_SET);
    const int frame_num = a.has("n") ? std::min<long long>(a.num("n", 0), total) : static_cast<int>(total);
    if (frame_num <= 0) die("no pictures to code");
    DeviceBuffers b = make_buffers(g, delay);
    const int pad_b = g.Hp - g.H, pad_r = g.Wp - g.W;
    std::vector<uint8_t> out, payload;
    dcvc::stream::SpsTable sps;
    const auto t0 = std::chrono::steady_clock::now();
    int idx = 0;
    while (idx < frame_num) {
        const bool intra = is_intra_picture(idx, intra_period);
        const int want = intra ? 1 : std::min(delay, frame_num - idx);
        const int slots = intra ? 1 : delay;
        const int ldx = 3 * slots;
        for (int j = 0; j < slots; ++j) {
            if (j < want) {          // a short last chunk repeats its final picture (test_video.py:104-110)
                if (fread(b.h_yuv, 1, g.frame_bytes(), in) != g.frame_bytes()) die("short read");
                hip_ok(hipMemcpyAsync(b.yuv8, b.h_yuv, g.frame_bytes(), hipMemcpyHostToDevice, b.st), "H2D");
            }
            abi_ok(dcvc_yuv420_to_x(b.yuv8, b.yuv8 + g.y_bytes(), g.H, g.W, static_cast<char*>(b.x) + 6 * j, ldx, b.st), "yuv420_to_x");
            hip_ok(hipStreamSynchronize(b.st), "sync");      // the staging buffers are reused
        }
        int ec = 0, reset = 0, qp = qp_i;
        long long nbytes = 0;
        if (intra) {
            ec = dcvc_dmci_compress(c.intra, b.x, g.H, g.W, qp_i, pad_b, pad_r, b.x_hat, b.st);
            abi_ok(ec, "intra compress");
            nbytes = dcvc_dmci_get_stream(c.intra, nullptr, 0);
            payload.resize(static_cast<size_t>(nbytes));
            abi_ok(dcvc_dmci_get_stream(c.intra, payload.data(), payload.size()), "get_stream");
            if (c.ld) abi_ok(dcvc_dmcld_add_ref_feature_from_frame(c.ld, b.x_hat, g.Hp, g.Wp, 1, b.st), "add_ref");
            if (c.ht) abi_ok(dcvc_dmcht_add_ref_feature_from_frame(c.ht, b.x_hat, g.Hp, g.Wp, 1, b.st), "add_ref");
        } else {
            qp = qp_p;
            reset = (reset_interval > 0 && (idx + delay) % reset_interval == 1) ? 1 : 0;
            if (c.ld) {
                ec = dcvc_dmcld_compress(c.ld, b.x, g.H, g.W, qp, reset, pad_b, pad_r, b.st);
                abi_ok(ec, "inter compress");
                nbytes = dcvc_dmcld_get_stream(c.ld, nullptr, 0);
                payload.resize(static_cast<size_t>(nbytes));
                abi_ok(dcvc_dmcld_get_stream(c.ld, payload.data(), payload.size()), "get_stream");
            } else {
                ec = dcvc_dmcht_compress(c.ht, b.x, g.H, g.W, qp, reset, pad_b, pad_r, b.st);
                abi_ok(ec, "inter compress");
                nbytes = dcvc_dmcht_get_stream(c.ht, nullptr, 0);
                payload.resize(static_cast<size_t>(nbytes));
                abi_ok(dcvc_dmcht_get_stream(c.ht, payload.data(), payload.size()), "get_stream");
            }
        }
        bool is_new = false;
        const int sps_id = sps.id_for(g.H, g.W, is_new);
        if (is_new) dcvc::stream::put_sps(out, sps_id, g.H, g.W);
        dcvc::stream::put_ip(out, intra, sps_id, qp, ec, reset != 0, payload.data(), payload.size());
        idx += want;
    }
    hip_ok(hipStreamSynchronize(b.st), "sync");
    fclose(in);
    FILE* of = fopen(a.str("o").c_str(), "wb");
    if (!of || fwrite(out.data(), 1, out.size(), of) != out.size()) die("cannot write " + a.str("o"));
    fclose(of);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("encoded %d pictures (%dx%d) -> %zu bytes, %.4f bpp, %.1f pictures/s (file I/O included)\n", frame_num, g.W, g.H,
           out.size(), 8.0 * out.size() / (static_cast<double>(frame_num) * g.H * g.W), frame_num / secs);
    return 0;
}

// ------------------------------------------------------------------------------------ decode
int decode(const Args& a)
{
    Codecs c = make_codecs(a.str("intra"), a.str("inter"));
    std::vector<uint8_t> bin;
    {
        std::ifstream f(a.str("i"), std::ios::binary | std::ios::ate);
        if (!f) die("cannot open " + a.str("i"));
        bin.resize(static_cast<size_t>(f.tellg()));
        f.seekg(0);
        f.read(reinterpret_cast<char*>(bin.data()), static_cast<std::streamsize>(bin.size()));
    }
    FILE* rec = a.has("o") ? fopen(a.str("o").c_str(), "wb") : nullptr;
    FILE* ref = a.has("ref") ? fopen(a.str("ref").c_str(), "rb") : nullptr;
    if (a.has("o") && !rec) die("cannot write " + a.str("o"));
    if (a.has("ref") && !ref) die("cannot open " + a.str("ref"));
    const int limit = a.num("n", 1 << 30);
    dcvc::stream::Reader rd(bin.data(), bin.size());
    dcvc::stream::SpsTable sps;
    Geometry g;
    DeviceBuffers b;
    bool have_buffers = false;
    std::vector<uint8_t> src;
    // log (src/utils/common.py:46-116)
    std::vector<int> types;
    std::vector<double> bits, psnr, psnr_y, psnr_u, psnr_v;
    size_t pending_sps_bits = 0;
    int decoded = 0;
    const auto t0 = std::chrono::steady_clock::now();
    bool source_ended = false;
    if (c.ht && c.frames_per_p > 1 && !a.has("n") && !ref) {
        fprintf(stderr, "dcvc: warning: %d-picture chunks and neither -n nor --ref: a short last chunk is written with its "
                        "padding pictures (the container does not carry the picture count)\n", c.frames_per_p);
    }
    while (!rd.at_end() && decoded < limit && !source_ended) {
        int nal = 0, sid = 0;
        const size_t unit_start = rd.position();
        rd.header(nal, sid);
        if (nal == dcvc::stream::kSps) {
            int h = 0, w = 0;
            rd.sps_remaining(h, w);
            sps.add(sid, h, w);
            pending_sps_bits += 8 * (rd.position() - unit_start);
            continue;
        }
        const dcvc::stream::SpsTable::Sps* s = sps.find(sid);
        if (!s) die("picture refers to an unknown parameter set");
        if (!have_buffers || s->height != g.H || s->width != g.W) {
            // the size comes straight from the (untrusted) stream: refuse what no model of this family codes
            // instead of attempting a multi-terabyte allocation
            if (s->height < 2 || s->width < 2 || s->height > kMaxPictureSide || s->width > kMaxPictureSide) {
                die("unsupported picture size in the stream: " + std::to_string(s->width) + "x" + std::to_string(s->height));
            }
            if (have_buffers) free_buffers(b);     // a stream may switch parameter sets: do not leak the old set
            g = geometry(s->height, s->width);
            b = make_buffers(g, c.frames_per_p);
            src.resize(g.frame_bytes());
            have_buffers = true;
        }
        int qp = 0, ec = 0;
        bool reset = false;
        const uint8_t* payload = nullptr;
        size_t n = 0;
        rd.ip_remaining(qp, ec, reset, payload, n);
        const bool intra = nal == dcvc::stream::kIntra;
        int frames = 1;
        if (intra) {
            abi_ok(dcvc_dmci_decompress(c.intra, payload, n, qp, g.H, g.W, ec, b.x_hat, b.st), "intra decompress");
            if (c.ld) abi_ok(dcvc_dmcld_add_ref_feature_from_frame(c.ld, b.x_hat, g.Hp, g.Wp, 0, b.st), "add_ref");
            if (c.ht) abi_ok(dcvc_dmcht_add_ref_feature_from_frame(c.ht, b.x_hat, g.Hp, g.Wp, 0, b.st), "add_ref");
        } else if (c.ld) {
            abi_ok(dcvc_dmcld_decompress(c.ld, payload, n, qp, g.H, g.W, ec, reset ? 1 : 0, b.x_hat, b.st), "inter decompress");
        } else if (c.ht) {
            abi_ok(dcvc_dmcht_decompress(c.ht, payload, n, qp, g.H, g.W, ec, reset ? 1 : 0, b.x_hat, b.st), "inter decompress");
            frames = c.frames_per_p;
        } else {
            die("the stream holds P pictures but no inter model was given");
        }
        const double unit_bits = 8.0 * (rd.position() - unit_start) + pending_sps_bits;
        pending_sps_bits = 0;
        for (int j = 0; j < frames && decoded < limit; ++j) {
            const char* xh = static_cast<const char*>(b.x_hat) + static_cast<size_t>(j) * g.Hp * g.Wp * 3 * 2;
            char* y16 = static_cast<char*>(b.y16);
            abi_ok(dcvc_x_to_yuv420(xh, g.Wp, g.H, g.W, y16, y16 + g.y_bytes() * 2, b.out8, b.out8 + g.y_bytes(), b.st), "x_to_yuv420");
            // the source first: when it ends inside a chunk, the remaining pictures of the chunk are the encoder's
            // padding (repeats of the final picture) and must reach neither the log nor rec.yuv
            if (ref) {
                if (fread(src.data(), 1, g.frame_bytes(), ref) != g.frame_bytes()) {
                    if (j > 0) { source_ended = true; break; }
                    die("reference file is shorter than the stream");
                }
            }
            if (rec) {
                hip_ok(hipMemcpyAsync(b.h_yuv, b.out8, g.frame_bytes(), hipMemcpyDeviceToHost, b.st), "D2H");
                hip_ok(hipStreamSynchronize(b.st), "sync");
                if (fwrite(b.h_yuv, 1, g.frame_bytes(), rec) != g.frame_bytes()) die("short write");
            }
            if (ref) {
                hip_ok(hipMemcpyAsync(b.h_p16, b.y16, g.frame_bytes() * 2, hipMemcpyDeviceToHost, b.st), "D2H");
                hip_ok(hipStreamSynchronize(b.st), "sync");
                const size_t ny = g.y_bytes(), nc = ny / 4;
                const double py = psnr_plane(src.data(), b.h_p16, ny);
                const double pu = psnr_plane(src.data() + ny, b.h_p16 + ny, nc);
                const double pv = psnr_plane(src.data() + ny + nc, b.h_p16 + ny + nc, nc);
                psnr.push_back((6 * py + pu + pv) / 8); psnr_y.push_back(py); psnr_u.push_back(pu); psnr_v.push_back(pv);
            }
            types.push_back(intra ? 0 : 1);
            bits.push_back(j == 0 ? unit_bits : 0.0);
            ++decoded;
        }
    }
    if (have_buffers) free_buffers(b);
    if (rec) fclose(rec);
    if (ref) fclose(ref);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("decoded %d pictures (%dx%d), %.1f pictures/s (file I/O included)\n", decoded, g.W, g.H, decoded / secs);
    if (a.has("json")) {
        if (psnr.size() != types.size()) die("--json needs --ref (PSNR per picture)");
        const double px = static_cast<double>(g.H) * g.W;
        double ib = 0, pb = 0, ip[4] = {0, 0, 0, 0}, pp[4] = {0, 0, 0, 0};
        int ni = 0, np = 0;
        for (size_t i = 0; i < types.size(); ++i) {
            double* t = types[i] == 0 ? ip : pp;
            (types[i] == 0 ? ib : pb) += bits[i];
            (types[i] == 0 ? ni : np) += 1;
            t[0] += psnr[i]; t[1] += psnr_y[i]; t[2] += psnr_u[i]; t[3] += psnr_v[i];
        }
        FILE* jf = fopen(a.str("json").c_str(), "w");
        if (!jf) die("cannot write " + a.str("json"));
        const char* sfx[4] = {"", "_y", "_u", "_v"};
        fprintf(jf, "{\n  \"arith_policy\": %d,\n  \"frame_pixel_num\": %.0f,\n  \"i_frame_num\": %d,\n  \"p_frame_num\": %d,\n", dcvc_arith_policy_version(), px, ni, np);
        fprintf(jf, "  \"ave_i_frame_bpp\": %.9g,\n  \"ave_p_frame_bpp\": %.9g,\n", ni ? ib / ni / px : 0.0, np ? pb / np / px : 0.0);
        for (int k = 0; k < 4; ++k) {
            fprintf(jf, "  \"ave_i_frame_psnr%s\": %.9g,\n  \"ave_p_frame_psnr%s\": %.9g,\n  \"ave_all_frame_psnr%s\": %.9g,\n", sfx[k],
                    ni ? ip[k] / ni : 0.0, sfx[k], np ? pp[k] / np : 0.0, sfx[k], (ip[k] + pp[k]) / std::max(1, ni + np));
        }
        fprintf(jf, "  \"ave_all_frame_bpp\": %.9g,\n  \"test_time\": %.3f\n}\n", (ib + pb) / (std::max(1, ni + np) * px), secs);
        fclose(jf);
    }
    return 0;
}

}  // namespace

int main(int argc, char** argv)
{
    // one hardware queue per stream-priority level, before the HIP runtime starts ("Runtime settings":
    // with ROCclr's default of 4 the throughput of several codec objects in one process depends on their creation order);
    // a value in the environment wins
    setenv("GPU_MAX_HW_QUEUES", "1", 0);
    if (argc < 2) die("usage: dcvc encode|decode ... (see the head of dcvc_cli.hip)");
    const std::string mode = argv[1];
    try {
        const Args a = parse(argc, argv);
        if (!a.has("intra") || !a.has("i")) die("--intra and -i are required");
        if (mode == "encode") {
            if (!a.has("o")) die("-o is required");
            return encode(a);
        }
        if (mode == "decode") return decode(a);
    } catch (const std::exception& e) {
        die(e.what());
    }
    die("unknown mode " + mode);
}
