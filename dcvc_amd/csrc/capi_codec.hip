// C ABI of the picture codecs (include/dcvc_amd_codec.h).
#include "capi_common.h"
#include "codec/dmc_ht.h"
#include "codec/dmc_ld.h"
#include "codec/dmci.h"
#include "dcvc_amd_codec.h"
#include "kernels/arith.h"

#include <cstring>

struct dcvc_dmci {
    dcvc::DmciCodec codec;
};

struct dcvc_dmcld {
    dcvc::DmcLdCodec codec;
};

struct dcvc_dmcht {
    explicit dcvc_dmcht(bool is_hts) : codec(is_hts) {}
    dcvc::DmcHtCodec codec;
};

namespace {

dcvc::ParamStore make_store(int n, const char* const* names, const void* const* data, const int* dtypes,
                            const int* ndims, const int64_t* dims)
{
    dcvc::ParamStore ps;
    const int64_t* d = dims;
    for (int i = 0; i < n; ++i) {
        ps.add(names[i], data[i], dtypes[i], d, ndims[i]);
        d += ndims[i];
    }
    return ps;
}

void check_padding16(int height, int width, int padding_b, int padding_r)
{
    const int pb = (height + 15) / 16 * 16 - height, pr = (width + 15) / 16 * 16 - width;
    if (padding_b != pb || padding_r != pr) {
        throw std::invalid_argument("compress: padding must extend the picture to multiples of 16");
    }
}

}  // namespace

extern "C" {

int dcvc_arith_policy_version(void) { return dcvc::kArithPolicyVersion; }

dcvc_dmci* dcvc_dmci_create(void)
{
    dcvc_dmci* c = nullptr;
    dcvc::guarded([&] { c = new dcvc_dmci(); });
    return c;
}

void dcvc_dmci_destroy(dcvc_dmci* c)
{
    delete c;
}

int dcvc_dmci_set_param(dcvc_dmci* c, int n, const char* const* names, const void* const* data,
                        const int* dtypes, const int* ndims, const int64_t* dims, float skip_thres)
{
    return dcvc::guarded([&] {
        c->codec.set_param(make_store(n, names, data, dtypes, ndims, dims), skip_thres);
    });
}

int dcvc_dmci_compress(dcvc_dmci* c, const void* x, int height, int width, int qp, int padding_b,
                       int padding_r, void* x_hat, void* stream)
{
    int ec = -1;
    const int rc = dcvc::guarded([&] {
        check_padding16(height, width, padding_b, padding_r);
        ec = c->codec.compress(static_cast<const dcvc::half_t*>(x), height, width, qp,
                               static_cast<dcvc::half_t*>(x_hat), static_cast<hipStream_t>(stream));
    });
    return rc < 0 ? rc : ec;
}

int64_t dcvc_dmci_get_stream(dcvc_dmci* c, uint8_t* dst, size_t cap)
{
    const auto& s = c->codec.stream_bytes();
    if (dst != nullptr) std::memcpy(dst, s.data(), s.size() < cap ? s.size() : cap);
    return static_cast<int64_t>(s.size());
}

int dcvc_dmci_decompress(dcvc_dmci* c, const uint8_t* bit_stream, size_t nbytes, int qp, int height,
                         int width, int ec_parallel, void* x_hat, void* stream)
{
    return dcvc::guarded([&] {
        c->codec.decompress(bit_stream, nbytes, qp, height, width, ec_parallel,
                            static_cast<dcvc::half_t*>(x_hat), static_cast<hipStream_t>(stream));
    });
}

int dcvc_dmci_set_use_graphs(dcvc_dmci* c, int on)
{
    return dcvc::guarded([&] { c->codec.set_use_graphs(on != 0); });
}

int64_t dcvc_dmci_debug_read(dcvc_dmci* c, const char* name, void* dst, size_t cap, void* stream)
{
    int64_t n = -1;
    const int rc = dcvc::guarded([&] {
        n = static_cast<int64_t>(c->codec.debug_read(name, dst, cap, static_cast<hipStream_t>(stream)));
    });
    return rc < 0 ? rc : n;
}

// ------------------------------------------------------------------------------------ DMC-LD
dcvc_dmcld* dcvc_dmcld_create(void)
{
    dcvc_dmcld* c = nullptr;
    dcvc::guarded([&] { c = new dcvc_dmcld(); });
    return c;
}

void dcvc_dmcld_destroy(dcvc_dmcld* c)
{
    delete c;
}

int dcvc_dmcld_set_param(dcvc_dmcld* c, int n, const char* const* names, const void* const* data,
                         const int* dtypes, const int* ndims, const int64_t* dims, float skip_thres)
{
    return dcvc::guarded([&] {
        c->codec.set_param(make_store(n, names, data, dtypes, ndims, dims), skip_thres);
    });
}

int dcvc_dmcld_add_ref_feature_from_frame(dcvc_dmcld* c, const void* frame, int height, int width,
                                          int apply_adaptor, void* stream)
{
    return dcvc::guarded([&] {
        c->codec.add_ref_feature_from_frame(static_cast<const dcvc::half_t*>(frame), height, width,
                                            apply_adaptor != 0, static_cast<hipStream_t>(stream));
    });
}

int dcvc_dmcld_compress(dcvc_dmcld* c, const void* x, int height, int width, int qp,
                        int reset_feature_memory, int padding_b, int padding_r, void* stream)
{
    int ec = -1;
    const int rc = dcvc::guarded([&] {
        check_padding16(height, width, padding_b, padding_r);
        ec = c->codec.compress(static_cast<const dcvc::half_t*>(x), height, width, qp,
                               reset_feature_memory != 0, static_cast<hipStream_t>(stream));
    });
    return rc < 0 ? rc : ec;
}

int64_t dcvc_dmcld_get_stream(dcvc_dmcld* c, uint8_t* dst, size_t cap)
{
    const auto& s = c->codec.stream_bytes();
    if (dst != nullptr) std::memcpy(dst, s.data(), s.size() < cap ? s.size() : cap);
    return static_cast<int64_t>(s.size());
}

int dcvc_dmcld_decompress(dcvc_dmcld* c, const uint8_t* bit_stream, size_t nbytes, int qp, int height,
                          int width, int ec_parallel, int reset_feature_memory, void* x_hat,
                          void* stream)
{
    return dcvc::guarded([&] {
        c->codec.decompress(bit_stream, nbytes, qp, height, width, ec_parallel, reset_feature_memory != 0,
                            static_cast<dcvc::half_t*>(x_hat), static_cast<hipStream_t>(stream));
    });
}

int64_t dcvc_dmcld_export_state(dcvc_dmcld* c, void* dst, size_t cap, void* stream)
{
    int64_t n = -1;
    const int rc = dcvc::guarded([&] {
        n = static_cast<int64_t>(c->codec.export_state(dst, cap, static_cast<hipStream_t>(stream)));
    });
    return rc < 0 ? rc : n;
}

int dcvc_dmcld_import_state(dcvc_dmcld* c, const void* src, size_t bytes, int height, int width, void* stream)
{
    return dcvc::guarded([&] { c->codec.import_state(src, bytes, height, width, static_cast<hipStream_t>(stream)); });
}

int dcvc_dmcld_set_use_graphs(dcvc_dmcld* c, int on)
{
    return dcvc::guarded([&] { c->codec.set_use_graphs(on != 0); });
}

int64_t dcvc_dmcld_debug_read(dcvc_dmcld* c, const char* name, void* dst, size_t cap, void* stream)
{
    int64_t n = -1;
    const int rc = dcvc::guarded([&] {
        n = static_cast<int64_t>(c->codec.debug_read(name, dst, cap, static_cast<hipStream_t>(stream)));
    });
    return rc < 0 ? rc : n;
}

// ------------------------------------------------------------------------------------ DMC-HT
dcvc_dmcht* dcvc_dmcht_create(int is_hts)
{
    dcvc_dmcht* c = nullptr;
    dcvc::guarded([&] { c = new dcvc_dmcht(is_hts != 0); });
    return c;
}

void dcvc_dmcht_destroy(dcvc_dmcht* c)
{
    delete c;
}

int dcvc_dmcht_set_param(dcvc_dmcht* c, int n, const char* const* names, const void* const* data,
                         const int* dtypes, const int* ndims, const int64_t* dims, float skip_thres)
{
    return dcvc::guarded([&] {
        c->codec.set_param(make_store(n, names, data, dtypes, ndims, dims), skip_thres);
    });
}

int dcvc_dmcht_add_ref_feature_from_frame(dcvc_dmcht* c, const void* frame, int height, int width,
                                          int apply_adaptor, void* stream)
{
    return dcvc::guarded([&] {
        c->codec.add_ref_feature_from_frame(static_cast<const dcvc::half_t*>(frame), height, width,
                                            apply_adaptor != 0, static_cast<hipStream_t>(stream));
    });
}

int dcvc_dmcht_compress(dcvc_dmcht* c, const void* x, int height, int width, int qp,
                        int reset_feature_memory, int padding_b, int padding_r, void* stream)
{
    int ec = -1;
    const int rc = dcvc::guarded([&] {
        check_padding16(height, width, padding_b, padding_r);
        ec = c->codec.compress(static_cast<const dcvc::half_t*>(x), height, width, qp,
                               reset_feature_memory != 0, static_cast<hipStream_t>(stream));
    });
    return rc < 0 ? rc : ec;
}

int64_t dcvc_dmcht_get_stream(dcvc_dmcht* c, uint8_t* dst, size_t cap)
{
    const auto& s = c->codec.stream_bytes();
    if (dst != nullptr) std::memcpy(dst, s.data(), s.size() < cap ? s.size() : cap);
    return static_cast<int64_t>(s.size());
}

int dcvc_dmcht_decompress(dcvc_dmcht* c, const uint8_t* bit_stream, size_t nbytes, int qp, int height,
                          int width, int ec_parallel, int reset_feature_memory, void* x_hat,
                          void* stream)
{
    return dcvc::guarded([&] {
        c->codec.decompress(bit_stream, nbytes, qp, height, width, ec_parallel, reset_feature_memory != 0,
                            static_cast<dcvc::half_t*>(x_hat), static_cast<hipStream_t>(stream));
    });
}

int64_t dcvc_dmcht_export_state(dcvc_dmcht* c, void* dst, size_t cap, void* stream)
{
    int64_t n = -1;
    const int rc = dcvc::guarded([&] {
        n = static_cast<int64_t>(c->codec.export_state(dst, cap, static_cast<hipStream_t>(stream)));
    });
    return rc < 0 ? rc : n;
}

int dcvc_dmcht_import_state(dcvc_dmcht* c, const void* src, size_t bytes, int height, int width, void* stream)
{
    return dcvc::guarded([&] { c->codec.import_state(src, bytes, height, width, static_cast<hipStream_t>(stream)); });
}

int dcvc_dmcht_set_recon_mask(dcvc_dmcht* c, unsigned mask)
{
    return dcvc::guarded([&] { c->codec.set_recon_mask(mask); });
}

int64_t dcvc_dmcht_export_feature(dcvc_dmcht* c, void* dst, size_t cap, void* stream)
{
    int64_t n = -1;
    const int rc = dcvc::guarded([&] {
        n = static_cast<int64_t>(c->codec.export_feature(dst, cap, static_cast<hipStream_t>(stream)));
    });
    return rc < 0 ? rc : n;
}

int dcvc_dmcht_import_feature(dcvc_dmcht* c, const void* src, size_t bytes, int height, int width, void* stream)
{
    return dcvc::guarded([&] { c->codec.import_feature(src, bytes, height, width, static_cast<hipStream_t>(stream)); });
}

int dcvc_dmcht_run_recon_heads(dcvc_dmcht* c, unsigned mask, void* x_hat, void* stream)
{
    return dcvc::guarded([&] {
        c->codec.run_recon_heads(mask, static_cast<dcvc::half_t*>(x_hat), static_cast<hipStream_t>(stream));
    });
}

int dcvc_dmcht_set_use_graphs(dcvc_dmcht* c, int on)
{
    return dcvc::guarded([&] { c->codec.set_use_graphs(on != 0); });
}

int64_t dcvc_dmcht_debug_read(dcvc_dmcht* c, const char* name, void* dst, size_t cap, void* stream)
{
    int64_t n = -1;
    const int rc = dcvc::guarded([&] {
        n = static_cast<int64_t>(c->codec.debug_read(name, dst, cap, static_cast<hipStream_t>(stream)));
    });
    return rc < 0 ? rc : n;
}

}  // extern "C"
