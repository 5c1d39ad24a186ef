// C ABI of the host rANS coder (include/dcvc_amd_rans.h).
#include "capi_common.h"
#include "dcvc_amd_rans.h"
#include "rans/rans_coder.h"

#include <cstring>
#include <deque>

struct dcvc_rans_encoder {
    dcvc::RansEncoder enc;
    // the Python-facing calls hand over temporaries, so segments are kept here until flush
    std::deque<std::vector<int16_t>> keep_y;
    std::deque<std::vector<int8_t>> keep_z;
};

struct dcvc_rans_decoder {
    dcvc::RansDecoder dec;
};

extern "C" {

const char* dcvc_last_error(void)
{
    return dcvc::last_error().c_str();
}

int dcvc_pmf_to_quantized_cdf(const float* pmf, int n, uint32_t* cdf_out)
{
    return dcvc::guarded([&] {
        const auto cdf = dcvc::pmf_to_quantized_cdf(pmf, n);
        std::memcpy(cdf_out, cdf.data(), cdf.size() * sizeof(uint32_t));
    });
}

dcvc_rans_encoder* dcvc_rans_encoder_create(void)
{
    dcvc_rans_encoder* e = nullptr;
    dcvc::guarded([&] { e = new dcvc_rans_encoder(); });
    return e;
}

void dcvc_rans_encoder_destroy(dcvc_rans_encoder* e)
{
    delete e;
}

int dcvc_rans_encoder_set_cdf(dcvc_rans_encoder* e, const int32_t* cdfs, int num_cdf, int stride,
                              const int32_t* cdf_sizes, int index)
{
    return dcvc::guarded([&] { e->enc.set_cdf(cdfs, num_cdf, stride, cdf_sizes, index); });
}

int dcvc_rans_encoder_set_entropy_coder_parallel(dcvc_rans_encoder* e, int n)
{
    return dcvc::guarded([&] { e->enc.set_parallel(n); });
}

int dcvc_rans_encoder_reset(dcvc_rans_encoder* e)
{
    return dcvc::guarded([&] {
        e->enc.reset();
        e->keep_y.clear();
        e->keep_z.clear();
    });
}

int dcvc_rans_encoder_encode_y(dcvc_rans_encoder* e, const int16_t* symbols, int count)
{
    return dcvc::guarded([&] {
        e->keep_y.emplace_back(symbols, symbols + count);
        e->enc.push_y(e->keep_y.back().data(), count);
    });
}

int dcvc_rans_encoder_encode_z(dcvc_rans_encoder* e, const int8_t* symbols, int count,
                               int cdf_offset, int ch)
{
    return dcvc::guarded([&] {
        e->keep_z.emplace_back(symbols, symbols + count);
        e->enc.push_z(e->keep_z.back().data(), count, cdf_offset, ch);
    });
}

int dcvc_rans_encoder_flush(dcvc_rans_encoder* e)
{
    return dcvc::guarded([&] { e->enc.flush(); });
}

int64_t dcvc_rans_encoder_get_encoded_stream(dcvc_rans_encoder* e, uint8_t* dst, size_t cap)
{
    const auto& s = e->enc.stream();
    if (dst != nullptr) {
        std::memcpy(dst, s.data(), s.size() < cap ? s.size() : cap);
    }
    return static_cast<int64_t>(s.size());
}

dcvc_rans_decoder* dcvc_rans_decoder_create(void)
{
    dcvc_rans_decoder* d = nullptr;
    dcvc::guarded([&] { d = new dcvc_rans_decoder(); });
    return d;
}

void dcvc_rans_decoder_destroy(dcvc_rans_decoder* d)
{
    delete d;
}

int dcvc_rans_decoder_set_cdf(dcvc_rans_decoder* d, const int32_t* cdfs, int num_cdf, int stride,
                              const int32_t* cdf_sizes, int index)
{
    return dcvc::guarded([&] { d->dec.set_cdf(cdfs, num_cdf, stride, cdf_sizes, index); });
}

int dcvc_rans_decoder_set_entropy_coder_parallel(dcvc_rans_decoder* d, int n)
{
    return dcvc::guarded([&] { d->dec.set_parallel(n); });
}

int dcvc_rans_decoder_set_stream(dcvc_rans_decoder* d, const uint8_t* data, size_t size)
{
    return dcvc::guarded([&] { d->dec.set_stream(data, size); });
}

int dcvc_rans_decoder_decode_y(dcvc_rans_decoder* d, const uint8_t* indexes, int count, int8_t* out)
{
    return dcvc::guarded([&] { d->dec.decode_y(indexes, count, out); });
}

int dcvc_rans_decoder_decode_z(dcvc_rans_decoder* d, int count, int cdf_offset, int ch, int8_t* out)
{
    return dcvc::guarded([&] { d->dec.decode_z(count, cdf_offset, ch, out); });
}

}  // extern "C"
