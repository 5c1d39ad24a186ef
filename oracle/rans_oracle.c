/*
 * rans_oracle.c - TEST INFRASTRUCTURE ONLY (parity oracle). Never linked into the product.
 *
 * Plain, single-threaded C restatement of the reference rANS coder:
 *   state update / renormalisation / bypass : /root/reference/src/cpp/py_rans/rans.cpp:31-106
 *   value mapping, escape coding            : rans.cpp:108-181
 *   y / z symbol loops                      : rans.cpp:239-257, 276-294, 417-429, 452-463
 *   sub-stream split + container            : py_rans.cpp:13-33, 104-249, 412-492
 *   pmf -> quantised cdf                    : py_rans.cpp:35-94
 * Pinned against the compiled reference (oracle/_ref) by tests/test_rans.py and against the
 * committed vectors in tests/golden/rans_*.npz.
 *
 * Interface (ctypes): a tiny "job" API - the caller hands every segment up front.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PROB_BITS 16
#define RANS_L (1u << 23)
#define ENC_SHIFT (23 - PROB_BITS + 8)
#define BYPASS_BITS 2
#define BYPASS_MAX 3u
#define MAX_PAR 8

typedef struct {
    const int32_t* cdf;      /* [num][stride] */
    const int32_t* sizes;    /* [num] cdf lengths; escape value = size - 2 */
    int num, stride;
} orc_table;

typedef struct {
    int is_y;                /* 1: int16 (sym<<8)+idx, table 1; 0: int8, table 0 */
    const void* data;
    int count, cdf_offset, ch;
} orc_segment;

/* ------------------------------------------------------------------ encoding */
typedef struct {
    uint32_t r;
    uint8_t* p;
} enc_t;

static void put(enc_t* e, uint32_t start, uint32_t freq)
{
    const uint32_t r_max = freq << ENC_SHIFT;
    while (e->r >= r_max) {
        *--e->p = (uint8_t)e->r;
        e->r >>= 8;
    }
    e->r = ((e->r / freq) << PROB_BITS) + (e->r % freq) + start;
}

static void put_bits(enc_t* e, uint32_t val)
{
    const uint32_t freq = 1u << (PROB_BITS - BYPASS_BITS);
    const uint32_t r_max = freq << ENC_SHIFT;
    while (e->r >= r_max) {
        *--e->p = (uint8_t)e->r;
        e->r >>= 8;
    }
    e->r = (e->r << BYPASS_BITS) | val;
}

static void encode_one(enc_t* e, int32_t sym, const orc_table* t, int idx)
{
    const int32_t max_value = (int8_t)(t->sizes[idx] - 2);
    const int32_t* cdf = t->cdf + (size_t)idx * t->stride;
    int32_t value = abs(sym) * 2 - (sym > 0);
    if (value >= max_value) {
        uint16_t bins[40];
        int nb = 0, n_bypass = 0, j;
        const uint32_t raw = (uint32_t)(value - max_value);
        int32_t v;
        value = max_value;
        while ((raw >> (n_bypass * BYPASS_BITS)) != 0) {
            n_bypass++;
        }
        v = n_bypass;
        while (v >= (int32_t)BYPASS_MAX) {
            bins[nb++] = BYPASS_MAX;
            v -= BYPASS_MAX;
        }
        bins[nb++] = (uint16_t)v;
        for (j = 0; j < n_bypass; j++) {
            bins[nb++] = (uint16_t)((raw >> (j * BYPASS_BITS)) & BYPASS_MAX);
        }
        for (j = nb - 1; j >= 0; j--) {
            put_bits(e, bins[j]);
        }
    }
    put(e, (uint16_t)cdf[value], (uint16_t)(cdf[value + 1] - cdf[value]));
}

static void split(int count, int n, int i, int* begin, int* len)
{
    const int size0 = count / n;
    *begin = size0 * i;
    *len = (i == n - 1) ? count - size0 * (n - 1) : size0;
}

static int identical_tail(const uint8_t* a, int na, const uint8_t* b, int nb)
{
    int same = 0, i;
    int check = na < nb ? na : nb;
    if (check > 8) {
        check = 8;
    }
    for (i = 0; i < check; i++) {
        if (a[na - 1 - i] != 0) {
            break;
        }
        if (b[nb - 1 - i] != 0) {
            break;
        }
        same++;
    }
    if (same == 0 && a[na - 1] == b[nb - 1]) {
        same = 1;
    }
    return same;
}

/* Encode all segments (in call order) with n sub-streams; returns the container size, writes at
 * most cap bytes to out. */
int64_t orc_rans_encode(const orc_table* tables /* [2] */, const orc_segment* segs, int n_segs,
                        int n, uint8_t* out, int64_t cap)
{
    uint8_t* buf[MAX_PAR];
    uint8_t* beg[MAX_PAR];
    int len[MAX_PAR];
    int64_t total = 0;
    int i, s, k;
    for (i = 0; i < n; i++) {
        size_t symbols = 0, bytes;
        enc_t e;
        for (s = 0; s < n_segs; s++) {
            int b, l;
            split(segs[s].count, n, i, &b, &l);
            symbols += (size_t)l;
        }
        bytes = symbols * 4 + 16;
        buf[i] = (uint8_t*)malloc(bytes);
        e.r = RANS_L;
        e.p = buf[i] + bytes;
        for (s = 0; s < n_segs; s++) {
            int b, l;
            split(segs[s].count, n, i, &b, &l);
            if (segs[s].is_y) {
                const int16_t* y = (const int16_t*)segs[s].data;
                for (k = b + l - 1; k >= b; k--) {
                    const int16_t c = y[k];
                    encode_one(&e, (int8_t)(c >> 8), &tables[1], c & 0xff);
                }
            } else {
                const int8_t* z = (const int8_t*)segs[s].data;
                for (k = b + l - 1; k >= b; k--) {
                    encode_one(&e, z[k], &tables[0], (k % segs[s].ch) + segs[s].cdf_offset);
                }
            }
        }
        e.p -= 4;
        e.p[0] = (uint8_t)(e.r >> 0);
        e.p[1] = (uint8_t)(e.r >> 8);
        e.p[2] = (uint8_t)(e.r >> 16);
        e.p[3] = (uint8_t)(e.r >> 24);
        beg[i] = e.p;
        len[i] = (int)(buf[i] + bytes - e.p);
    }

    if (n == 1) {
        total = len[0];
        if (total <= cap) {
            memcpy(out, beg[0], (size_t)total);
        }
    } else {
        const int pairs = n / 2, tail = n % 2;
        const int n_off = pairs - 1 + tail;
        int group[MAX_PAR / 2], same[MAX_PAR / 2], p, cumulative;
        int64_t pos;
        total = (int64_t)n_off * 4;
        for (p = 0; p < pairs; p++) {
            same[p] = identical_tail(beg[2 * p], len[2 * p], beg[2 * p + 1], len[2 * p + 1]);
            group[p] = len[2 * p] + len[2 * p + 1] - same[p];
            total += group[p];
        }
        if (tail) {
            total += len[n - 1];
        }
        if (total <= cap) {
            cumulative = group[0];
            for (k = 0; k < n_off; k++) {
                const int32_t v = cumulative;
                memcpy(out + 4 * k, &v, 4); /* little-endian host, as the reference */
                if (k + 1 < pairs) {
                    cumulative += group[k + 1];
                }
            }
            pos = (int64_t)n_off * 4;
            for (p = 0; p < pairs; p++) {
                const uint8_t* a = beg[2 * p];
                const uint8_t* b = beg[2 * p + 1];
                const int nb = len[2 * p + 1] - same[p];
                memcpy(out + pos, a, (size_t)len[2 * p]);
                for (k = 0; k < nb; k++) {
                    out[pos + len[2 * p] + k] = b[nb - 1 - k];
                }
                pos += group[p];
            }
            if (tail) {
                memcpy(out + pos, beg[n - 1], (size_t)len[n - 1]);
            }
        }
    }
    for (i = 0; i < n; i++) {
        free(buf[i]);
    }
    return total;
}

/* ------------------------------------------------------------------ decoding */
typedef struct {
    uint32_t r;
    const uint8_t* p;
} dec_t;

typedef struct {
    uint8_t* bytes[MAX_PAR];
    dec_t st[MAX_PAR];
    int n;
} orc_decoder;

static uint32_t get_bits(dec_t* d)
{
    const uint32_t val = d->r & BYPASS_MAX;
    d->r >>= BYPASS_BITS;
    if (d->r < RANS_L) {
        d->r = (d->r << 8) | *d->p++;
    }
    return val;
}

static int8_t decode_one(dec_t* d, const orc_table* t, int idx)
{
    const int32_t max_value = (int8_t)(t->sizes[idx] - 2);
    const int32_t* cdf = t->cdf + (size_t)idx * t->stride;
    const int32_t cum = (int32_t)(d->r & 0xffffu);
    int32_t value;
    int s = 1;
    while (cdf[s] <= cum) {
        s++;
    }
    s--;
    d->r = (uint32_t)(cdf[s + 1] - cdf[s]) * (d->r >> PROB_BITS) + (d->r & 0xffffu) - (uint32_t)cdf[s];
    while (d->r < RANS_L) {
        d->r = (d->r << 8) | *d->p++;
    }
    value = s;
    if (value == max_value) {
        int32_t val = (int32_t)get_bits(d);
        int32_t n_bypass = val, raw = 0, j;
        while (val == (int32_t)BYPASS_MAX) {
            val = (int32_t)get_bits(d);
            n_bypass += val;
        }
        for (j = 0; j < n_bypass; j++) {
            val = (int32_t)get_bits(d);
            raw |= val << (j * BYPASS_BITS);
        }
        value = raw + max_value;
    }
    return (int8_t)((value % 2 == 1) ? (value + 1) / 2 : -(value + 1) / 2);
}

static void open_stream(orc_decoder* d, int i, const uint8_t* src, int64_t size, int reversed)
{
    int64_t k;
    /* slack so that a final renormalisation byte read stays inside the allocation */
    d->bytes[i] = (uint8_t*)calloc((size_t)size + 16, 1);
    for (k = 0; k < size; k++) {
        d->bytes[i][k] = reversed ? src[size - 1 - k] : src[k];
    }
    d->st[i].p = d->bytes[i];
    d->st[i].r = (uint32_t)d->st[i].p[0] | ((uint32_t)d->st[i].p[1] << 8)
                 | ((uint32_t)d->st[i].p[2] << 16) | ((uint32_t)d->st[i].p[3] << 24);
    d->st[i].p += 4;
}

orc_decoder* orc_rans_decoder_open(const uint8_t* data, int64_t size, int n)
{
    orc_decoder* d = (orc_decoder*)calloc(1, sizeof(orc_decoder));
    d->n = n;
    if (n == 1) {
        open_stream(d, 0, data, size, 0);
    } else if (n == 2) {
        open_stream(d, 0, data, size, 0);
        open_stream(d, 1, data, size, 1);
    } else {
        const int pairs = n / 2, tail = n % 2;
        const int n_off = pairs - 1 + tail;
        const uint8_t* payload = data + 4 * n_off;
        const int64_t payload_size = size - 4 * n_off;
        int32_t off[MAX_PAR];
        int p;
        for (p = 0; p < n_off; p++) {
            memcpy(&off[p], data + 4 * p, 4);
        }
        for (p = 0; p < pairs; p++) {
            const int64_t b = p == 0 ? 0 : off[p - 1];
            int64_t e;
            if (p < n_off) {
                e = off[p];
            } else {
                e = tail ? off[n_off - 1] : payload_size;
            }
            open_stream(d, 2 * p, payload + b, e - b, 0);
            open_stream(d, 2 * p + 1, payload + b, e - b, 1);
        }
        if (tail) {
            const int64_t b = off[n_off - 1];
            open_stream(d, n - 1, payload + b, payload_size - b, 0);
        }
    }
    return d;
}

void orc_rans_decoder_close(orc_decoder* d)
{
    int i;
    for (i = 0; i < d->n; i++) {
        free(d->bytes[i]);
    }
    free(d);
}

void orc_rans_decode_y(orc_decoder* d, const orc_table* tables, const uint8_t* indexes, int count,
                       int8_t* out)
{
    int i, k;
    for (i = 0; i < d->n; i++) {
        int b, l;
        split(count, d->n, i, &b, &l);
        for (k = b; k < b + l; k++) {
            out[k] = decode_one(&d->st[i], &tables[1], indexes[k]);
        }
    }
}

void orc_rans_decode_z(orc_decoder* d, const orc_table* tables, int count, int cdf_offset, int ch,
                       int8_t* out)
{
    int i, k;
    for (i = 0; i < d->n; i++) {
        int b, l;
        split(count, d->n, i, &b, &l);
        for (k = b; k < b + l; k++) {
            out[k] = decode_one(&d->st[i], &tables[0], (k % ch) + cdf_offset);
        }
    }
}

/* ------------------------------------------------------------------ pmf -> cdf */
int orc_pmf_to_quantized_cdf(const float* pmf, int n, uint32_t* cdf)
{
    const uint32_t prob_max = 1u << PROB_BITS;
    const int m = n + 1;
    uint32_t total = 0;
    int i, j;
    cdf[0] = 0;
    for (i = 0; i < n; i++) {
        cdf[i + 1] = (uint32_t)(pmf[i] * prob_max + 0.5);
    }
    for (i = 0; i < m; i++) {
        total += cdf[i];
    }
    for (i = 0; i < m; i++) {
        cdf[i] = (uint32_t)(((uint64_t)prob_max * cdf[i]) / total);
    }
    for (i = 1; i < m; i++) {
        cdf[i] += cdf[i - 1];
    }
    cdf[m - 1] = prob_max;
    for (i = 0; i < m - 1; i++) {
        if (cdf[i] + 1 > cdf[i + 1]) {
            uint32_t best_freq = ~0u;
            int best = -1;
            for (j = 0; j < m - 1; j++) {
                const uint32_t freq = cdf[j + 1] - cdf[j];
                if (freq >= 2 && freq < best_freq) {
                    best_freq = freq;
                    best = j;
                }
            }
            if (best < 0) {
                return -1;
            }
            if (best < i) {
                for (j = best + 1; j <= i; j++) {
                    cdf[j] -= 1;
                }
            } else {
                for (j = i + 1; j <= best; j++) {
                    cdf[j] += 1;
                }
            }
        }
    }
    return 0;
}
