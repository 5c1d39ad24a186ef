// mfma_probe - measures the ARITHMETIC of v_mfma_f32_32x32x16_f16 on gfx950 so that the CPU
// oracle can restate it bit for bit (DESIGN.md "arithmetic policy").
//
// Every trial is one MFMA: D = A(32x16) * B(16x32) + C(32x32), one wave. The host generates the
// operands (designed cases and random ones), the device result D is dumped next to them and
// analysed offline by tools/analyze_mfma_probe.py.
//
//   designed trials: every row of A is the same vector a[16], every column of B the same b[16],
//                    C is constant c  ->  all 1024 outputs equal  d = c + sum_k a_k b_k
//                    (we record d(0,0) and how many outputs differ from it).
//   random trials  : full random A, B, C at several exponent spreads; all of D is recorded.
//
// Output: gpurun_out/mfma_probe.bin
//   header  int32 {magic 0x4d464d41, n_designed, n_random}
//   designed[n_designed]: half a[16], half b[16], float c, float d, int32 n_mismatch
//   random[n_random]    : half A[32][16], half B[16][32] (k-major), float C[32][32], float D[32][32]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(2);                                                              \
        }                                                                         \
    } while (0)

// A: [T][32][16] row-major (k contiguous); B: [T][32][16] stored column-major i.e. Bt[j][k];
// C, D: [T][32][32] row-major D[i][j].
__global__ void mfma_trials(const _Float16* __restrict__ A, const _Float16* __restrict__ Bt,
                            const float* __restrict__ C, float* __restrict__ D)
{
    const int t = blockIdx.x;
    const int lane = threadIdx.x;
    const int rc = lane & 31;        // row of A / column of B held by this lane
    const int kh = lane >> 5;        // which 8-wide half of k
    half8 a = *reinterpret_cast<const half8*>(A + (size_t)t * 512 + rc * 16 + kh * 8);
    half8 b = *reinterpret_cast<const half8*>(Bt + (size_t)t * 512 + rc * 16 + kh * 8);
    float16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;   // C/D layout: col = lane&31
        acc[r] = C[(size_t)t * 1024 + row * 32 + rc];
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
        D[(size_t)t * 1024 + row * 32 + rc] = acc[r];
    }
}

static uint64_t g_rng = 0x9e3779b97f4a7c15ull;
static uint32_t rnd()
{
    g_rng ^= g_rng << 13;
    g_rng ^= g_rng >> 7;
    g_rng ^= g_rng << 17;
    return (uint32_t)(g_rng >> 11);
}
static double urand() { return (rnd() & 0xffffff) / 16777216.0; }

struct Designed {
    _Float16 a[16], b[16];
    float c;
};

static _Float16 H(double v) { return (_Float16)v; }

int main(int argc, char** argv)
{
    const char* out_path = argc > 1 ? argv[1] : "gpurun_out/mfma_probe.bin";
    std::vector<Designed> des;
    auto blank = [] {
        Designed d;
        for (int k = 0; k < 16; ++k) {
            d.a[k] = H(0);
            d.b[k] = H(0);
        }
        d.c = 0.f;
        return d;
    };
    // F1: one product against C = 1 (rounding of a single addition), every k position.
    {
        const double pa[] = { 0x1p-12, 1.5 * 0x1p-12, 0x1p-13, (1 + 0x1p-10) * 0x1p-12, 0x1.8p-11 };
        const double pb[] = { 0x1p-12, 0x1p-13, 0x1p-12, 0x1p-12, 0x1p-12 };
        for (int k = 0; k < 16; ++k)
            for (int v = 0; v < 5; ++v)
                for (int s = 0; s < 2; ++s)
                    for (int cs = 0; cs < 2; ++cs) {
                        Designed d = blank();
                        d.a[k] = H(s ? -pa[v] : pa[v]);
                        d.b[k] = H(pb[v]);
                        d.c = cs ? -1.f : 1.f;
                        des.push_back(d);
                    }
    }
    // F2: two half-ulp products against C = 1 (are products summed exactly before meeting C?)
    for (int k1 = 0; k1 < 16; ++k1)
        for (int k2 = k1 + 1; k2 < 16; ++k2) {
            Designed d = blank();
            d.a[k1] = H(0x1p-12);
            d.b[k1] = H(0x1p-12);
            d.a[k2] = H(0x1p-12);
            d.b[k2] = H(0x1p-12);
            d.c = 1.f;
            des.push_back(d);
        }
    // F3: big product 1.0 at k1, half-ulp product at k2, C = half ulp (is C inside the exact sum?)
    for (int k1 = 0; k1 < 16; ++k1)
        for (int k2 = 0; k2 < 16; ++k2) {
            if (k1 == k2) continue;
            Designed d = blank();
            d.a[k1] = H(1.0);
            d.b[k1] = H(1.0);
            d.a[k2] = H(0x1p-12);
            d.b[k2] = H(0x1p-12);
            d.c = 0x1p-24f;
            des.push_back(d);
        }
    // F4: cancellation +2^E, -2^E and a small survivor 2^-e as a product / as C.
    {
        const int trip[][3] = { { 0, 1, 2 }, { 0, 8, 15 }, { 3, 4, 12 }, { 15, 0, 7 }, { 5, 13, 6 }, { 2, 10, 3 } };
        for (int E = 0; E <= 30; E += 5)
            for (int e = 0; e <= 28; e += 1)
                for (auto& tr : trip)
                    for (int mode = 0; mode < 2; ++mode) {
                        Designed d = blank();
                        const int ea = E / 2, eb = E - ea;
                        d.a[tr[0]] = H(std::ldexp(1.0, ea));
                        d.b[tr[0]] = H(std::ldexp(1.0, eb));
                        d.a[tr[1]] = H(-std::ldexp(1.0, ea));
                        d.b[tr[1]] = H(std::ldexp(1.0, eb));
                        if (mode == 0) {
                            const int sa = e / 2, sb = e - sa;
                            d.a[tr[2]] = H(std::ldexp(1.0, -sa));
                            d.b[tr[2]] = H(std::ldexp(1.5, -sb));
                        } else {
                            d.c = std::ldexp(1.5f, -e);
                        }
                        des.push_back(d);
                    }
    }
    // F5: fp16 subnormal inputs and fp32 subnormal C.
    for (int k = 0; k < 16; k += 5) {
        Designed d = blank();
        d.a[k] = H(0x1p-24);   // smallest fp16 subnormal
        d.b[k] = H(1.0);
        des.push_back(d);
        d = blank();
        d.a[k] = H(3 * 0x1p-24);
        d.b[k] = H(0x1p-24);
        des.push_back(d);
        d = blank();
        d.a[k] = H(0x1p-15);   // subnormal
        d.b[k] = H(0x1.8p3);
        d.c = 1.0f;
        des.push_back(d);
        d = blank();
        d.c = 0x1p-140f;       // fp32 subnormal passes through?
        des.push_back(d);
        d = blank();
        d.a[k] = H(0x1p-14);
        d.b[k] = H(0x1p-14);
        d.c = 0x1p-140f;
        des.push_back(d);
    }
    // F6: ordering probe - 16 products of descending / ascending magnitude with ties
    for (int rep = 0; rep < 64; ++rep) {
        Designed d = blank();
        for (int k = 0; k < 16; ++k) {
            const int e = (int)(rnd() % 24);
            d.a[k] = H(std::ldexp(1.0 + (rnd() % 1024) / 1024.0, -(e / 2)) * ((rnd() & 1) ? -1 : 1));
            d.b[k] = H(std::ldexp(1.0 + (rnd() % 1024) / 1024.0, -(e - e / 2)));
        }
        d.c = (float)std::ldexp(1.0 + urand(), -(int)(rnd() % 12)) * ((rnd() & 1) ? -1.f : 1.f);
        des.push_back(d);
    }

    const int n_des = (int)des.size();
    const int n_rand = 1536;
    const int T = n_des + n_rand;
    std::vector<_Float16> hA((size_t)T * 512), hB((size_t)T * 512);
    std::vector<float> hC((size_t)T * 1024), hD((size_t)T * 1024);
    for (int t = 0; t < n_des; ++t) {
        for (int i = 0; i < 32; ++i)
            for (int k = 0; k < 16; ++k) {
                hA[(size_t)t * 512 + i * 16 + k] = des[t].a[k];
                hB[(size_t)t * 512 + i * 16 + k] = des[t].b[k];
            }
        for (int i = 0; i < 1024; ++i) hC[(size_t)t * 1024 + i] = des[t].c;
    }
    for (int t = n_des; t < T; ++t) {
        const int kind = (t - n_des) % 6;
        const int spread = kind == 0 ? 0 : kind == 1 ? 4 : kind == 2 ? 10 : kind == 3 ? 16 : kind == 4 ? 22 : 30;
        for (int i = 0; i < 512; ++i) {
            const int e1 = spread ? (int)(rnd() % (spread + 1)) : 0;
            const int e2 = spread ? (int)(rnd() % (spread + 1)) : 0;
            double va = (urand() * 2 - 1) * std::ldexp(1.0, 4 - e1);
            double vb = (urand() * 2 - 1) * std::ldexp(1.0, 4 - e2);
            if (kind == 5 && (rnd() % 8) == 0) va = std::ldexp((double)(rnd() % 1024), -24);  // subnormal
            hA[(size_t)t * 512 + i] = H(va);
            hB[(size_t)t * 512 + i] = H(vb);
        }
        for (int i = 0; i < 1024; ++i) {
            const int e = spread ? (int)(rnd() % (spread + 1)) : 0;
            hC[(size_t)t * 1024 + i] = (rnd() % 4 == 0) ? 0.f : (float)((urand() * 2 - 1) * std::ldexp(1.0, 6 - e));
        }
    }

    _Float16 *dA, *dB;
    float *dC, *dD;
    CHECK(hipMalloc(&dA, hA.size() * 2));
    CHECK(hipMalloc(&dB, hB.size() * 2));
    CHECK(hipMalloc(&dC, hC.size() * 4));
    CHECK(hipMalloc(&dD, hD.size() * 4));
    CHECK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dC, hC.data(), hC.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mfma_trials, dim3(T), dim3(64), 0, 0, dA, dB, dC, dD);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(hD.data(), dD, hD.size() * 4, hipMemcpyDeviceToHost));

    FILE* f = fopen(out_path, "wb");
    if (!f) {
        perror(out_path);
        return 3;
    }
    const int32_t hdr[3] = { 0x4d464d41, n_des, n_rand };
    fwrite(hdr, 4, 3, f);
    for (int t = 0; t < n_des; ++t) {
        fwrite(des[t].a, 2, 16, f);
        fwrite(des[t].b, 2, 16, f);
        fwrite(&des[t].c, 4, 1, f);
        const float d0 = hD[(size_t)t * 1024];
        int32_t mism = 0;
        for (int i = 0; i < 1024; ++i) {
            uint32_t x, y;
            memcpy(&x, &hD[(size_t)t * 1024 + i], 4);
            memcpy(&y, &d0, 4);
            mism += (x != y);
        }
        fwrite(&d0, 4, 1, f);
        fwrite(&mism, 4, 1, f);
    }
    for (int t = n_des; t < T; ++t) {
        fwrite(&hA[(size_t)t * 512], 2, 512, f);
        fwrite(&hB[(size_t)t * 512], 2, 512, f);
        fwrite(&hC[(size_t)t * 1024], 4, 1024, f);
        fwrite(&hD[(size_t)t * 1024], 4, 1024, f);
    }
    fclose(f);
    printf("mfma_probe: %d designed + %d random trials -> %s\n", n_des, n_rand, out_path);
    return 0;
}
