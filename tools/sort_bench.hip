// sort_bench.hip -- the device radix sort (gffx_amd/csrc/device/radix_sort.hpp) alone, on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gffx_amd/csrc/device -I include tools/sort_bench.hip -o tools/_kb/sort_bench
//   tools/_kb/sort_bench [n_records=1000000] [reps=20]
//   tools/_kb/sort_bench [n_records=1000000] [reps=20] [top digit 1|0] [seqids=25] [start bits=28]
// Sorts n {seqid, start, end} records (25 seqids, starts below 2^28: bench.py's regions) by (seqid, start), times the whole
// sort and checks the result against std::stable_sort.  With the top digit (radix_sort.hpp: the mixed-radix digit of
// (seqid, start >> 24) in place of the passes "byte 3 of start" and "seqid") the sort is four passes when the digit fits 256
// values -- 25 seqids x 28-bit starts: yes; 25 x 32-bit starts or 300 seqids: no, five (or six) passes as before.
#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "radix_sort.hpp"

namespace gffx {
int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
    return code;
}
}  // namespace gffx
using namespace gffx;

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));            \
            return 1;                                                          \
        }                                                                      \
    } while (0)

int main(int argc, char **argv) {
    const unsigned long long n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1000000ull;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const bool use_top = argc > 3 ? atoi(argv[3]) != 0 : true;
    const uint32_t n_seq = argc > 4 ? (uint32_t)atoi(argv[4]) : 25u;
    const int start_bits = argc > 5 ? atoi(argv[5]) : 28;
    std::mt19937 rng(7);
    std::vector<uint32_t> h(3 * n);
    for (unsigned long long i = 0; i < n; ++i)
        h[3 * i] = rng() % n_seq, h[3 * i + 1] = start_bits >= 32 ? rng() : rng() & ((1u << start_bits) - 1u), h[3 * i + 2] = rng();
    uint32_t *a, *b, *in, *work, *err;
    SortPlan plan{};
    for (int k = 0; k < 4; ++k) plan.word[plan.n_passes] = 1, plan.shift[plan.n_passes++] = (uint8_t)(8 * k);
    plan.word[plan.n_passes] = 0, plan.shift[plan.n_passes++] = 0;
    if (n_seq > 256) plan.word[plan.n_passes] = 0, plan.shift[plan.n_passes++] = 8;
    uint32_t *h_note = nullptr;
    CK(hipHostMalloc((void **)&h_note, 64, hipHostMallocCoherent | hipHostMallocMapped));
    h_note[0] = h_note[1] = 0;
    int passes_run = 0;
    CK(hipMalloc(&a, 12 * n));
    CK(hipMalloc(&b, 12 * n));
    CK(hipMalloc(&in, 12 * n));
    CK(hipMalloc(&err, 64));
    CK(hipMalloc(&work, DeviceSort::work_words(n, plan.n_passes) * 4));
    CK(hipMemcpy(in, h.data(), 12 * n, hipMemcpyHostToDevice));
    CK(hipMemset(err, 0, 64));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> ms;
    uint32_t *sorted = nullptr;
    for (int r = 0; r < reps + 2; ++r) {
        CK(hipMemcpyAsync(a, in, 12 * n, hipMemcpyDeviceToDevice, st));
        CK(hipEventRecord(e0, st));
        if (DeviceSort::run<3>(st, a, b, n, plan, n_seq, work, err, &sorted, SortNote{nullptr, nullptr, 0}, false, use_top ? h_note : nullptr, (uint32_t)r + 1,
                               &passes_run))
            return 1;
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        if (r >= 2) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    uint32_t herr[4];
    CK(hipMemcpy(herr, err, 16, hipMemcpyDeviceToHost));
    const char *verdict;
    std::vector<uint32_t> got(3 * n);
    CK(hipMemcpy(got.data(), sorted, 12 * n, hipMemcpyDeviceToHost));
    std::vector<uint32_t> idx(n);
    for (uint32_t i = 0; i < n; ++i) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) {
        return h[3 * (size_t)x] != h[3 * (size_t)y] ? h[3 * (size_t)x] < h[3 * (size_t)y] : h[3 * (size_t)x + 1] < h[3 * (size_t)y + 1];
    });
    bool ok = true;
    for (unsigned long long i = 0; i < n && ok; ++i)
        for (int k = 0; k < 3; ++k) ok &= got[3 * i + k] == h[3 * (size_t)idx[i] + k];
    verdict = ok ? "equal to std::stable_sort" : "WRONG";
    const double med = ms[ms.size() / 2] * 1e3;
    printf("{\"records\": %llu, \"passes\": %d, \"sort_us\": %.1f, \"min_us\": %.1f, \"us_per_pass_incl_hist\": %.2f, \"GBps\": %.0f, \"err\": %u, \"result\": \"%s\", "
           "\"items\": %d, \"lookback\": %d, \"seqids\": %u, \"start_bits\": %d}\n",
           n, passes_run, med, ms[0] * 1e3, med / passes_run, 24.0 * n * passes_run / (med * 1e-6) / 1e9, herr[0], verdict, kSortItems,
           kSortLookBack, n_seq, start_bits);
    return 0;
}
