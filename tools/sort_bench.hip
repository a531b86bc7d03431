// sort_bench.hip -- the device radix sort (gffx_amd/csrc/device/radix_sort.hpp) alone, on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gffx_amd/csrc/device -I include tools/sort_bench.hip -o tools/_kb/sort_bench
//   tools/_kb/sort_bench [n_records=1000000] [reps=20]
// Sorts n {seqid, start, end} records (25 seqids, starts below 2^28: bench.py's regions) by (seqid, start), times the whole
// sort and checks the result against std::stable_sort (skipped for a GFFX_SORT_ABL timing build, whose output is wrong).
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
    std::mt19937 rng(7);
    std::vector<uint32_t> h(3 * n);
    for (unsigned long long i = 0; i < n; ++i) h[3 * i] = rng() % 25, h[3 * i + 1] = rng() & 0x0FFFFFFFu, h[3 * i + 2] = rng();
    uint32_t *a, *b, *in, *work, *err;
    SortPlan plan{};
    for (int k = 0; k < 4; ++k) plan.word[plan.n_passes] = 1, plan.shift[plan.n_passes++] = (uint8_t)(8 * k);
    plan.word[plan.n_passes] = 0, plan.shift[plan.n_passes++] = 0;
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
        if (DeviceSort::run<3>(st, a, b, n, plan, 25, work, err, &sorted)) return 1;
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        if (r >= 2) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    uint32_t herr[4];
    CK(hipMemcpy(herr, err, 16, hipMemcpyDeviceToHost));
    const char *verdict = "not checked (ablation build)";
#if GFFX_SORT_ABL == 0
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
#endif
    const double med = ms[ms.size() / 2] * 1e3;
    printf("{\"records\": %llu, \"passes\": %d, \"sort_us\": %.1f, \"min_us\": %.1f, \"us_per_pass_incl_hist\": %.2f, \"GBps\": %.0f, \"err\": %u, \"result\": \"%s\", "
           "\"items\": %d, \"lookback\": %d, \"abl\": %d}\n",
           n, plan.n_passes, med, ms[0] * 1e3, med / plan.n_passes, 24.0 * n * plan.n_passes / (med * 1e-6) / 1e9, herr[0], verdict, kSortItems,
           kSortLookBack, GFFX_SORT_ABL);
    return 0;
}
