// windows_launch.hpp -- the window kernels' launch dispatch: run-time {mode, seqid records in LDS, block width, offsets, positions,
// form} -> template arguments of k_join_pairs / k_join_roots (join_pairs_kernels.hpp), per KIND of launch (kLaunchPlain /
// kLaunchGroup / kLaunchTickets).  The 64 + 16 instantiations of ONE kind are one translation unit (engine_windows_plain.hip,
// _group.hip, _tickets.hip define GFFX_WINDOWS_LAUNCH_KIND and include this file: `make -j` compiles them side by side -- one unit
// with all 240 took five minutes); engine_windows.hip calls launch_windows_kind<KIND>.
#pragma once
#include "engine_private.hpp"
#include "join_pairs_kernels.hpp"

namespace gffx {

struct WindowsLaunch {
    int device;
    hipStream_t stream;
    uint32_t grid, threads, lds;
    bool roots, offs, pos, wide, ml;
    int mode;
    const PairArgs *a;
};
// Beyond the default 64 KB of dynamic LDS a kernel has to opt in, per function and per device (engine_windows.hip)
int lds_opt_in(const void *func, int device, uint32_t lds, uint32_t max_lds);
template <int KIND>
int launch_windows_kind(const WindowsLaunch &L);

#if defined(GFFX_WINDOWS_LAUNCH_KIND) || defined(GFFX_WINDOWS_LAUNCH_ALL)  // (the translation unit that instantiates this kind; ALL: engine.hip, the tools' one-unit build)

template <int MODE, bool ML, int T, bool OFFS, bool POS, bool WIDE, int KIND>
static inline int launch_pairs_t(const WindowsLaunch &L) {
    const int rc = lds_opt_in(reinterpret_cast<const void *>(&k_join_pairs<MODE, ML, T, OFFS, POS, WIDE, KIND>), L.device, L.lds,
                              T == 1024 ? 2 * kWinMaxLds : kWinMaxLds);
    if (rc) return rc;
    hipLaunchKernelGGL((k_join_pairs<MODE, ML, T, OFFS, POS, WIDE, KIND>), dim3(L.grid), dim3(T), L.lds, L.stream, *L.a);
    return GFFX_OK;
}
template <int MODE, bool ML, int T, bool WIDE, int KIND>
static inline int launch_roots_t(const WindowsLaunch &L) {
    const int rc = lds_opt_in(reinterpret_cast<const void *>(&k_join_roots<MODE, ML, T, WIDE, KIND>), L.device, L.lds, T == 1024 ? 2 * kWinMaxLds : kWinMaxLds);
    if (rc) return rc;
    hipLaunchKernelGGL((k_join_roots<MODE, ML, T, WIDE, KIND>), dim3(L.grid), dim3(T), L.lds, L.stream, *L.a);
    return GFFX_OK;
}
template <int MODE, bool ML, int KIND>
static inline int launch_mode(const WindowsLaunch &L) {
    if (L.roots) {
        if (L.wide) return L.threads == 1024 ? launch_roots_t<MODE, ML, 1024, true, KIND>(L) : launch_roots_t<MODE, ML, 512, true, KIND>(L);
        return L.threads == 1024 ? launch_roots_t<MODE, ML, 1024, false, KIND>(L) : launch_roots_t<MODE, ML, 512, false, KIND>(L);
    }
#define GFFX_P(T, O, P)                                                     \
    if (L.threads == T && L.offs == O && L.pos == P) {                      \
        if (L.wide) return launch_pairs_t<MODE, ML, T, O, P, true, KIND>(L); \
        return launch_pairs_t<MODE, ML, T, O, P, false, KIND>(L);           \
    }
    GFFX_P(1024, false, false) GFFX_P(1024, true, false) GFFX_P(1024, false, true) GFFX_P(1024, true, true)
    GFFX_P(512, false, false) GFFX_P(512, true, false) GFFX_P(512, false, true) GFFX_P(512, true, true)
#undef GFFX_P
    return fail(GFFX_E_INVALID, "windows launch: no instantiation for %u threads", L.threads);
}
template <int KIND>
int launch_windows_kind(const WindowsLaunch &L) {
#define GFFX_CASE(M, ML_) \
    if (L.mode == M && L.ml == ML_) return launch_mode<M, ML_, KIND>(L);
    GFFX_CASE(0, true) GFFX_CASE(0, false) GFFX_CASE(1, true) GFFX_CASE(1, false) GFFX_CASE(2, true) GFFX_CASE(2, false)
#undef GFFX_CASE
    return fail(GFFX_E_INVALID, "windows launch: bad mode %d", L.mode);
}
#ifdef GFFX_WINDOWS_LAUNCH_ALL
template int launch_windows_kind<kLaunchPlain>(const WindowsLaunch &L);
template int launch_windows_kind<kLaunchGroup>(const WindowsLaunch &L);
template int launch_windows_kind<kLaunchTickets>(const WindowsLaunch &L);
#else
template int launch_windows_kind<GFFX_WINDOWS_LAUNCH_KIND>(const WindowsLaunch &L);
#endif

#endif

}  // namespace gffx
