// Every switch that changes WHICH KERNEL (or which split of it) a library call takes, in ONE place: read from the
// environment once per process, listed with its default, and reportable (mmdgan_tuning_describe) - so a test log or a
// bench line can say under which selection it ran instead of that being hidden process state.  The reference has no
// counterpart (one fixed TF graph, graph_func.py:851-854); the defaults below ARE the build's fixed graph, and
// tests/test_production_gpu.py + tests/golden/production_kernels.json pin the kernels they select.
// Anything that was an experiment's A/B lever and lost (tile forcing, split forcing, block caps, event flags, contiguous
// item runs, ...) is gone: its measurement is in DESIGN.md, not in the library's surface.
#pragma once
#include <stdlib.h>

namespace mmdgan {

struct Tuning {
    int force_direct;            // MMDGAN_FORCE_DIRECT=1        every conv on the generic direct kernels (debugging reference)
    int thin_valu;               // MMDGAN_THIN_VALU=1           thin first / last layers on the VALU kernels, not the MFMA ones
    int wino;                    // MMDGAN_WINO=0                no F(2x2,3x3) kernels (3x3 layers on the implicit GEMM)
    long wino_min_tiles;         // MMDGAN_WINO_MIN_TILES=n      F(2x2,3x3) from n tiles on (default -1: 512, or 128 with caller-transformed weights)
    long wino_ksplit_below;      // MMDGAN_WINO_KSPLIT_BELOW=n   its reduction split over workspace slabs for grids below n workgroups (385)
    int wino_wgrad;              // MMDGAN_WINO_WGRAD=0          3x3 weight gradients on the implicit GEMM
    int wino_wgrad_slab;         // MMDGAN_WINO_WGRAD_SLAB=0     ... on the atomics form of the Winograd-domain kernel, not the slab form
    int wino2;                   // MMDGAN_WINO2=0|1|2           F(2x2,2x2) for 4x4 stride-2: never | from 256 / 384 workgroups (default) | every eligible shape
    int wino2_ksplit;            // MMDGAN_WINO2_KSPLIT=0        no reduction split of its small launches
    long wino2_ksplit_below;     // MMDGAN_WINO2_KSPLIT_BELOW=n  split grids below n work items (384)
    int wino2_wgrad;             // MMDGAN_WINO2_WGRAD=0         4x4 stride-2 weight gradients on the implicit GEMM
    long wino2_wgrad_min_tiles;  // MMDGAN_WINO2_WGRAD_MIN_TILES=n   ... from n tiles on (256)
    int wgrad_cus;               // MMDGAN_WGRAD_CUS=n           workgroups (= CUs) the one-round weight-gradient kernels size their grid for (224)
    int gemm_skinny;             // MMDGAN_GEMM_SKINNY=0         D's head product on the tiled kernel, not the skinny-N MFMA one
    int mmd_d16;               // MMDGAN_MMD_D16=0             the pairwise loss with d = 16 on the run-time-d instantiation of its kernel
    int gemm_panel;              // MMDGAN_GEMM_PANEL=0          short-K dense products (G's first layer, the dense weight gradients) on the tiled kernel
    int wino43;                  // MMDGAN_WINO43=0|1|2          F(4x4,3x3) for 3x3 stride-1 (H, W multiples of 4): never | from wino43_min_tiles on (default) | every eligible shape
    long wino43_min_tiles;       // MMDGAN_WINO43_MIN_TILES=n    ... from n 4x4 tiles on (128)
    long wino43_ksplit_below;    // MMDGAN_WINO43_KSPLIT_BELOW=n its reduction split over workspace slabs for grids below n workgroups (256)
    int wino43_wgrad;            // MMDGAN_WINO43_WGRAD=0|1|2    F(4x4,3x3) weight gradients: never | from wino43_wgrad_min_tiles on (default) | every eligible shape
    long wino43_wgrad_min_tiles; // MMDGAN_WINO43_WGRAD_MIN_TILES=n  ... from n 4x4 tiles on (128)
    int wino43_wgrad_cus;        // MMDGAN_WINO43_WGRAD_CUS=n    workgroups that kernel sizes its one-round grid for (0: wgrad_cus)
};

inline const Tuning &tuning_defaults() {
    static const Tuning d = {0, 0, 1, -1, 385, 1, 1, 1, 1, 384, 1, 256, 224, 1, 1, 1, 1, 128, 256, 1, 128, 0};
    return d;
}

inline const Tuning &tuning() {
    static const Tuning t = [] {
        Tuning v = tuning_defaults();
        auto geti = [](const char *name, int dflt) { const char *e = getenv(name); return e && *e ? atoi(e) : dflt; };
        auto getl = [](const char *name, long dflt) { const char *e = getenv(name); return e && *e ? atol(e) : dflt; };
        v.force_direct = geti("MMDGAN_FORCE_DIRECT", v.force_direct) == 1;
        v.thin_valu = geti("MMDGAN_THIN_VALU", v.thin_valu) == 1;
        v.wino = geti("MMDGAN_WINO", v.wino) != 0;
        v.wino_min_tiles = getl("MMDGAN_WINO_MIN_TILES", v.wino_min_tiles);
        v.wino_ksplit_below = getl("MMDGAN_WINO_KSPLIT_BELOW", v.wino_ksplit_below);
        v.wino_wgrad = geti("MMDGAN_WINO_WGRAD", v.wino_wgrad) != 0;
        v.wino_wgrad_slab = geti("MMDGAN_WINO_WGRAD_SLAB", v.wino_wgrad_slab) != 0;
        v.wino2 = geti("MMDGAN_WINO2", v.wino2);
        v.wino2_ksplit = geti("MMDGAN_WINO2_KSPLIT", v.wino2_ksplit) != 0;
        v.wino2_ksplit_below = getl("MMDGAN_WINO2_KSPLIT_BELOW", v.wino2_ksplit_below);
        v.wino2_wgrad = geti("MMDGAN_WINO2_WGRAD", v.wino2_wgrad) != 0;
        v.wino2_wgrad_min_tiles = getl("MMDGAN_WINO2_WGRAD_MIN_TILES", v.wino2_wgrad_min_tiles);
        const int cus = geti("MMDGAN_WGRAD_CUS", v.wgrad_cus);
        v.wgrad_cus = cus > 0 ? cus : v.wgrad_cus;
        v.gemm_skinny = geti("MMDGAN_GEMM_SKINNY", v.gemm_skinny) != 0;
        v.gemm_panel = geti("MMDGAN_GEMM_PANEL", v.gemm_panel) != 0;
        v.mmd_d16 = geti("MMDGAN_MMD_D16", v.mmd_d16) != 0;
        v.wino43 = geti("MMDGAN_WINO43", v.wino43);
        v.wino43_min_tiles = getl("MMDGAN_WINO43_MIN_TILES", v.wino43_min_tiles);
        v.wino43_ksplit_below = getl("MMDGAN_WINO43_KSPLIT_BELOW", v.wino43_ksplit_below);
        v.wino43_wgrad = geti("MMDGAN_WINO43_WGRAD", v.wino43_wgrad);
        v.wino43_wgrad_min_tiles = getl("MMDGAN_WINO43_WGRAD_MIN_TILES", v.wino43_wgrad_min_tiles);
        v.wino43_wgrad_cus = geti("MMDGAN_WINO43_WGRAD_CUS", v.wino43_wgrad_cus);
        return v;
    }();
    return t;
}

}  // namespace mmdgan
