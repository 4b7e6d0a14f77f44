// beom_launch_tiled.h — the launches of the tiled sweeps (k_mont_visc, k_uv_fused) out of ONE tile geometry.
// beom_engine.hip includes beom_kernels.h twice, in namespace t8 (64 x 8 tiles: frames of many rounds of workgroups)
// and in namespace t4 (64 x 4 tiles, one row per thread: frames of one or two rounds of workgroups, where a
// workgroup's lifetime is the step time), and this file once for each: TNS = the namespace, TSUF = the suffix of the names.
#define TFN3(a, b) a##_##b
#define TFN2(a, b) TFN3(a, b)
#define TFN(name) TFN2(name, TSUF)

static bool TFN(raw_mont_visc)(beom_engine *E, bool leith) {
    const dim3 g = TNS::mont_visc_grid(E->d), b(BEOM_BLOCK);
    switch (E->d.nlay) {
#define CASE_NL(n) case n: if (leith) hipLaunchKernelGGL((TNS::k_mont_visc<n, true>), g, b, 0, E->stream, E->d); \
                           else hipLaunchKernelGGL((TNS::k_mont_visc<n, false>), g, b, 0, E->stream, E->d); return true;
        CASE_NL(1) CASE_NL(2) CASE_NL(3) CASE_NL(4) CASE_NL(5) CASE_NL(6) CASE_NL(7) CASE_NL(8)
#undef CASE_NL
        default: return false;
    }
}
static void TFN(raw_uv_fused)(beom_engine *E, bool first_x, bool prod, bool zv, double gene, double ramp, double ctim) {
    const dim3 g = TNS::uv_fused_grid(E->d), b(TNS::kUvBlock);
    DevView &d = E->d;
#define UV_GO(kern, fx, pr, z) hipLaunchKernelGGL((TNS::kern<fx, pr, z>), g, b, 0, E->stream, d, gene, ramp, ctim)
#define UV_PICK(fx) do { \
        if (d.stress_fold) { if (zv) UV_GO(k_uv_fused_sf, fx, true, true); else if (prod) UV_GO(k_uv_fused_sf, fx, true, false); else UV_GO(k_uv_fused_sf, fx, false, false); } \
        else { if (zv) UV_GO(k_uv_fused, fx, true, true); else if (prod) UV_GO(k_uv_fused, fx, true, false); else UV_GO(k_uv_fused, fx, false, false); } \
    } while (0)
    // (stress_fold: distribute_stress formed inside the sweep — its own instantiations, so that the unforced ones stay lean)
    if (first_x) UV_PICK(true); else UV_PICK(false);
#undef UV_PICK
#undef UV_GO
}
#undef TFN
#undef TFN2
#undef TFN3
