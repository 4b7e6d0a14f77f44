// beom_dense.h — fast kernels for dense frames (every BASELINE config).
// Each launch_dense_* returns false when it has no specialised kernel for the handle's
// configuration; the caller then uses the generic template kernels with NbDense.
#pragma once
#include "beom_dev.h"

static inline bool launch_dense_h(DevView &, hipStream_t, double, double, double) { return false; }
static inline bool launch_dense_mont(DevView &, hipStream_t) { return false; }
static inline bool launch_dense_visc(DevView &, hipStream_t) { return false; }
template <bool XDIR>
static inline bool launch_dense_uv(DevView &, hipStream_t, double, double, double) { return false; }
