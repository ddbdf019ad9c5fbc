// oracle/ref_utils_exports.cpp - doorways into the reference's header-only src/utils.h (DCBlocker / MovingAverage,
// src/utils.h:76-99, 139-169), built into oracle/_ref/libref_core.so by `make -C oracle ref_core` when a genuine
// <boost/circular_buffer.hpp> exists.  TEST INFRASTRUCTURE ONLY; contains no DSP.
#include "utils.h"

extern "C" {
void *refc_dc_create(int delay) { return new DCBlocker<float>(delay); }
void refc_dc_destroy(void *h) { delete (DCBlocker<float> *)h; }
void refc_dc_process(void *h, float *arr, int n) { ((DCBlocker<float> *)h)->removeDC(arr, n); }
void refc_dc_reset(void *h) { ((DCBlocker<float> *)h)->reset(); }
}
