// Tuning switches: every PASN_* environment variable the library honours, read ONCE.
//
// The kernels' routing (which kernel a layer takes, with what geometry) depends on these.  Reading the process environment on the launch
// path (66 distinct names, 13 per pointwise-conv call in round 3) made the benchmarked path a function of whatever happened to be exported,
// re-evaluated at every call.  Now: the first tune() call snapshots the PASN_* variables of the environment that are in the REGISTRY
// (tuning.hip: name, class, one-line meaning); later changes of the environment are not seen until pasn_tuning_reload() (tests, A/B
// tools).  An unregistered PASN_* variable is ignored by the library and reported by pasn_tuning_report() so a typo cannot pass for a
// measurement.  Two classes:
//   route  -- documented switches of the product build (turn a kernel / fusion off, widen or narrow a route); DESIGN.md lists them.
//   dev    -- geometry overrides and timing ablations (results may be WRONG under *_ABL): compiled out unless the library is built with
//             -DPASN_TUNING (tune_dev() is then a constant nullptr and the code behind it folds away).  The test suite's geometry sweeps
//             (forced T chunks / tile shapes) are `geom` entries: kept in the product build because parity tests drive them.
#pragma once

namespace pasn {

// value of a registered switch in the current snapshot, or nullptr (unset).  Thread-safe; O(1) when nothing is set.
const char* tune(const char* name);

#ifdef PASN_TUNING
inline const char* tune_dev(const char* name) { return tune(name); }
#else
inline const char* tune_dev(const char*) { return nullptr; }
#endif

inline bool tune_is(const char* name, char c) {  // set and first character == c
    const char* e = tune(name);
    return e && e[0] == c;
}

}  // namespace pasn
