// Shared host-side helpers of libdtc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dtc_hip.h"

namespace dtc {

void set_error(const char* fmt, ...);
int check_launch(const char* what);

// Optional per-launch HIP-event timing (dtc_prof_enable).  `work` is the algorithmic work of
// the launch in the unit of its roofline (FLOP for MFMA-bound kernels, bytes for HBM-bound).
struct ProfScope {
    ProfScope(const char* name, double work, hipStream_t s, double bytes = 0.0);
    ~ProfScope();
    int slot;
    hipStream_t stream;
};

const char* prof_shape_name(const char* base, int M, int N, int K);

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace dtc

#define DTC_REQUIRE(cond, ...)              \
    do {                                    \
        if (!(cond)) {                      \
            dtc::set_error(__VA_ARGS__);    \
            return DTC_ERR_ARG;             \
        }                                   \
    } while (0)
