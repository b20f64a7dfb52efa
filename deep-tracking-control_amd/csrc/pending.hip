// Entry points declared in include/dtc_hip.h whose kernels are not written yet: they fail loudly.
#include "common.hpp"
#define NOT_YET(name) dtc::set_error(name ": not implemented in this build"); return DTC_ERR_ARG
extern "C" {
int64_t dtc_gru_workspace(int, int, int) { return 0; }
int dtc_gru_fwd(const float*, const float*, const float*, const float*, float*, float*, float*, void*, int, int, int, void*) { NOT_YET("dtc_gru_fwd"); }
int dtc_gru_bwd(const float*, const float*, const float*, const float*, const float*, const float*, float*, float*, float*, float*, void*, int, int, int, void*) { NOT_YET("dtc_gru_bwd"); }
}
