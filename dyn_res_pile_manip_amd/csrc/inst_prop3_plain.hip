// Explicit instantiations (k_prop_inst.h): this translation unit holds the device code of these kernels; csrc/drp_capi.hip launches them.
#define DRP_PROP_INSTANTIATE
#include "k_prop_inst.h"
#ifndef DRP_UNITY
KM_PROP3_LIST_TAPE(KM_INST_PROP3, false)
#endif
