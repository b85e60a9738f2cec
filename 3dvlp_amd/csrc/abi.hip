// ABI bookkeeping entry points of libvlp3d_hip.so (include/vlp3d.h).
#include "common.h"

extern "C" int vlp3d_abi_version(void) { return 3; }
extern "C" int vlp3d_fp_contract(void) { return VLP3D_CONTRACT; }
