// ring.hip -- time-step ring pipeline over the GPUs of one node (RCCL point-to-point over xGMI).
// Placeholder until the single-GPU path is validated on hardware.
#include "common.h"
#include "engine.h"
using namespace hps;
extern "C" int hps_ring_unique_id (char*) { set_error("ring: not built yet"); return HPS_ERR_UNSUPPORTED; }
extern "C" int hps_ring_init (void*, const char*, int, int) { set_error("ring: not built yet"); return HPS_ERR_UNSUPPORTED; }
extern "C" int hps_ring_run (void*, int) { set_error("ring: not built yet"); return HPS_ERR_UNSUPPORTED; }
