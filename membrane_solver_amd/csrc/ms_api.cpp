// Host side of libmembrane_hip.so: context management, HBM residency, the
// device-resident minimizer step, and the extern "C" ABI of
// include/membrane_hip.h.  Arithmetic lives in ms_kernels.hip.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <ctime>
#include <map>
#include <string>
#include <new>

#include <dlfcn.h>

#include "ms_internal.h"

static void shard_comm_destroy(void* comm);  // (RCCL binding further down)

using namespace ms;

namespace {
std::string g_last_error;  // for failures that have no context yet
}

#include "ms_api_ctx.inc"  // the context: everything a ms_ctx holds
#include "ms_api_phases.inc"  // pass launchers, folds, mailboxes, the energy / gradient / direction phases
#include "ms_api_context.inc"  // create / destroy, setters and getters, curvature fields
#include "ms_api_tilt.inc"  // tilt-module evaluation, the relaxations (host-driven, device program, fused), leaflet fields
#include "ms_api_step.inc"  // energy / gradient entry points, line-search rounds, ms_step, volume projection, the resident step, ms_minimize
#include "ms_api_shard.inc"  // phase API, boundary exchange, the sharded drivers (RCCL, peer-to-peer), ms_shard_step
#include "ms_api_misc.inc"  // state rebinding, statistics, profiling, the tiling planner, the kernel-provider seam
