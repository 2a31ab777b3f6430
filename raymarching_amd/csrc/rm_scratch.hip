// rm_scratch.hip -- static device memory for callers that cannot allocate (code under HIP graph capture,
// tight serving loops, harnesses that forbid hipMalloc).  Every GPU the library runs on holds its own copy
// of the array, at its own address, until the process ends.
#include "rm_scratch.h"

#include <mutex>

static_assert(RM_SCRATCH_QUEUE_BYTES % 256 == 0 && RM_SCRATCH_SLOT_BYTES % 256 == 0, "scratch sections stay 256-byte aligned");

// In host code this name is only a handle for the runtime's symbol table, never a device address: the
// address of a GPU's copy comes from hipGetSymbolAddress below and is what every kernel argument, memset
// and copy uses.
static __device__ __align__(256) unsigned char scratch[RM_SCRATCH_BYTES];

namespace {
std::mutex g_mu;
void *g_ptr[RM_SCRATCH_MAX_DEVICES];      // per device: where that GPU's copy lives
}  // namespace

hipError_t rm_scratch_address(int device, void **ptr) {
    if (!ptr || device < 0 || device >= RM_SCRATCH_MAX_DEVICES) return hipErrorInvalidValue;
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_ptr[device]) {
        int current = -1;
        hipError_t e = hipGetDevice(&current);
        if (e != hipSuccess) return e;
        if (current != device) return hipErrorInvalidDevice;      // the symbol resolves on the current device
        void *p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        g_ptr[device] = p;
    }
    *ptr = g_ptr[device];
    return hipSuccess;
}
