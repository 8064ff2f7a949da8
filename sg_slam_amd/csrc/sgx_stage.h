// sgx_stage.h — per-thread, grow-only device staging slots for the host-pointer (one frame, synchronous) entry points.
// A Tracking thread calls sgx_match_project_frame / sgx_match_project_local / sgx_pose_optimization once per frame; allocating their ~40 device buffers
// with hipMalloc / hipFree on every call costs milliseconds — more than the kernels.  A slot keeps its allocation from call to call and only grows.
// The slots are never freed (they live as long as the thread that tracks; tearing HIP allocations down from a thread_local destructor at process exit
// races the runtime's own shutdown).
#pragma once
#include "sgx_rt.h"
#include "../../include/sgx.h"
#ifdef SGX_DEBUG_TAPS
#include "../../include/sgx_debug.h"      // test / tuning taps: compiled into tests/taps/libsgx_taps.so and the emulator only
#endif
#include <vector>

struct SgxStage {
    struct Slot { void *p = nullptr; size_t cap = 0; };
    std::vector<Slot> slots;
    int get(int k, size_t bytes, void **out)
    {
        if ((int)slots.size() <= k) slots.resize((size_t)k + 1);
        Slot &s = slots[(size_t)k];
        if (!s.p || s.cap < bytes) {
            if (s.p) (void)hipFree(s.p);
            s.p = nullptr; s.cap = 0;
            const size_t c = bytes + bytes / 2 + 256;
            if (hipMalloc(&s.p, c) != hipSuccess) { s.p = nullptr; return SGX_ERR_NOMEM; }
            s.cap = c;
        }
        *out = s.p;
        return SGX_OK;
    }
};
inline SgxStage &sgx_stage() { static thread_local SgxStage st; return st; }

// a staged buffer: slot `k` of this thread, optionally filled from host memory (asynchronously on the legacy stream)
struct SgxStaged {
    void *p = nullptr;
    int put(int k, const void *src, size_t n)
    {
        const int rc = sgx_stage().get(k, n ? n : 1, &p);
        if (rc != SGX_OK) return rc;
        if (src && n && hipMemcpyAsync(p, src, n, hipMemcpyHostToDevice, 0) != hipSuccess) return SGX_ERR_DEVICE;
        return SGX_OK;
    }
};
