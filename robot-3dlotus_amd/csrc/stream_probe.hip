// lotus-hip: can two HIP streams make progress independently of each other?  The data-parallel step (parallel.py) keeps two
// RCCL communicators in flight at once — the gradient buckets on the communication stream, the SyncBatchNorm statistics on the
// training stream (genrobo3d/train/utils/distributed.py:196-205, train/train_simple_policy.py:116-117 reach the same state
// through DistributedDataParallel + SyncBatchNorm).  A collective kernel BLOCKS until its peers have started theirs, and the
// order in which a GPU starts the kernels of two streams is not the same on every rank.  That is harmless as long as a
// blocked kernel of one stream cannot hold back the other stream; it is a deadlock when it can:
//
//     rank A, one hardware queue:  [bucket 3] -> [statistics 17]      rank B, one hardware queue:  [statistics 17] -> [bucket 3]
//
// HIP maps streams onto GPU_MAX_HW_QUEUES (4) hardware queues per priority and which two share one is decided by creation
// order, i.e. by everything else the process did before.  This probe answers the question on the device instead of assuming:
// park a kernel on `blocked` that spins on a host flag, launch a second kernel on `other`, and see whether it finishes while
// the first one is still parked.  The parked kernel gives up by itself after a bounded time (wall clock), so the probe cannot
// hang a box.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <chrono>
#include <thread>

#include "../../include/lotus_hip.h"

void lotus_set_error(const char* fmt, ...);

namespace {
// s_memrealtime / wall_clock64 ticks at 100 MHz on gfx9
constexpr long long kTicksPerMs = 100000;

__global__ void park_kernel(volatile int* flag, long long max_ticks, int* gave_up) {
  if (threadIdx.x != 0) return;
  const long long t0 = wall_clock64();
  while (__atomic_load_n((const int*)flag, __ATOMIC_RELAXED) == 0) {
    if (wall_clock64() - t0 > max_ticks) {
      *gave_up = 1;
      return;
    }
    __builtin_amdgcn_s_sleep(64);
  }
}

__global__ void touch_kernel(int* out) {
  if (threadIdx.x == 0) *out = 1;
}
}  // namespace

extern "C" int lotus_stream_probe(void* blocked, void* other, int timeout_ms) {
  if (blocked == other || timeout_ms < 1 || timeout_ms > 2000) {
    lotus_set_error("lotus_stream_probe: two different streams and 1..2000 ms");
    return LOTUS_E_ARG;
  }
  int* host = nullptr;  // [0] release flag (host writes), [1] parked kernel gave up, [2] second kernel ran
  if (hipHostMalloc((void**)&host, 3 * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
    lotus_set_error("lotus_stream_probe: hipHostMalloc failed");
    return LOTUS_E_LAUNCH;
  }
  host[0] = host[1] = host[2] = 0;
  int* dev = nullptr;
  hipEvent_t done = nullptr;
  int rc = LOTUS_E_LAUNCH;
  if (hipHostGetDevicePointer((void**)&dev, host, 0) == hipSuccess && hipEventCreateWithFlags(&done, hipEventDisableTiming) == hipSuccess) {
    // the parked kernel leaves by itself well after the host stopped waiting
    park_kernel<<<1, 64, 0, (hipStream_t)blocked>>>(dev, (long long)(timeout_ms + 500) * kTicksPerMs, dev + 1);
    touch_kernel<<<1, 64, 0, (hipStream_t)other>>>(dev + 2);
    bool ok = hipGetLastError() == hipSuccess && hipEventRecord(done, (hipStream_t)other) == hipSuccess;
    int independent = 0;
    if (ok) {
      const auto t0 = std::chrono::steady_clock::now();
      for (;;) {
        hipError_t q = hipEventQuery(done);
        if (q == hipSuccess) {
          independent = 1;
          break;
        }
        if (q != hipErrorNotReady) {
          ok = false;
          break;
        }
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(timeout_ms)) break;
        std::this_thread::sleep_for(std::chrono::microseconds(50));
      }
    }
    __atomic_store_n(&host[0], 1, __ATOMIC_RELEASE);  // release the parked kernel, then drain both streams
    ok = hipStreamSynchronize((hipStream_t)blocked) == hipSuccess && ok;
    ok = hipStreamSynchronize((hipStream_t)other) == hipSuccess && ok;
    if (!ok)
      lotus_set_error("lotus_stream_probe: launch or query failed (%s)", hipGetErrorString(hipGetLastError()));
    else
      rc = independent;
  } else {
    lotus_set_error("lotus_stream_probe: setup failed");
  }
  if (done) (void)hipEventDestroy(done);
  (void)hipHostFree(host);
  return rc;
}
