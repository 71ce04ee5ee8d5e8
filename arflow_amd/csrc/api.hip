#include "common.hpp"

#include <atomic>
#include <cstdio>

extern "C" int arflow_abi_version(void) { return 10; }

// A HIP error that was already pending on the calling thread when an entry point was entered (left behind by
// the framework or by an earlier, unchecked call).  Kept, not dropped: first one is reported once on stderr.
static std::atomic<int> g_stale_code{0};
static std::atomic<bool> g_stale_reported{false};

void af_record_stale_error(int code) {
  int expected = 0;
  g_stale_code.compare_exchange_strong(expected, code);  // keep the FIRST one until the host takes it
  if (!g_stale_reported.exchange(true))
    std::fprintf(stderr,
                 "libarflow_hip: a HIP error was already pending when an arflow entry point was called: %s (%d). "
                 "It is NOT from this library's launch; read it with arflow_take_stale_error().\n",
                 hipGetErrorString((hipError_t)code), code);
}

extern "C" int arflow_take_stale_error(void) { return g_stale_code.exchange(0); }

// Profiling aid (tools/kbench.py): a one-wave no-op kernel whose NAME delimits the dispatches of consecutive C-ABI calls
// in a rocprofv3 kernel / counter trace, so that counters can be summed per CALL (an entry point may launch several kernels).
__global__ void af_marker_kernel(int tag) { (void)tag; }
extern "C" int arflow_profile_marker(int tag, arflow_stream_t stream) {
  hipLaunchKernelGGL(af_marker_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, tag);
  return af_launch_status();
}

extern "C" const char* arflow_strerror(int code) {
  switch (code) {
    case ARFLOW_OK: return "ok";
    case ARFLOW_ENULL: return "required pointer is NULL";
    case ARFLOW_ESHAPE: return "non-positive, inconsistent or too large dimension";
    case ARFLOW_EPARAM: return "unsupported mode or parameter value";
    default: break;
  }
  if (code <= ARFLOW_ELAUNCH_BASE) return hipGetErrorString((hipError_t)(ARFLOW_ELAUNCH_BASE - code));
  return "unknown arflow error code";
}

// Rows of every `sums` buffer: at least the workgroup count of any reduction kernel on a [B, *, H, W] problem -- the
// 8 x 32 pixel tiling rounded up to 8 (af_grid_for_tiles; the 16 x 64 tilings need fewer), and the one-thread-per-pixel
// kernels' ceil(W/256) x H x B grid.
int af_sums_rows(int B, int H, int W) {
  const long a = 8L * (((long)af_cdiv(W, 32) * af_cdiv(H, 8) * B + 7) / 8);
  const long b = (long)af_cdiv(W, 256) * H * B;
  return (int)(a > b ? a : b);
}
extern "C" int arflow_sums_rows(int B, int H, int W) { return (B > 0 && H > 0 && W > 0) ? af_sums_rows(B, H, W) : ARFLOW_ESHAPE; }
