#include "common.hpp"

extern "C" int arflow_abi_version(void) { return 6; }

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
