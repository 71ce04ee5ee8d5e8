"""ctypes binding of libarflow_hip.so (the C ABI declared in include/arflow_hip.h).

There is NO fallback: if the shared library is missing or a symbol is absent, importing the ops
raises.  Build with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C arflow_amd/csrc``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libarflow_hip.so')

c_fp = ctypes.c_void_p  # device pointers travel as integers
c_i = ctypes.c_int
c_l = ctypes.c_long
c_f = ctypes.c_float

# name -> argtypes, exactly the prototypes of include/arflow_hip.h
PROTOTYPES = {
    'arflow_abi_version': [],
    'arflow_take_stale_error': [],
    'arflow_profile_marker': [c_i, c_fp],
    'arflow_sums_rows': [c_i, c_i, c_i],
    'arflow_corr_fwd': [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_f, c_fp],
    'arflow_corr_sign_planes': [c_i, c_i, c_i],
    'arflow_corr_strided_supported': [c_i, c_i, c_i],
    'arflow_corr_fwd_strided': [c_fp, c_fp, c_fp, c_l, c_fp, c_i, c_i, c_i, c_i, c_i, c_f, c_fp],
    'arflow_corr_bwd_strided': [c_fp, c_l, c_fp, c_l, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_f, c_fp],
    'arflow_corr_bwd': [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_f, c_fp],
    'arflow_corr_fwd_bf16': [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_f, c_fp],
    'arflow_corr_bwd_bf16': [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_f, c_fp],
    'arflow_warp_fwd_bf16': [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_l, c_i, c_i, c_i, c_fp],
    'arflow_warp_bwd_bf16': [c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_l, c_i, c_i, c_i, c_fp],
    'arflow_warp_nearest_fwd': [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_l, c_i, c_i, c_i, c_fp],
    'arflow_warp_nearest_bwd': [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_l, c_i, c_i, c_i, c_fp],
    'arflow_warp_bicubic_fwd': [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_l, c_i, c_i, c_i, c_fp],
    'arflow_warp_bicubic_bwd': [c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_l, c_i, c_i, c_i, c_fp],
    'arflow_ssim_fwd': [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp],
    'arflow_ssim_bwd': [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp],
    'arflow_corr_general_out_size': [c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp],
    'arflow_corr_general_fwd': [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp],
    'arflow_corr_general_bwd': [c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp],
    'arflow_featnorm_fwd': [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_l, c_i, c_fp],
    'arflow_featnorm_bwd': [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_l, c_i, c_fp],
    'arflow_level_supported': [c_i, c_i, c_i],
    'arflow_level_bwd_ws_bytes': [c_i, c_i, c_i, c_i],
    'arflow_level_fwd': [c_fp, c_fp, c_fp, c_l, c_i, c_i, c_fp, c_fp, c_l, c_fp, c_i, c_fp, c_l, c_fp, c_l, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_i, c_i, c_fp],
    'arflow_level_bwd': [c_fp, c_l, c_fp, c_fp, c_l, c_fp, c_l, c_fp, c_fp, c_fp, c_fp, c_l, c_fp, c_l, c_fp, c_fp, c_i, c_fp, c_fp, c_fp, c_i, c_i, c_fp, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_i, c_i, c_fp],
    'arflow_level_fwd_m': [c_fp, c_fp, c_fp, c_l, c_i, c_i, c_fp, c_fp, c_l, c_fp, c_i, c_fp, c_l, c_fp, c_l, c_fp, c_fp, c_fp, c_fp, c_i, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_i, c_i, c_fp],
    'arflow_up2_bwd': [c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp],
    'arflow_bias_act_mom_rows': [c_i, c_l],
    'arflow_bias_act_fwd_mom': [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_l, c_f, c_fp],
    'arflow_level_acc_rows': [c_i, c_i, c_i, c_i, c_i],
    'arflow_level_moments': [c_fp, c_fp, c_fp, c_i, c_l, c_fp],
    'arflow_level_warp_fwd': [c_fp, c_fp, c_fp, c_l, c_i, c_i, c_fp, c_fp, c_l, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp],
    'arflow_level_corr_fwd': [c_fp, c_fp, c_fp, c_i, c_i, c_fp, c_l, c_fp, c_l, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_f, c_fp],
    'arflow_level_corr_bwd': [c_fp, c_l, c_fp, c_fp, c_l, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_f, c_fp],
    'arflow_bias_act_fwd': [c_fp, c_fp, c_fp, c_i, c_i, c_l, c_f, c_fp],
    'arflow_bias_act_bwd': [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_l, c_f, c_fp],
    'arflow_warp_fwd': [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_l, c_i, c_i, c_i, c_fp],
    'arflow_warp_bwd': [c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_l, c_i, c_i, c_i, c_fp],
    'arflow_splat_map': [c_fp, c_fp, c_i, c_i, c_i, c_l, c_i, c_fp],
    'arflow_coord_mask': [c_fp, c_fp, c_i, c_i, c_i, c_l, c_i, c_fp],
    'arflow_occ_bidir': [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_l, c_l, c_f, c_f, c_fp],
    'arflow_census_fwd': [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp],
    'arflow_census_bwd': [c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp],
    'arflow_census_warp_supported': [c_i, c_i],
    'arflow_census_warp_fwd': [c_fp, c_fp, c_fp, c_l, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp],
    'arflow_census_warp_bwd': [c_fp, c_fp, c_fp, c_l, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp],
    'arflow_census_warp_pair_fwd': [c_fp, c_fp, c_l, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp],
    'arflow_census_warp_pair_bwd': [c_fp, c_fp, c_l, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp],
    'arflow_uflow_pair_bwd': [c_fp, c_fp, c_l, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_l, c_fp, c_fp, c_fp, c_i, c_i, c_f, c_f, c_i, c_i, c_i, c_fp],
    'arflow_down4_gray_z': [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp],
    'arflow_splat_smooth_fwd': [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_l, c_f, c_f, c_i, c_i, c_i, c_i, c_fp],
    'arflow_down4_gray': [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp],
    'arflow_photo_fwd': [c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp],
    'arflow_photo_bwd': [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp],
    'arflow_smooth_fwd': [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_l, c_f, c_f, c_i, c_i, c_i, c_fp],
    'arflow_smooth_bwd': [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_l, c_f, c_f, c_i, c_i, c_i, c_fp],
    'arflow_down4': [c_fp, c_fp, c_i, c_i, c_i, c_fp],
    'arflow_up4_clamp_mul': [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp],
}

ABI_VERSION = 10
_lib = None


class ArflowHipError(RuntimeError):
    pass


def load():
    """Load the library once; raise loudly if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ArflowHipError(
            'libarflow_hip.so not found at %s: the HIP extension is not built and there is no CPU '
            'fallback.  Run `make -C arflow_amd/csrc` (or __graft_entry__.build()).' % LIB_PATH)
    # torch first: the library must bind to the HIP runtime PyTorch has loaded (the one that owns the device
    # memory and streams it is handed).  Loaded ahead of torch it pulls in the system runtime instead and the
    # process ends up with two: launches then fail with hipErrorNoDevice.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = c_l if name == 'arflow_level_bwd_ws_bytes' else c_i
    lib.arflow_strerror.argtypes = [c_i]
    lib.arflow_strerror.restype = ctypes.c_char_p
    if lib.arflow_abi_version() != ABI_VERSION:
        raise ArflowHipError('libarflow_hip.so ABI %d != expected %d; rebuild' % (lib.arflow_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().arflow_strerror(code)
        raise ArflowHipError('%s failed: %s (code %d)' % (what, msg.decode() if msg else '?', code))


def poll_stale_error(lib, what):
    """A HIP error that was ALREADY pending when an entry point was entered (left by the framework or by an unchecked
    earlier call) is moved into a process-wide slot by the library (csrc/common.hpp af_clear_stale_error) instead of being
    blamed on -- or silently cleared by -- our launch.  Read and clear that slot after every call and surface it."""
    code = lib.arflow_take_stale_error()
    if code:
        import warnings
        warnings.warn('a HIP error (hipError_t %d) was pending when %s was called: it comes from an earlier call of '
                      'another library on this thread, not from arflow_amd' % (code, what), RuntimeWarning, stacklevel=3)
