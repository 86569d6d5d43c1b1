"""ctypes binding of libaphantasia_hip.so (include/aphantasia_hip.h).

The product path has exactly one implementation: the HIP library built by
aphantasia_amd/_build.py for gfx950.  If it is missing or fails to load this module raises --
there is no CPU or PyTorch fallback.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_size_t, c_void_p

# torch MUST be imported before the library is dlopen'ed: PyTorch-ROCm bundles its own libamdhip64.so
# (same SONAME as /opt/rocm's).  Loaded first, it satisfies this library's DT_NEEDED entry and the process
# has ONE HIP runtime; loaded second, the process ends up with two runtimes and the later one to initialise
# reports "no ROCm-capable device is detected".
import torch  # noqa: F401  (load order matters)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libaphantasia_hip.so')

APH_OUT_NCHW_RAW, APH_OUT_NCHW_NORM, APH_OUT_PATCH_F16, APH_GRAD_PATCH_F16, APH_OUT_PATCH_F16_HILO = 0, 1, 2, 3, 4
APH_AUG_STRIDE = 16
SIM_TYPES = {'cossim': 0, 'cos': 0, None: 0, 'mix': 1, 'ang': 2, 'dot': 3}


class SampleGeom(ctypes.Structure):
    _fields_ = [(n, c_int) for n in ('H', 'W', 'Hp', 'Wp', 'py0', 'px0', 'S', 'size', 'patch')]


_PROTOTYPES = {
    'aph_version': (c_int, []),
    'aph_last_error': (c_char_p, []),
    'aph_synth_plan_create': (c_int, [c_int, c_int, c_int, POINTER(c_void_p)]),
    'aph_synth_plan_destroy': (c_int, [c_void_p]),
    'aph_synth_fft_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, POINTER(c_float), c_int, c_void_p, c_void_p, c_void_p]),
    'aph_synth_fft_bwd': (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_float, POINTER(c_float), c_int, c_void_p, c_void_p]),
    'aph_irfft2': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    'aph_rfft2': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    'aph_synth_spatial_fwd': (c_int, [c_void_p, c_void_p, c_float, c_float, POINTER(c_float), c_int, c_void_p, c_void_p]),
    'aph_synth_spatial_bwd': (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_float, c_float, POINTER(c_float), c_int, c_void_p, c_void_p]),
    'aph_synth_stats': (c_int, [c_void_p, c_void_p, c_void_p]),
    'aph_synth_set_stats': (c_int, [c_void_p, c_void_p, c_void_p]),
    'aph_rgb_priors_ws_bytes': (c_size_t, []),
    'aph_rgb_sharp': (c_int, [c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    'aph_rgb_priors': (c_int, [c_void_p, c_int, c_int, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    'aph_idwt_level_fwd': (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p]),
    'aph_idwt_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    'aph_idwt_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'aph_idwt_level_bwd': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_float, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'aph_sample_ws_bytes': (c_size_t, [POINTER(SampleGeom), c_int]),
    'aph_sample_fwd': (c_int, [POINTER(SampleGeom), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    'aph_sample_bwd': (c_int, [POINTER(SampleGeom), c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    'aph_triangle_blur': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p]),
    'aph_resize_bicubic': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    'aph_flip_w': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'aph_grid_warp': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    'aph_attn_test': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'aph_frame_affine': (c_int, [c_void_p, c_int, c_int, c_int, POINTER(c_float), c_void_p, c_void_p]),
    'aph_patchify_f16': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'aph_patchify_f16_hilo': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'aph_unpatchify_f32': (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    'aph_vit_create': (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_void_p)]),
    'aph_vit_destroy': (c_int, [c_void_p]),
    'aph_vit_workspace_bytes': (c_size_t, [c_void_p]),
    'aph_vit_set_weight': (c_int, [c_void_p, c_char_p, c_void_p, c_size_t]),
    'aph_vit_forward': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    'aph_vit_forward_hilo': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    'aph_vit_enable_hilo': (c_int, [c_void_p]),
    'aph_vit_backward': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_float, c_void_p]),
    'aph_vit_backward_h': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_float, c_void_p]),
    'aph_vit_profile': (c_int, [c_void_p, c_int]),
    'aph_vit_profile_read': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    'aph_gemm_set_mfma32': (c_int, [c_int]),
    'aph_vit_set_fuse_ln': (c_int, [c_int]),
    'aph_crop_adjoint_set_gather': (c_int, [c_int]),
    'aph_crop_adjoint_set_shape': (c_int, [c_int, c_int, c_int, c_int, c_int]),
    'aph_vit_set_grad_stream_f16': (c_int, [c_int]),
    'aph_gemm_set_ws_min_tiles': (c_int, [c_int]),
    'aph_gemm_set_ws_pgroup': (c_int, [c_int]),
    'aph_gemm_set_rs': (c_int, [c_int]),
    'aph_mfma_rate': (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'aph_gemm_rs_probe': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    'aph_gemm_ws_probe': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    'aph_gemm_f16': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'aph_gemm_f16_ld': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    'aph_sim_loss': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, POINTER(c_float), c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    'aph_axpy_f32': (c_int, [c_void_p, c_void_p, c_float, c_size_t, c_void_p]),
    'aph_rgb_to_u8': (c_int, [c_void_p, c_int, c_int, c_float, c_void_p, c_void_p]),
    'aph_linear_head': (c_int, [c_void_p, c_int, c_int, c_void_p, c_float, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    'aph_comm_unique_id': (c_int, [c_void_p]),
    'aph_comm_init': (c_int, [c_int, c_int, c_void_p, POINTER(c_void_p)]),
    'aph_allreduce_f32': (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    'aph_comm_ranks': (c_int, [c_void_p, POINTER(c_int)]),
    'aph_comm_destroy': (c_int, [c_void_p]),
    'aph_adam_step': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_size_t, c_void_p]),
    'aph_adam_step_guarded': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_size_t, c_void_p, c_void_p]),
}

EXPORTS = tuple(_PROTOTYPES)

# hooks of a -DAPH_EXPERIMENTS build (include/aphantasia_hip_experiments.h): bound when the loaded library has them
_EXPERIMENT_PROTOTYPES = {
    'aph_vit_set_fused_max_rows': (c_int, [c_int]),
    'aph_vit_set_fused_attn': (c_int, [c_int]),
    'aph_gemm_pack_frag': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'aph_attn_set_bwd_one': (c_int, [c_int]),
    'aph_attn_set_ablate': (c_int, [c_int]),
}


class Library:
    """A loaded C-ABI library with checked calls: `lib.call('aph_x', ...)` raises RuntimeError with
    aph_last_error() on a negative return code."""

    def __init__(self, path):
        if not os.path.isfile(path):
            raise RuntimeError(
                "aphantasia_amd: HIP library not found at %s -- build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
                "There is no CPU fallback." % path)
        self.path = path
        self.cdll = ctypes.CDLL(path)
        for name, (res, args) in _PROTOTYPES.items():
            fn = getattr(self.cdll, name)     # AttributeError if a declared symbol is missing
            fn.restype = res
            fn.argtypes = args
        self.experiments = all(hasattr(self.cdll, n) for n in _EXPERIMENT_PROTOTYPES)
        if self.experiments:
            for name, (res, args) in _EXPERIMENT_PROTOTYPES.items():
                fn = getattr(self.cdll, name)
                fn.restype = res
                fn.argtypes = args

    def last_error(self):
        return (self.cdll.aph_last_error() or b'').decode(errors='replace')

    def call(self, name, *args):
        rc = getattr(self.cdll, name)(*args)
        if rc < 0:
            raise RuntimeError('%s failed (%d): %s' % (name, rc, self.last_error()))
        return rc


_lib = None


def lib():
    """The product library (loaded once)."""
    global _lib
    if _lib is None:
        import torch
        if torch.cuda.is_available():
            # one HIP runtime per process: torch's is already loaded; let it create the primary context and
            # the caching allocator's device state before the library makes its first hipMalloc
            torch.cuda.init()
            torch.empty(1, device='cuda')
        _lib = Library(LIB_PATH)
    return _lib


def ptr(t):
    """tensor -> void* (None -> NULL)"""
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def floats(seq):
    """python floats -> float* host array (or NULL)"""
    if seq is None:
        return None
    arr = (c_float * len(seq))(*[float(v) for v in seq])
    return arr
