"""ctypes binding of libbp_gpu.so (C ABI: include/bp_gpu.h).

The library is the product; there is no Python or CPU fallback.  Importing this
module without the built shared object raises immediately.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbp_gpu.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `make -C bulletproofs_amd/csrc` "
        "(or __graft_entry__.build()); bulletproofs_amd has no CPU fallback"
    )



def _load_hip_runtime():
    """libbp_gpu.so leaves the HIP runtime unbound; exactly one copy may live in a
    process.  PyTorch wheels bundle their own libamdhip64, so when torch is
    importable that copy is used (torch tensors' device pointers and
    torch.distributed then share the library's runtime); otherwise the system one.
    BP_HIP_RUNTIME=/path/to/libamdhip64.so overrides."""
    path = os.environ.get("BP_HIP_RUNTIME")
    if not path:
        try:
            import torch  # noqa: F401

            cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
            if os.path.exists(cand):
                path = cand
        except ImportError:
            pass
    if not path:
        for cand in ("/opt/rocm/lib/libamdhip64.so", "libamdhip64.so"):
            if cand.startswith("/") and not os.path.exists(cand):
                continue
            path = cand
            break
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


hip_runtime = _load_hip_runtime()
lib = C.CDLL(LIB_PATH)

u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)
vp = C.c_void_p
sz = C.c_size_t
ci = C.c_int

BP_OK, BP_E_ARG, BP_E_CURVE, BP_E_HIP, BP_E_OOM, BP_E_STATE = 0, -1, -2, -3, -4, -5
ERR_NAMES = {0: "BP_OK", -1: "BP_E_ARG", -2: "BP_E_CURVE", -3: "BP_E_HIP", -4: "BP_E_OOM", -5: "BP_E_STATE"}

# name -> (restype, argtypes).  Pointers are passed as void* so numpy addresses,
# torch data_ptr() values and None all work.
SIGNATURES = {
    "bp_init": (ci, [ci, C.POINTER(vp)]),
    "bp_shutdown": (None, [vp]),
    "bp_last_error": (C.c_char_p, [vp]),
    "bp_sync": (ci, [vp]),
    "bp_set_msm_window": (ci, [vp, ci]),
    "bp_msm": (ci, [vp, ci, vp, vp, vp, vp, sz, vp]),
    "bp_msm_dev": (ci, [vp, ci, vp, vp, vp, vp, sz, vp]),
    "bp_msm2_dev": (ci, [vp, ci, vp, vp, vp, vp, sz, vp, vp, vp, vp, sz, vp]),
    "bp_bases_register": (ci, [vp, ci, vp, vp, vp, sz, C.POINTER(vp)]),
    "bp_bases_free": (None, [vp]),
    "bp_bases_len": (sz, [vp]),
    "bp_msm_resident": (ci, [vp, vp, sz, sz, vp, vp]),
    "bp_msm_resident_pair": (ci, [vp, vp, sz, sz, vp, vp, vp, vp]),
    "bp_msm_resident_plus": (ci, [vp, vp, sz, sz, vp, vp, vp, vp, vp, sz, vp]),
    "bp_msm_windows_dev": (ci, [vp, ci, vp, vp, vp, vp, sz, vp, C.POINTER(ci), C.POINTER(ci)]),
    "bp_msm_combine_windows_dev": (ci, [vp, ci, vp, ci, ci, ci, vp]),
    "bp_batch_begin": (ci, [vp, vp, sz, C.POINTER(vp)]),
    "bp_batch_axpy": (ci, [vp, sz, vp, sz, vp]),
    "bp_batch_axpy_dev": (ci, [vp, sz, vp, sz, vp]),
    "bp_batch_axpy_s_vector": (ci, [vp, vp, sz, vp, vp, sz]),
    "bp_batch_msm": (ci, [vp, vp, vp, vp, vp, sz, vp]),
    "bp_batch_free": (None, [vp]),
    "bp_msm_aos": (ci, [vp, ci, vp, vp, sz, vp]),
    "bp_bases_register_aos": (ci, [vp, ci, vp, sz, C.POINTER(vp)]),
    "bp_msm_shard_window_bits": (ci, [sz, ci]),
    "bp_msm_shard_windows_dev": (ci, [vp, ci, vp, vp, vp, vp, sz, ci, vp, C.POINTER(ci)]),
    "bp_msm_shard_windows_resident": (ci, [vp, vp, sz, sz, vp, ci, vp, C.POINTER(ci)]),
    "bp_inner_product": (ci, [vp, ci, vp, vp, sz, vp]),
    "bp_sum": (ci, [vp, ci, vp, sz, vp]),
    "bp_fold_scalars": (ci, [vp, ci, vp, vp, sz, vp, vp, vp]),
    "bp_vec_op": (ci, [vp, ci, ci, vp, vp, vp, sz, vp]),
    "bp_s_vector": (ci, [vp, ci, vp, vp, sz, vp]),
    "bp_fold_s_vectors": (ci, [vp, ci, vp, vp, vp, sz, sz, vp]),
    "bp_range_agg_poly": (ci, [vp, ci, vp, sz, C.c_uint, vp, vp, vp, vp, vp, vp]),
    "bp_range_agg_eval": (ci, [vp, ci, vp, sz, vp, vp, vp]),
    "bp_range_agg_verify_scalars": (ci, [vp, ci, vp, vp, sz, vp, vp, vp, vp, sz, C.c_uint, vp, vp]),
    "bp_batch_axpy_ipa": (ci, [vp, vp, sz, sz, sz, vp, vp, sz, vp, vp, vp]),
    "bp_batch_axpy_range_agg": (ci, [vp, vp, vp, vp, sz, vp, vp, vp, vp, sz, C.c_uint, vp]),
    "bp_ipa_verify_msm": (ci, [vp, vp, sz, sz, sz, vp, vp, sz, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "bp_range_agg_verify_msm": (ci, [vp, vp, vp, vp, sz, vp, vp, vp, vp, sz, C.c_uint, vp, vp, vp, vp, vp, sz, vp]),
    "bp_powers": (ci, [vp, ci, vp, sz, vp]),
    "bp_batch_inverse": (ci, [vp, ci, ci, vp, sz]),
    "bp_fold_generators": (ci, [vp, ci, vp, vp, vp, vp, vp, vp, sz, vp, vp, vp, vp, vp]),
    "bp_scale_points": (ci, [vp, ci, vp, vp, vp, vp, sz, vp, vp, vp]),
    "bp_normalize_batch": (ci, [vp, ci, vp, sz, vp, vp, vp]),
    "bp_inner_product_dev": (ci, [vp, ci, vp, vp, sz, vp]),
    "bp_inner_product_dev_noalloc": (ci, [vp, ci, vp, vp, sz, vp]),
    "bp_fold_scalars_dev": (ci, [vp, ci, vp, vp, sz, vp, vp, vp]),
    "bp_fold_generators_dev": (ci, [vp, ci, vp, vp, vp, vp, vp, vp, sz, vp, vp, vp, vp, vp]),
    "bp_s_vector_dev": (ci, [vp, ci, vp, vp, sz, vp]),
    "bp_ipa_begin": (ci, [vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, sz, C.POINTER(vp)]),
    "bp_ipa_round_lr": (ci, [vp, vp, vp, vp, vp]),
    "bp_ipa_round_fold": (ci, [vp, vp, vp]),
    "bp_ipa_len": (sz, [vp]),
    "bp_bases_precompute": (ci, [vp]),
    "bp_msm_resident_s_vector": (ci, [vp, vp, sz, vp, vp, sz, vp]),
    "bp_bases_register_compressed": (ci, [vp, ci, vp, sz, C.POINTER(vp)]),
    "bp_scalars_to_bytes": (ci, [vp, ci, vp, sz, vp]),
    "bp_scalars_from_bytes": (ci, [vp, ci, vp, sz, vp, vp]),
    "bp_points_compress": (ci, [vp, ci, vp, vp, vp, sz, vp]),
    "bp_points_decompress": (ci, [vp, ci, vp, sz, vp, vp, vp, vp]),
    "bp_ipa_begin_resident": (ci, [vp, vp, ci, sz, sz, vp, vp, sz, C.POINTER(vp)]),
    "bp_ipa_begin_resident_powers": (ci, [vp, vp, ci, sz, sz, vp, vp, sz, C.POINTER(vp)]),
    "bp_ipa_scale_h": (ci, [vp, vp]),
    "bp_ipa_set_u": (ci, [vp, vp]),
    "bp_ipa_set_u_resident": (ci, [vp, sz, vp]),
    "bp_ipa_read": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "bp_ipa_free": (None, [vp]),
    "bp_range_batch_begin": (ci, [vp, vp, sz, sz, C.c_uint, C.POINTER(vp)]),
    "bp_range_batch_free": (None, [vp]),
    "bp_range_batch_size": (sz, [vp]),
    "bp_range_batch_commit": (ci, [vp, vp, vp, vp, C.c_uint, vp, vp, vp]),
    "bp_range_batch_poly": (ci, [vp, vp, vp, vp, vp]),
    "bp_range_batch_tcommit": (ci, [vp, vp, vp]),
    "bp_range_batch_eval": (ci, [vp, vp, vp]),
    "bp_range_batch_ipa_begin": (ci, [vp, vp]),
    "bp_range_batch_ipa_len": (sz, [vp]),
    "bp_range_batch_round_lr": (ci, [vp, vp]),
    "bp_range_batch_round_fold": (ci, [vp, vp, vp]),
    "bp_range_batch_finish": (ci, [vp, vp]),
    "bp_bench_int_mad_peak": (ci, [vp, ci, C.POINTER(C.c_double)]),
    "bp_microbench": (ci, [vp, ci, ci, ci, C.POINTER(C.c_float)]),
    "bp_last_kernel_ms": (ci, [vp, ci, C.POINTER(C.c_float)]),
    "bp_set_timing": (ci, [vp, ci]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here = header/library mismatch
    _fn.restype = _res
    _fn.argtypes = _args
