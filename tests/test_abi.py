"""CPU tests of the drop-in boundary: the shared library loads, exports every
symbol include/bp_gpu.h declares, and rejects bad arguments without a GPU."""
import ctypes as C
import os
import re

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "bp_gpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bp_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    syms = declared_symbols()
    for must in ("bp_init", "bp_shutdown", "bp_msm", "bp_msm_dev", "bp_msm_resident", "bp_bases_register",
                 "bp_inner_product", "bp_fold_scalars", "bp_fold_generators", "bp_scale_points", "bp_s_vector",
                 "bp_ipa_begin", "bp_ipa_round_lr", "bp_ipa_round_fold", "bp_ipa_free", "bp_batch_inverse",
                 "bp_normalize_batch", "bp_vec_op", "bp_sum", "bp_powers"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from bulletproofs_amd import _lib  # preloads the one HIP runtime the library binds to

    assert os.path.samefile(_lib.LIB_PATH, os.path.join(ROOT, "bulletproofs_amd", "libbp_gpu.so"))
    lib = _lib.lib
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_python_binding_covers_header():
    from bulletproofs_amd import _lib

    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_argument_errors_without_device():
    from bulletproofs_amd import _lib

    lib = _lib.lib
    assert lib.bp_init(0, None) == _lib.BP_E_ARG
    h = C.c_void_p()
    rc = lib.bp_init(10_000, C.byref(h))  # no such device: BP_E_HIP without any GPU, BP_E_ARG with one
    assert rc in (_lib.BP_E_HIP, _lib.BP_E_ARG) and not h.value
    assert lib.bp_sync(None) == _lib.BP_E_ARG
    assert lib.bp_ipa_round_fold(None, None, None) == _lib.BP_E_ARG
    assert lib.bp_bases_len(None) == 0
    lib.bp_shutdown(None)
    lib.bp_ipa_free(None)
    lib.bp_bases_free(None)
    assert b"null" in lib.bp_last_error(None)
