"""ctypes binding of libirec_hip.so (include/irec.h).  Fails loudly when the library has not been built."""
import ctypes
import os

from .errors import IrecLibraryError  # noqa: F401  (re-exported: irec._lib.IrecLibraryError)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "libirec_hip.so")   # the product library; no environment variable is read

IREC_OK = 0
IREC_E_INVALID = -1
IREC_E_HIP = -2
IREC_E_NO_DEVICE = -3
IREC_E_WORKSPACE = -4
IREC_FLAG_FORCE_GENERIC = 1
IREC_FLAG_FUSED_PHILOX = 2
IREC_FLAG_ONE_TABLE = 4
IREC_FLAG_TEAM = 8
IREC_FLAG_NO_SPLIT = 16
IREC_FLAG_TEST_SPLIT_ORPHAN = 32   # test hook: the partner workgroups / teams of a shared block leave at once
IREC_FLAG_LISTED_ORDER = 262144     # team encoder: deal the rows of a mid-size call as listed, not by cost (diagnostics)
IREC_FLAG_MARGINS = 524288           # irec_beam_encode_ex: the call reports its top-B margins (out_margin)
IREC_FLAG_NO_TEN = 1048576            # diagnostics: plain calls of at most ten beams stay on encode_team_kernel<10,..> (not encode_ten_kernel)
IREC_FLAG_TABLES_PRESENT = 65536     # the previous call on this workspace / stream was this call's twin: no table launches
IREC_FLAG_REUSE_TABLES = 64        # keep a proposal table whose stamp in the workspace head matches the call's key
IREC_FLAG_SHAPE_SHIFT = 8          # diagnostic workgroup shapes of the team encoder (include/irec.h)
IREC_FLAG_SHAPE = {"default": 0, "2": 2 << 8, "3": 3 << 8, "1x2": 5 << 8, "team": 6 << 8}
IREC_TABLE_STEPS_DEFAULT = 32
IREC_TABLE_STEPS_MAX = 4096
BIG_PRIME = 10007
MAX_BEAMS = 256
MAX_PARTITIONS = 65536


class IrecParams(ctypes.Structure):
    _fields_ = [("kl_per_partition", ctypes.c_float), ("n_samples", ctypes.c_int32), ("n_beams", ctypes.c_int32),
                ("flags", ctypes.c_int32), ("table_dims", ctypes.c_int32 * 4), ("table_steps", ctypes.c_int32)]


class IrecTables(ctypes.Structure):
    """irec_tables of include/irec.h: the caller-supplied tables of a context (irec_create_with)."""
    _fields_ = [("lut10007", ctypes.c_void_p), ("aux_ratios", ctypes.c_void_p), ("n_aux_ratios", ctypes.c_int32)]


class IrecPlanInfo(ctypes.Structure):
    """irec_plan_info of include/irec.h."""
    _fields_ = [("kernel", ctypes.c_char * 64), ("table_kernel", ctypes.c_char * 32), ("grid", ctypes.c_int32),
                ("waves_per_wg", ctypes.c_int32), ("teams_per_wg", ctypes.c_int32), ("lds_bytes", ctypes.c_int32),
                ("table_steps", ctypes.c_int32), ("n_tables", ctypes.c_int32), ("split", ctypes.c_int32), ("n_cu", ctypes.c_int32),
                ("clock_mhz", ctypes.c_int32), ("split_beams", ctypes.c_int32), ("table_bytes", ctypes.c_int64), ("workspace_bytes", ctypes.c_int64)]

    def as_dict(self):
        return {name: (getattr(self, name).decode() if isinstance(getattr(self, name), bytes) else int(getattr(self, name)))
                for name, _ in self._fields_}


class IrecPlanDetail(ctypes.Structure):
    """irec_plan_detail of csrc/irec_internal.h (test hook irec_test_plan)."""
    _fields_ = [("kind", ctypes.c_int32), ("grid", ctypes.c_int32), ("teams_per_wg", ctypes.c_int32), ("coop_width", ctypes.c_int32),
                ("coop_beams", ctypes.c_int32), ("gang_chunks", ctypes.c_int32), ("placed", ctypes.c_int32), ("split_blocks", ctypes.c_int32),
                ("share_first", ctypes.c_int64), ("n_slots", ctypes.c_int64), ("slabs_in_workspace", ctypes.c_int64),
                ("slab_bytes", ctypes.c_int64), ("fixed_bytes", ctypes.c_int64), ("exchange_rows", ctypes.c_int32), ("exchange_keys", ctypes.c_int32)]

    def as_dict(self):
        return {name: int(getattr(self, name)) for name, _ in self._fields_}


_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_i32 = ctypes.c_int32
_PP = ctypes.POINTER(IrecParams)

# name -> (restype, argtypes); every symbol declared in include/irec.h
SIGNATURES = {
    "irec_last_error": (ctypes.c_char_p, []),
    "irec_version": (ctypes.c_char_p, []),
    "irec_n_samples": (_i32, [ctypes.c_double, ctypes.c_double]),
    "irec_codelength": (ctypes.c_double, [_i64, _i32]),
    "irec_build_lut": (ctypes.c_int, [_vp]),
    "irec_tf_shuffle_perm": (ctypes.c_int, [_i64, _i64, _vp]),
    "irec_philox_uniform_int": (ctypes.c_int, [_i64, _i64, _vp]),
    "irec_importance_encode": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, ctypes.c_double, ctypes.c_double, _i64,
                                              ctypes.POINTER(_i64), _vp]),
    "irec_tf_stateless_normal": (ctypes.c_int, [_i64, _i64, _i64, _vp]),
    "irec_importance_decode": (ctypes.c_int, [_vp, _vp, _i64, _i64, _i64, _vp]),
    "irec_importance_n_samples": (_i64, [ctypes.c_double]),
    "irec_tf_random_normal": (ctypes.c_int, [_i64, _i64, _vp]),
    "irec_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_vp)]),
    "irec_create_ex": (ctypes.c_int, [ctypes.c_int, _vp, ctypes.POINTER(_vp)]),
    "irec_create_with": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(IrecTables), ctypes.POINTER(_vp)]),
    "irec_max_partitions": (_i32, [_vp]),
    "irec_destroy": (None, [_vp]),
    "irec_encode_workspace_bytes": (ctypes.c_size_t, [_vp, _PP, _i32, _i32]),
    "irec_encode_workspace_bytes_for": (ctypes.c_size_t, [_vp, _PP, _i64, _i32, _i32]),
    "irec_test_plan": (ctypes.c_int, [_i32, _i32, _PP, _i64, _i32, _i32, _vp, _vp]),
    "irec_encode_plan": (ctypes.c_int, [_vp, _PP, _i64, _i32, _i32, ctypes.POINTER(IrecPlanInfo)]),
    "irec_block_kl": (ctypes.c_int, [_vp, _PP, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "irec_beam_encode": (ctypes.c_int, [_vp, _PP, _i64, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _i32,
                                        _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    "irec_beam_encode_ex": (ctypes.c_int, [_vp, _PP, _i64, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _i32,
                                           _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    "irec_beam_decode": (ctypes.c_int, [_vp, _PP, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    "irec_decode_workspace_bytes": (ctypes.c_size_t, [_vp, _PP, _i32]),
    "irec_beam_decode_ws": (ctypes.c_int, [_vp, _PP, _i64, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp,
                                           ctypes.c_size_t, _vp]),
    "irec_decode_tensors_supported": (_i32, [_PP, _i32, _i32]),
    "irec_beam_decode_tensors": (ctypes.c_int, [_vp, _PP, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp,
                                                ctypes.c_size_t, _vp]),
    "irec_io_last_error": (ctypes.c_char_p, []),
    "irec_ac_encode": (ctypes.c_int, [_vp, _i32, _vp, _i64, _i32, _vp, _i64, ctypes.POINTER(_i64)]),
    "irec_ac_decode": (ctypes.c_int, [_vp, _i32, _vp, _i64, _i32, _vp, _i64, ctypes.POINTER(_i64)]),
    "irec_rec_pack_bits": (_i64, [_vp, _i64, _vp, _i64]),
    "irec_rec_unpack_bits": (_i64, [_vp, _i64, _vp, _i64]),
    "irec_rec_encode_file": (_i64, [ctypes.c_uint32] * 6 + [_i32, _vp, _vp, _vp, _vp, _i64]),
    "irec_rec_decode_file": (ctypes.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64]),
    "irec_rec_encode_files": (_i64, [ctypes.c_uint32] * 6 + [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _i64, _vp, _i32]),
    "irec_rec_decode_files": (ctypes.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i32]),
    "irec_device_uniform_int": (ctypes.c_int, [_vp, _i64, _i64, _vp, _vp]),
    "irec_shim_stats": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "irec_shim_cat_elu": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "irec_shim_residual_elu": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_float, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "irec_test_decoder_sqrt": (ctypes.c_int, [_vp, _vp, _vp]),
    "irec_test_reduce_scatter": (ctypes.c_int, [_vp, _vp, _vp, _i32, _vp]),
    "irec_test_select": (ctypes.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "irec_test_select_quick": (ctypes.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "irec_test_proposal_table": (ctypes.c_int, [_vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    "irec_device_tables": (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp),
                                          ctypes.POINTER(_vp)]),
}

_lib = None


_lib_path = None


def load(path=None):
    """Load libirec_hip.so.  No fallback of any kind: a missing library is an error.

    `path`: a diagnostic build (csrc/variants/*.so) to load INSTEAD of the product library -- an explicit argument of the
    A/B tooling (scripts/with_lib.py), which must make this call before anything else loads the library; the product
    itself never passes it and no environment variable can redirect the loader."""
    global _lib, _lib_path
    if _lib is not None:
        if path is not None and os.path.abspath(path) != _lib_path:
            raise IrecLibraryError(f"irec._lib.load({path!r}): {_lib_path} is loaded already")
        return _lib
    variant = path is not None
    path = os.path.abspath(path) if variant else LIB_PATH
    if not os.path.exists(path):
        raise IrecLibraryError(
            f"{path} not found: build it with `make -C relative-entropy-coding_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`).  irec has no CPU fallback.")
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        if variant and not hasattr(lib, name):
            continue             # a diagnostic build of older sources (same-box A/B against an earlier round) may lack newer entries
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib, _lib_path = lib, path
    return _lib


def check(status, what=""):
    if status != IREC_OK:
        msg = load().irec_last_error().decode("utf-8", "replace")
        raise IrecLibraryError(f"{what}: irec_status {status}: {msg}")
