"""ctypes binding of libpfa_hip.so (include/pfa.h).

The library is the product: there is no CPU fallback.  ``load()`` raises
``PfaLibraryError`` when the shared object has not been built, and ``init()``
raises when no gfx950 device is visible.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PFA_LIB") or os.path.join(_HERE, "libpfa_hip.so")   # PFA_LIB: timing experiments with other builds

PFA_ABI_VERSION = 7
PFA_NO_ROW_IDX = 0xFFFFFFFF
PFA_FLAG_CDS = 1
PFA_WANT_VAR_BITMAP = 1
PFA_WANT_SN_BITMAP = 2
PFA_SYN_BINS = 65
PFA_MAX_ROWS = 1 << 24
PFA_MAX_ROWS_CDS = PFA_MAX_ROWS
PFA_COMM_ID_BYTES = 128
PFA_ERR_FORMAT = -7
PFA_NO_KEY = 0xFFFFFFFF
PFA_BATCH_REGULAR = 1

u8p = ctypes.POINTER(ctypes.c_uint8)
u32p = ctypes.POINTER(ctypes.c_uint32)
u64p = ctypes.POINTER(ctypes.c_uint64)

# numpy mirrors of the C structs (layout checked against sizeof in tests)
DESC_DTYPE = np.dtype([
    ("base_off", "<u8"), ("n_rows", "<u4"), ("n_sites", "<u4"), ("row_stride", "<u4"),
    ("row_idx_off", "<u4"), ("flags", "<u4"), ("sfs_off", "<u4"), ("tile_off", "<u4"),
    ("mat_rows", "<u4"),
], align=True)
RESULT_DTYPE = np.dtype([
    ("H", "<u8"), ("H_S", "<u8"), ("H_N", "<u8"), ("seg_sites", "<u4"), ("sfs_bad", "<u4"),
    ("seg_S", "<u4"), ("seg_N", "<u4"), ("nstops", "<u4"), ("missing_cols", "<u4"),
    ("status", "<u4"), ("reserved", "<u4"),
], align=True)
FASTA_REC_DTYPE = np.dtype([
    ("name_off", "<u8"), ("name_len", "<u4"), ("seq_len", "<u4"),
    ("first_line_off", "<u8"), ("end_off", "<u8"),
], align=True)
FILE_INFO_DTYPE = np.dtype([
    ("status", "<i4"), ("n_rows", "<u4"), ("n_sites", "<u4"), ("row_stride", "<u4"),
    ("base_off", "<u8"), ("names_off", "<u8"),
], align=True)
PFA_FILE_NOT_FASTA = 1
PFA_FILE_UNEQUAL = 2
PFA_FILE_DEFERRED = 3
PFA_FILE_UNSUPPORTED = 4
PFA_ERR_IO = -8
PFA_ERR_UNSUPPORTED = -9
FASTA_ROW_DTYPE = np.dtype([("name_off", "<u8"), ("name_len", "<u4"), ("seq_len", "<u4"), ("seq_off", "<u8")], align=True)
assert FILE_INFO_DTYPE.itemsize == 32 and FASTA_ROW_DTYPE.itemsize == 24
assert DESC_DTYPE.itemsize == 40 and RESULT_DTYPE.itemsize == 56 and FASTA_REC_DTYPE.itemsize == 32


class SynthParamsC(ctypes.Structure):
    _fields_ = [("seed", ctypes.c_uint64), ("first_aln", ctypes.c_uint64), ("n_aln", ctypes.c_uint32),
                ("n_rows", ctypes.c_uint32), ("n_sites", ctypes.c_uint32), ("row_stride", ctypes.c_uint32),
                ("seg_thresh", ctypes.c_uint32), ("tri_thresh", ctypes.c_uint32), ("n_units", ctypes.c_uint32),
                ("coding", ctypes.c_uint32)]


class IngestOptsC(ctypes.Structure):
    _fields_ = [("n_threads", ctypes.c_uint32), ("stride_align", ctypes.c_uint32), ("unit_flags", ctypes.c_uint32),
                ("n_keys", ctypes.c_uint32), ("keys", ctypes.POINTER(ctypes.c_char_p)), ("slot_bytes", ctypes.c_uint64),
                ("n_slots", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]


class IngestBatchC(ctypes.Structure):
    _fields_ = [("n_files", ctypes.c_uint32), ("n_units", ctypes.c_uint32), ("first_pos", ctypes.c_uint32), ("flags", ctypes.c_uint32),
                ("file_idx", u32p), ("status", ctypes.POINTER(ctypes.c_int32)), ("n_rows", u32p), ("n_sites", u32p),
                ("key_rows", u32p), ("unit_file", u32p), ("unit_key", u32p), ("unit_rows", u32p), ("bad_bytes", u32p),
                ("plan", ctypes.c_void_p)]


class PfaLibraryError(RuntimeError):
    pass


class PfaError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libpfa_hip error {code}: {message}")
        self.code = code


# every symbol include/pfa.h declares: (restype, argtypes)
_vp = ctypes.c_void_p
_PROTOTYPES = {
    "pfa_abi_version": (ctypes.c_int, []),
    "pfa_build_id": (ctypes.c_char_p, []),
    "pfa_device_count": (ctypes.c_int, []),
    "pfa_init": (ctypes.c_int, [ctypes.c_int]),
    "pfa_prewarm": (None, [ctypes.c_int]),
    "pfa_prewarm_join": (None, []),
    "pfa_device_index": (ctypes.c_int, []),
    "pfa_shutdown": (None, []),
    "pfa_last_error": (ctypes.c_char_p, []),
    "pfa_sync": (ctypes.c_int, []),
    "pfa_dev_alloc": (ctypes.c_int, [ctypes.c_size_t, ctypes.POINTER(_vp)]),
    "pfa_dev_free": (ctypes.c_int, [_vp]),
    "pfa_upload": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t]),
    "pfa_upload_async": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t]),
    "pfa_download": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t]),
    "pfa_tile_sites": (ctypes.c_uint32, [ctypes.c_uint32, ctypes.c_uint32]),
    "pfa_var_words_per_tile": (ctypes.c_uint32, [ctypes.c_uint32, ctypes.c_uint32]),
    "pfa_plan_create": (ctypes.c_int, [_vp, ctypes.c_uint32, _vp, ctypes.c_size_t, ctypes.c_uint64,
                                       ctypes.c_uint32, ctypes.POINTER(_vp)]),
    "pfa_plan_run": (ctypes.c_int, [_vp, _vp, ctypes.c_int]),
    "pfa_plan_set_timing": (ctypes.c_int, [_vp, ctypes.c_int]),
    "pfa_plan_kernel_ms": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_float), ctypes.c_int]),
    "pfa_plan_fetch": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pfa_plan_run_fetch": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pfa_plan_results_dev": (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(ctypes.c_size_t)]),
    "pfa_plan_codon_sets": (ctypes.c_int, [_vp, _vp, ctypes.c_uint32, _vp, ctypes.c_uint64]),
    "pfa_plan_n_tiles": (ctypes.c_uint32, [_vp]),
    "pfa_plan_tiles_per_chunk": (ctypes.c_uint32, [_vp]),
    "pfa_plan_lds_bytes": (ctypes.c_uint32, [_vp]),
    "pfa_plan_rows_per_chunk": (ctypes.c_uint32, [_vp]),
    "pfa_plan_tile_width": (ctypes.c_uint32, [_vp]),
    "pfa_plan_n_launches": (ctypes.c_uint32, [_vp]),
    "pfa_plan_describe": (ctypes.c_char_p, [_vp]),
    "pfa_host_l3_domains": (ctypes.c_int, [ctypes.c_char_p, _vp, ctypes.c_int]),
    "pfa_plan_sfs_words": (ctypes.c_uint64, [_vp]),
    "pfa_plan_destroy": (None, [_vp]),
    "pfa_scan_batch": (ctypes.c_int, [_vp, ctypes.c_uint64, _vp, ctypes.c_uint32, _vp, ctypes.c_size_t,
                                      _vp, _vp, _vp, _vp, _vp, _vp]),
    "pfa_bitmap_positions": (ctypes.c_int64, [_vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                              ctypes.c_uint32, _vp, ctypes.c_uint64]),
    "pfa_format_plain_rows": (ctypes.c_int64, [_vp, _vp, ctypes.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int,
                                               _vp, ctypes.c_uint64, _vp]),
    "pfa_format_cds_rows": (ctypes.c_int64, [_vp, _vp, _vp, _vp, ctypes.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int,
                                             _vp, ctypes.c_uint64, _vp]),
    "pfa_format_double": (ctypes.c_int, [ctypes.c_double, _vp]),
    "pfa_classify_codon_mask": (ctypes.c_uint32, [ctypes.c_uint64]),
    "pfa_classify_codon_masks_dev": (ctypes.c_int, [_vp, ctypes.c_uint32, _vp]),
    "pfa_codon_syn_thirds": (ctypes.c_uint32, [ctypes.c_uint32]),
    "pfa_synth_fill_dev": (ctypes.c_int, [_vp, ctypes.POINTER(SynthParamsC), _vp, _vp]),
    "pfa_synth_fill_host": (ctypes.c_int, [_vp, ctypes.POINTER(SynthParamsC), _vp, _vp]),
    "pfa_fasta_index": (ctypes.c_int64, [_vp, ctypes.c_uint64, _vp, ctypes.c_uint64]),
    "pfa_fasta_pack": (ctypes.c_int, [_vp, ctypes.c_uint64, _vp, ctypes.c_uint64, _vp, ctypes.c_uint32]),
    "pfa_fasta_load_files": (ctypes.c_int, [_vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, _vp,
                                            ctypes.c_uint64, _vp, _vp, ctypes.c_uint64, u64p, u64p]),
    "pfa_fasta_index_compact": (ctypes.c_int64, [_vp, ctypes.c_uint64, _vp, ctypes.c_uint64]),
    "pfa_fasta_pack2": (ctypes.c_int, [_vp, ctypes.c_uint64, _vp]),
    "pfa_ingest_open": (ctypes.c_int, [ctypes.POINTER(ctypes.c_char_p), ctypes.c_uint32, ctypes.POINTER(IngestOptsC), ctypes.POINTER(_vp)]),
    "pfa_ingest_open_dir": (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(IngestOptsC), ctypes.POINTER(_vp), ctypes.POINTER(ctypes.c_uint32)]),
    "pfa_ingest_open_names": (ctypes.c_int, [ctypes.c_char_p, _vp, _vp, ctypes.c_uint32, ctypes.POINTER(IngestOptsC), ctypes.POINTER(_vp)]),
    "pfa_ingest_path": (ctypes.c_char_p, [_vp, ctypes.c_uint32]),
    "pfa_ingest_feed": (ctypes.c_int, [_vp, _vp, ctypes.c_uint32, ctypes.c_int]),
    "pfa_ingest_backlog": (ctypes.c_uint32, [_vp]),
    "pfa_ingest_next": (ctypes.c_int, [_vp, ctypes.POINTER(IngestBatchC)]),
    "pfa_ingest_stats": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_double)]),
    "pfa_ingest_names": (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp)]),
    "pfa_ingest_close": (None, [_vp]),
    "pfa_format_rows_indexed": (ctypes.c_int64, [_vp, _vp, _vp, _vp, ctypes.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                                 ctypes.c_int, ctypes.c_int, _vp, ctypes.c_uint64, _vp]),
    "pfa_host_alloc": (ctypes.c_int, [ctypes.c_size_t, ctypes.POINTER(_vp)]),
    "pfa_host_free": (ctypes.c_int, [_vp]),
    "pfa_host_cpu_share": (ctypes.c_double, [ctypes.c_char_p, ctypes.c_char_p]),
    "pfa_comm_unique_id": (ctypes.c_int, [_vp]),
    "pfa_comm_init": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _vp]),
    "pfa_comm_rank": (ctypes.c_int, []),
    "pfa_comm_world": (ctypes.c_int, []),
    "pfa_comm_barrier": (ctypes.c_int, []),
    "pfa_comm_all_reduce_max": (ctypes.c_int, [ctypes.POINTER(ctypes.c_double)]),
    "pfa_comm_all_gather": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t]),
    "pfa_comm_all_gather_results": (ctypes.c_int, [_vp, _vp]),
    "pfa_comm_gatherv": (ctypes.c_int, [_vp, ctypes.c_uint64, _vp, _vp, ctypes.c_int]),
    "pfa_comm_destroy": (ctypes.c_int, []),
}

_lib = None
_device = None


def load():
    """dlopen libpfa_hip.so and bind every prototype (works without a GPU)."""
    global _lib
    if _lib is not None:
        return _lib
    if int(os.environ.get("WORLD_SIZE", "1") or 1) > 1:
        # ranks of a multi-GPU job, whoever launched them (dist.spawn_ranks sets it too): RCCL's intra-node transport shares
        # buffers through IPC handles and this pool's kernel driver only implements the dmabuf flavour — with the runtime's
        # legacy default hipIpcGetMemHandle fails inside ncclCommInitRank.  Must be in the environment before the HIP
        # runtime comes up (the first pfa_* call below); a value the user exported is kept.
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not os.path.exists(LIB_PATH):
        raise PfaLibraryError(
            f"{LIB_PATH} is missing: build it with `python -m polyfasta_amd._build` "
            "(hipcc, gfx950).  polyfasta_amd has no CPU fallback.")
    want = None
    if not os.environ.get("PFA_LIB"):
        # the library must have been compiled from THESE sources (contents, not modification times): a tree whose
        # sources changed under an old library rebuilds it when hipcc is there, and refuses otherwise
        from . import _build
        want = _build.source_hash()
        if want is not None and _build.built_id() != want:
            try:
                _build.build()
            except Exception as exc:
                raise PfaLibraryError(f"{LIB_PATH} was built from other sources than this tree's (build id {_build.built_id()}, "
                                      f"sources {want}) and cannot be rebuilt here: {exc}") from exc
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as exc:  # pragma: no cover
        raise PfaLibraryError(f"cannot load {LIB_PATH}: {exc}") from exc
    for name, (restype, argtypes) in _PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.pfa_abi_version() != PFA_ABI_VERSION:
        raise PfaLibraryError("libpfa_hip.so ABI version mismatch; rebuild it")
    if not os.environ.get("PFA_LIB") and want is not None and lib.pfa_build_id().decode() != want:
        # (rebuilt just now, but this process had the old file mapped already: python -m polyfasta_amd loads it early)
        raise PfaLibraryError(f"{LIB_PATH} has been rebuilt from this tree's sources; an older build was already loaded "
                              "in this process: run the command again")
    _lib = lib
    return lib


def last_error():
    return (load().pfa_last_error() or b"").decode("utf-8", "replace")


def check(rc):
    if rc < 0:
        raise PfaError(rc, last_error())
    return rc


def init(device=None):
    """Bind this process to one GPU: `device`, else LOCAL_RANK (one process per GPU), else 0.
    Raises without a gfx950 device, and when LOCAL_RANK names a device that is not there —
    ranks never share a GPU silently (PFA_ALLOW_SHARED_GPU=1: rehearsals on a one-GPU box)."""
    global _device
    lib = load()
    if device is None:
        if _device is not None:
            return _device
        device = int(os.environ.get("LOCAL_RANK", "0"))
        ndev = lib.pfa_device_count()
        if ndev > 0 and device >= ndev and os.environ.get("PFA_ALLOW_SHARED_GPU"):
            device %= ndev
    if _device == device:
        return device
    check(lib.pfa_init(int(device)))
    _device = device
    return device


def shutdown():
    global _device
    if _lib is not None:
        _lib.pfa_shutdown()
    _device = None


def ptr(arr):
    """void* of a C-contiguous numpy array (None -> NULL)."""
    if arr is None:
        return None
    assert arr.flags["C_CONTIGUOUS"]
    return arr.ctypes.data_as(ctypes.c_void_p)
