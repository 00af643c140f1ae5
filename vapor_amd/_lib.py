"""ctypes binding of libvapor_hip.so (include/vapor_hip.h).

There is no CPU fallback: if the library is missing or no MI355X is visible, every
entry point raises.  Build with `python -m vapor_amd.build` (or __graft_entry__.build()).
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# VAPOR_HIP_LIB: developer builds of the same library (tools/phase_timing.py).  The CPU twin of the C ABI (oracle/, test
# infrastructure) reports the build flag "cpu-twin" and is refused unless VAPOR_ALLOW_TWIN=1 is set as well, which only
# tests/at_size_check.py does (a child process that replays rows of a GPU run on the oracle): a stray VAPOR_HIP_LIB can
# never turn a product run into a CPU run that passes the loader's checks.
SO_PATH = os.environ.get("VAPOR_HIP_LIB") or os.path.join(_HERE, "libvapor_hip.so")

PAIR_DTYPE = np.dtype([("seq1", "<i4"), ("seq2", "<i4"), ("off2", "<i4"), ("k", "<i4"), ("flags", "<u4")])
READ_DTYPE = np.dtype([("ref_a", "<i4"), ("alt_a", "<i4"), ("ref_b", "<i4"), ("alt_b", "<i4"), ("kind", "<i4"),
                       ("locus", "<i4"), ("len_ref", "<i4"), ("len_alt", "<i4")])
SEG_DTYPE = np.dtype([("parent", "<i4"), ("off", "<i4"), ("len", "<i4"), ("flags", "<u4")])
SEG_REVCOMP = 1
MAX_SEGMENTS = 16
STATS_STRIDE = 16
LOCUS_STRIDE = 8
GT_TABLE_N = 65
ST_N_HITS, ST_FIRST_J, ST_LAST_J, ST_C1_KEPT, ST_C1_SUM_ABS = 0, 1, 2, 3, 4
ST_C2_KEPT, ST_C2_COUNT10, ST_N_DIAG, ST_N_LOWER, ST_C2_KEPT_DIAG, ST_STATUS = 5, 6, 7, 8, 9, 15
ST_DIR_C2X, ST_DIR_N, ST_DIR_SUM2, ST_DIR_LISTS = 10, 11, 12, 13
PF_C1, PF_C2, PF_DIR = 1, 2, 4
HF_C1_KEPT, HF_C2_DIAG, HF_C2_ANTI = 1, 2, 4
SEQ_UPPER = 1
E_HIP, E_OVERFLOW, E_KEYERROR, E_ARG, E_NOMEM = -1, -2, -3, -4, -5
MAX_SEQ_LEN = 65535
ABI_VERSION = 3
ABI_DEV_OFFSET = 1000000

EXPORTS = [
    "vapor_abi_version", "vapor_build_flags", "vapor_source_id", "vapor_last_error", "vapor_init", "vapor_destroy", "vapor_set_param",
    "vapor_seqset_create", "vapor_seqset_create_ptrs", "vapor_seqset_create_derived", "vapor_seqset_planes", "vapor_seqset_destroy", "vapor_plan_create", "vapor_plan_destroy",
    "vapor_plan_run", "vapor_plan_timings", "vapor_plan_record_counts", "vapor_plan_algorithmic_bytes", "vapor_plan_fetch_hits",
    "vapor_dotplot_batch", "vapor_score_batch", "vapor_selfplot_qc", "vapor_clean_hits",
    "vapor_plan_set_reads", "vapor_plan_run_loci", "vapor_set_stream", "vapor_plan_run_loci_async", "vapor_plan_sync", "vapor_plan_then", "vapor_plan_after",
    "vapor_cigar2alignstart", "vapor_cigar2alignstart_ops",
    "vapor_bam_open", "vapor_bam_close", "vapor_bam_set_threads", "vapor_bam_last_error", "vapor_bam_chop",
    "vapor_inflate_raw", "vapor_chop_records", "vapor_chop_records_many", "vapor_row_tails", "vapor_crc32",
    "vapor_bam_chop_device", "vapor_bam_batch_destroy", "vapor_bam_fileno", "vapor_bam_threads", "vapor_seqset_create_mixed", "vapor_bam_last_stats",
]

_lib = None


class VaporHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("libvapor_hip: status %d: %s" % (code, msg))
        self.code = code


def _share_torch_hip_runtime() -> None:
    """One HIP runtime per process.  A PyTorch-ROCm wheel carries its own libamdhip64 / libhsa-runtime64; when this library
    is loaded first it binds to the system's copy, and torch's copy then finds no GPU ("No HIP GPUs are available"): two
    runtimes do not share the device.  Loaded after torch it binds to torch's copy (same SONAME) and all is well - so when
    torch is installed but not imported yet, its runtime is loaded here (the shared object only, not the package), and
    the order of imports stops mattering.  VAPOR_HIP_RUNTIME=system keeps the system's copy."""
    import sys
    if "torch" in sys.modules or os.environ.get("VAPOR_HIP_RUNTIME", "") == "system":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        where = list(spec.submodule_search_locations or []) if spec else []
        for d in where:
            p = os.path.join(d, "lib", "libamdhip64.so")
            if os.path.exists(p):
                ctypes.CDLL(p, mode=ctypes.RTLD_GLOBAL)
                return
    except Exception:            # noqa: BLE001 - best effort; the plain load below still works when torch comes first or never
        pass


def load() -> ctypes.CDLL:
    """Load the HIP library or fail loudly."""
    global _lib
    if _lib is not None:
        return _lib
    lib_path = SO_PATH
    if not os.path.exists(lib_path):
        raise RuntimeError("%s is missing - the HIP extension has not been built "
                           "(run `python -m vapor_amd.build`); there is no CPU fallback" % lib_path)
    _share_torch_hip_runtime()
    lib = checked(ctypes.CDLL(lib_path), lib_path)
    _lib = lib
    return _lib


def checked(raw: ctypes.CDLL, lib_path: str) -> ctypes.CDLL:
    """Version and build flags first, on the raw handle (a stale library lacks newer symbols: bind() would die with an
    'undefined symbol' before the rebuild message could appear), then the binding."""
    stale = ("%s is a stale or foreign build (%%s): this package binds ABI version %d of the product build; rebuild with "
             "`python -m vapor_amd.build --force`" % (lib_path, ABI_VERSION))
    try:
        raw.vapor_abi_version.restype = ctypes.c_int
        raw.vapor_build_flags.restype = ctypes.c_char_p
        ver, flags = raw.vapor_abi_version(), raw.vapor_build_flags().decode()
    except AttributeError as e:
        raise RuntimeError(stale % e) from e
    dev_ok = bool(os.environ.get("VAPOR_HIP_LIB"))        # developer tools name their library explicitly
    if ver != ABI_VERSION and not (dev_ok and ver == ABI_VERSION + ABI_DEV_OFFSET):
        raise RuntimeError(stale % ("reports ABI version %d, build flags %r; a developer build loads only through VAPOR_HIP_LIB"
                                    % (ver, flags)))
    if "cpu-twin" in flags.split(","):
        if not (dev_ok and os.environ.get("VAPOR_ALLOW_TWIN") == "1"):
            raise RuntimeError("%s is the CPU twin of the C ABI (oracle/: test infrastructure), not the HIP library; there is "
                               "no CPU fallback (tests set VAPOR_ALLOW_TWIN=1 beside VAPOR_HIP_LIB)" % lib_path)
    elif flags and not dev_ok:
        raise RuntimeError("%s carries developer switches (%s)" % (lib_path, flags))
    elif not dev_ok:
        # the product library must be the one built from the sources beside it (content, not mtime: a stale binary newer
        # than the sources it travelled with would otherwise run - and be profiled - under the new sources' name)
        from . import build as _build
        try:
            raw.vapor_source_id.restype = ctypes.c_char_p
            have = raw.vapor_source_id().decode()
        except AttributeError as e:
            raise RuntimeError(stale % e) from e
        if os.path.exists(_build.KERNEL_FILES[0]) and have != _build.source_id():
            raise RuntimeError(stale % ("built from sources %s, the tree holds %s" % (have, _build.source_id())))
    try:
        return bind(raw)
    except AttributeError as e:
        raise RuntimeError(stale % e) from e


def bind(L: ctypes.CDLL) -> ctypes.CDLL:
    """Declares the argument types of every entry point of include/vapor_hip.h on a loaded library."""
    vp = ctypes.c_void_p
    i32p = ctypes.POINTER(ctypes.c_int32)
    i64p = ctypes.POINTER(ctypes.c_int64)
    u8p = ctypes.POINTER(ctypes.c_uint8)
    u32p = ctypes.POINTER(ctypes.c_uint32)
    f64p = ctypes.POINTER(ctypes.c_double)
    L.vapor_abi_version.restype = ctypes.c_int
    L.vapor_last_error.restype = ctypes.c_char_p
    L.vapor_build_flags.restype = ctypes.c_char_p
    L.vapor_source_id.restype = ctypes.c_char_p
    L.vapor_init.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    L.vapor_destroy.argtypes = [vp]
    L.vapor_set_param.argtypes = [vp, ctypes.c_char_p, ctypes.c_int64]
    L.vapor_seqset_create.argtypes = [vp, ctypes.c_int32, u8p, i64p, i32p, u8p, i32p, ctypes.POINTER(vp)]
    L.vapor_seqset_create_ptrs.argtypes = [vp, ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p), i32p, u8p, i32p, ctypes.POINTER(vp)]
    L.vapor_seqset_create_derived.argtypes = [vp, ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p), i32p, u8p, ctypes.c_int32, i32p, vp, u8p,
                                              i32p, ctypes.POINTER(vp)]
    L.vapor_seqset_planes.argtypes = [vp, ctypes.c_int32, vp, vp, vp]
    L.vapor_seqset_destroy.argtypes = [vp]
    L.vapor_plan_create.argtypes = [vp, vp, ctypes.c_int64, vp, ctypes.POINTER(vp)]
    L.vapor_plan_destroy.argtypes = [vp]
    L.vapor_plan_run.argtypes = [vp, i64p]
    L.vapor_plan_timings.argtypes = [vp, f64p, ctypes.c_int32]
    L.vapor_plan_record_counts.argtypes = [vp, i64p]
    L.vapor_set_stream.argtypes = [vp, vp]
    L.vapor_plan_run_loci_async.argtypes = [vp, vp]
    L.vapor_plan_sync.argtypes = [vp, f64p]
    L.vapor_plan_then.argtypes = [vp, vp]
    L.vapor_plan_after.argtypes = [vp, vp]
    L.vapor_cigar2alignstart.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_int64, i64p]
    L.vapor_cigar2alignstart_ops.argtypes = [vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, i64p]
    L.vapor_bam_open.argtypes = [ctypes.c_char_p, ctypes.POINTER(vp)]
    L.vapor_bam_close.argtypes = [vp]
    L.vapor_bam_set_threads.argtypes = [vp, ctypes.c_int32]
    L.vapor_bam_last_error.restype = ctypes.c_char_p
    L.vapor_bam_chop.argtypes = [vp, ctypes.c_int32, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, vp,
                                 vp, ctypes.c_int64, vp, ctypes.c_int64, vp, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), vp]
    L.vapor_bam_chop_device.argtypes = [vp, vp, ctypes.c_int32, vp, vp, vp, vp, vp, vp, ctypes.c_int32, vp, vp, vp, vp, vp, ctypes.POINTER(vp)]
    L.vapor_bam_batch_destroy.argtypes = [vp]
    L.vapor_bam_last_stats.argtypes = [vp, vp, ctypes.c_int32]
    L.vapor_bam_fileno.argtypes = [vp]
    L.vapor_bam_threads.argtypes = [vp]
    L.vapor_seqset_create_mixed.argtypes = [vp, ctypes.c_int32, vp, vp, vp, vp, vp, ctypes.c_int32, vp, vp, vp, vp, ctypes.POINTER(vp)]
    L.vapor_inflate_raw.argtypes = [vp, ctypes.c_int64, vp, ctypes.c_int64]
    L.vapor_crc32.argtypes = [vp, ctypes.c_int64, ctypes.c_int32]
    L.vapor_crc32.restype = ctypes.c_uint32
    L.vapor_chop_records.argtypes = [ctypes.c_int32, vp, vp, vp, vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, vp, vp]
    L.vapor_chop_records_many.argtypes = [ctypes.c_int32, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_int32, vp, vp, vp, vp, vp, vp, vp]
    L.vapor_row_tails.argtypes = [ctypes.c_int32, vp, vp, vp, vp, vp, vp, ctypes.c_int64, vp]
    L.vapor_plan_algorithmic_bytes.argtypes = [vp, i64p, i64p]
    L.vapor_plan_fetch_hits.argtypes = [vp, ctypes.c_int64, i64p, i32p, u8p, ctypes.c_int64, i64p]
    L.vapor_dotplot_batch.argtypes = [vp, vp, ctypes.c_int64, vp, i32p, ctypes.c_int64, i64p, i64p]
    L.vapor_score_batch.argtypes = [vp, vp, ctypes.c_int64, vp, i64p]
    L.vapor_selfplot_qc.argtypes = [vp, vp, ctypes.c_int32, i32p, i32p, i64p]
    L.vapor_clean_hits.argtypes = [vp, ctypes.c_int64, i32p, i64p, u32p, i64p, u8p]
    L.vapor_plan_set_reads.argtypes = [vp, ctypes.c_int64, vp, ctypes.c_int64, f64p]
    L.vapor_plan_run_loci.argtypes = [vp, vp, f64p, f64p]
    for name in EXPORTS:
        if name not in ("vapor_last_error", "vapor_bam_last_error", "vapor_build_flags", "vapor_source_id", "vapor_crc32"):
            getattr(L, name).restype = ctypes.c_int
    return L


_held = None


def load_holding_gil():
    """The same library through ctypes.PyDLL: its calls keep the interpreter lock.  For the host helpers that take a few
    microseconds and are called once per locus (vapor_chop_records): releasing the lock around them hands it to the other
    chunk's thread each time, and the hand-over costs more than the call (two chunks of a run are scored on two threads)."""
    global _held
    if _held is None:
        load()                                  # (version and build flags of SO_PATH are checked there)
        h = ctypes.PyDLL(SO_PATH)
        vp = ctypes.c_void_p
        h.vapor_chop_records.argtypes = [ctypes.c_int32, vp, vp, vp, vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, vp, vp]
        h.vapor_chop_records.restype = ctypes.c_int
        _held = h
    return _held


def check(rc: int) -> None:
    if rc != 0:
        raise VaporHipError(rc, load().vapor_last_error().decode("utf-8", "replace"))


def ptr(a: np.ndarray, t):
    return a.ctypes.data_as(ctypes.POINTER(t))
