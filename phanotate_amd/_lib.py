"""ctypes binding of libphx.so (include/phx.h).  Fails loudly: there is no Python/CPU fallback."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libphx.so")
MAXC = 16
N_STAGES = 14


class Params(C.Structure):
    _fields_ = [
        ("minlen", C.c_int32),
        ("n_start", C.c_int32),
        ("start", (C.c_char * 4) * MAXC),
        ("start_w", C.c_double * MAXC),
        ("n_stop", C.c_int32),
        ("stop", (C.c_char * 4) * MAXC),
        ("start_w_text", (C.c_char * 32) * MAXC),
    ]


class Gene(C.Structure):
    _fields_ = [("left", C.c_int32), ("right", C.c_int32), ("strand", C.c_int32), ("frame", C.c_int32), ("score", C.c_double)]


class Result(C.Structure):
    _fields_ = [("status", C.c_int32), ("n_genes", C.c_int32), ("genes", C.POINTER(Gene))]


class Globals(C.Structure):
    _fields_ = [
        ("L", C.c_int64), ("pstop", C.c_double),
        ("background_rbs", C.c_double * 28), ("training_rbs", C.c_double * 28),
        ("pos_max", C.c_double * 4), ("pos_min", C.c_double * 4),
        ("n_orf", C.c_int32), ("n_group", C.c_int32), ("n_node", C.c_int32), ("n_edge", C.c_int32), ("n_bridge", C.c_int32),
        ("n_limbs", C.c_int32), ("sssp_sweeps", C.c_int32), ("sssp_iters", C.c_int32), ("status", C.c_int32),
        ("sssp_kernel", C.c_int32), ("sssp_handed_back", C.c_int32), ("tie", C.c_int32), ("certified", C.c_int32),
        ("rbs_background_count", C.c_uint32 * 28), ("rbs_training_count", C.c_uint32 * 28), ("gc_max_count", C.c_uint32 * 4), ("gc_min_count", C.c_uint32 * 4),
        ("gc_count", C.c_int64),
    ]


GENE_DT = np.dtype([("left", "i4"), ("right", "i4"), ("strand", "i4"), ("frame", "i4"), ("score", "f8")], align=True)
ORF_DT = np.dtype([("start", "i4"), ("stop", "i4"), ("frame", "i4"), ("length", "i4"), ("rbs", "i4"), ("startidx", "i4"), ("group", "i4"),
                   ("hist", "i4", (9,)), ("pstop", "f8"), ("weight_rbs", "f8"), ("S", "f8"), ("weight", "f8")], align=True)
NODE_DT = np.dtype([("pos", "i4"), ("type", "i1"), ("frame", "i1"), ("pad", "i2"), ("other", "i4"), ("refidx", "i4"), ("o", "f8")], align=True)
EDGE_DT = np.dtype([("src", "i4"), ("dst", "i4"), ("w", "f8"), ("inexact", "i4"), ("pad", "i4"), ("d1", "f8"), ("d2", "f8"), ("err", "f8")], align=True)

_lib = None


class PhxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libphx error %d: %s" % (code, msg))
        self.code = code


def _preload_hip_runtime():
    """One HIP runtime per process.  libphx.so needs `libamdhip64.so.7`; the PyTorch-ROCm wheel ships its
    own copy with the same soname, loaded by path from torch/lib.  If libphx pulled in /opt/rocm's copy
    first and torch its own later, two runtimes would fight over the device (torch then fails to
    initialise) and stream / pointer handles would not be interchangeable.  So when a torch install is
    present, its runtime is mapped first and libphx binds to it by soname; without torch the system
    ROCm runtime is used."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return  # already mapped by torch
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load libphx.so.  Raises if the HIP extension has not been built — the product never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO):
        raise ImportError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C phanotate_amd/csrc`; phanotate_amd has no CPU fallback" % SO)
    _preload_hip_runtime()
    L = C.CDLL(SO)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    P = C.POINTER
    sig = {
        "phx_version": (C.c_int, []),
        "phx_device_count": (C.c_int, []),
        "phx_strerror": (C.c_char_p, [C.c_int]),
        "phx_last_error": (C.c_char_p, [vp]),
        "phx_default_params": (None, [P(Params)]),
        "phx_params_from_flags": (C.c_int, [C.c_char_p, C.c_char_p, i32, P(Params)]),
        "phx_set_exact": (C.c_int, [vp, C.c_int]),
        "phx_pool_create": (C.c_int, [P(Params), i32, P(i32), C.c_uint32, P(vp)]),
        "phx_pool_destroy": (None, [vp]),
        "phx_pool_last_error": (C.c_char_p, [vp]),
        "phx_pool_annotate": (C.c_int, [vp, i32, P(C.c_char_p), P(i64), i64, vp, vp, vp, P(Result)]),
        "phx_dump_text": (C.c_int, [vp, i32, P(vp), P(i64)]),
        "phx_dec_eval": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int]),
        "phx_dd_eval": (C.c_int, [C.c_char_p, C.c_double, C.c_double, C.c_double, C.c_double, P(C.c_double), P(C.c_double)]),
        "phx_dd_shortest": (C.c_int, [C.c_double, P(C.c_uint64), P(i32)]),
        "phx_create": (C.c_int, [P(Params), C.c_int, vp, P(vp)]),
        "phx_create_ex": (C.c_int, [P(Params), C.c_int, vp, C.c_uint32, P(vp)]),
        "phx_destroy": (None, [vp]),
        "phx_annotate": (C.c_int, [vp, i32, P(C.c_char_p), P(i64), P(Result)]),
        "phx_free_results": (None, [P(Result), i32]),
        "phx_upload": (C.c_int, [vp, i32, P(C.c_char_p), P(i64)]),
        "phx_attach": (C.c_int, [vp, i32, vp, P(i64)]),
        "phx_set_trnas": (C.c_int, [vp, vp, vp, vp]),
        "phx_run": (C.c_int, [vp]),
        "phx_run_async": (C.c_int, [vp]),
        "phx_wait": (C.c_int, [vp]),
        "phx_download": (C.c_int, [vp, P(Result)]),
        "phx_download_flat": (C.c_int, [vp, vp, i64, vp, vp, P(i64)]),
        "phx_certified": (C.c_int, [vp, vp]),
        "phx_tap_globals": (C.c_int, [vp, i32, P(Globals)]),
        "phx_tap_positions": (C.c_int, [vp, i32, vp, vp, vp, vp]),
        "phx_tap_orfs": (C.c_int, [vp, i32, vp]),
        "phx_tap_nodes": (C.c_int, [vp, i32, vp]),
        "phx_tap_edges": (C.c_int, [vp, i32, vp]),
        "phx_tap_path": (C.c_int, [vp, i32, vp, i32, P(i32), vp, i32]),
        "phx_tap_dist": (C.c_int, [vp, i32, vp, i64]),
        "phx_solve": (C.c_int, [vp, i32, i32, vp, vp, vp, i32, i32, i32, vp, i32, P(i32), vp]),
        "phx_set_profiling": (C.c_int, [vp, C.c_int]),
        "phx_set_profiling_stages": (C.c_int, [vp, C.c_uint32]),
        "phx_get_stage_ms": (C.c_int, [vp, P(C.c_float), P(i32), C.c_int]),
        "phx_stage_name": (C.c_char_p, [C.c_int]),
        "phx_batch_sizes": (C.c_int, [vp, P(i64), P(i64), P(i64), P(i64)]),
        "phx_plan_timeouts": (C.c_int64, [vp]),
        "phx_front_runs": (C.c_int64, [vp]),
        "phx_seg_runs": (C.c_int64, [vp]),
        "phx_seg_fallbacks": (C.c_int64, [vp]),
        "phx_seg_stats": (C.c_int, [vp, C.c_int32, C.c_void_p, C.c_int32]),
        "phx_synth_contig": (C.c_int, [C.c_uint64, i64, C.c_char_p]),
        "phx_fasta_read": (C.c_int, [C.c_char_p, P(vp)]),
        "phx_fasta_count": (i32, [vp]),
        "phx_fasta_record": (C.c_int, [vp, i32, P(vp), P(vp), P(i64)]),
        "phx_fasta_arrays": (C.c_int, [vp, vp, vp, vp]),
        "phx_fasta_free": (None, [vp]),
        "phx_format_tabular": (C.c_int, [i32, vp, vp, vp, vp, P(vp), P(i64)]),
        "phx_free_text": (None, [vp]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)  # AttributeError if the library does not export a declared symbol
        f.restype = res
        f.argtypes = args
    assert GENE_DT.itemsize == C.sizeof(Gene)
    _lib = L
    return L


EXPORTS = ["phx_version", "phx_device_count", "phx_strerror", "phx_last_error", "phx_default_params", "phx_params_from_flags", "phx_set_exact", "phx_pool_create", "phx_pool_destroy", "phx_pool_last_error", "phx_pool_annotate", "phx_dump_text", "phx_dec_eval", "phx_dd_eval", "phx_dd_shortest", "phx_create", "phx_create_ex", "phx_destroy",
           "phx_annotate", "phx_free_results", "phx_upload", "phx_attach", "phx_set_trnas", "phx_run", "phx_run_async", "phx_wait", "phx_download", "phx_download_flat", "phx_certified", "phx_tap_globals",
           "phx_tap_positions", "phx_tap_orfs", "phx_tap_nodes", "phx_tap_edges", "phx_tap_path", "phx_tap_dist", "phx_solve", "phx_set_profiling", "phx_set_profiling_stages",
           "phx_get_stage_ms", "phx_stage_name", "phx_batch_sizes", "phx_plan_timeouts", "phx_front_runs", "phx_seg_runs", "phx_seg_fallbacks", "phx_seg_stats", "phx_synth_contig", "phx_fasta_read", "phx_fasta_count",
           "phx_fasta_record", "phx_fasta_arrays", "phx_fasta_free", "phx_format_tabular", "phx_free_text"]
