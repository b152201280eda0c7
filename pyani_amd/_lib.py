"""ctypes binding of libpyani_gpu.so (include/pyani_gpu.h).  There is no CPU fallback: if the library is not
built, or no MI355X is visible when a context is created, this raises."""
import ctypes
import os
from pathlib import Path



def configure_runtime(hw_queues: int = 8) -> bool:
    """The ANIm engine's workers own one HIP stream each and count on their kernels overlapping.  ROCm maps a process's streams onto
    GPU_MAX_HW_QUEUES hardware queues (default 4); in a process that also runs RCCL the two workers can end up on one queue and
    serialise (MI355X, one rank through the real backend: 53.7 k pairs/s on C4 against 56.8 k with 8 queues; 58.5 k without RCCL).
    An EXPLICIT call (importing the package must not change the host application's environment): sets the variable if
    it is unset and returns whether it can still take effect, i.e. the HIP runtime has not started in this process (call it before the
    first torch.cuda / Engine use; launch scripts may simply export GPU_MAX_HW_QUEUES=8 — INTEGRATION.md)."""
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(int(hw_queues)))
    return _lib is None

LIB_PATH = Path(__file__).resolve().parent / "libpyani_gpu.so"

PG_OK, PG_E_ARG, PG_E_NODEVICE, PG_E_HIP, PG_E_IO, PG_E_NOMEM, PG_E_KEYSET, PG_E_EMPTY, PG_E_RNA = 0, -1, -2, -3, -4, -5, -6, -7, -8
PG_E_CAPACITY, PG_ANIM_NO_ALIGNMENT = -9, 1
K_TETRA_COUNT, K_TETRA_FINALIZE, K_TETRA_STATS, K_TETRA_PAIRS = 0, 1, 2, 3
(K_ANIM_SEED, K_ANIM_HIT, K_ANIM_CLUSTER, K_ANIM_GAPS, K_ANIM_EXTLANE, K_ANIM_EXTEND, K_ANIM_FINISH) = 4, 5, 6, 7, 8, 9, 10
K_ANIB_BUCKET, K_ANIB_FRAG = 11, 12
K_ANIM_FWD, K_ANIM_BWD = 13, 14
K_SKETCH_PAIRS = 15
K_COUNT = 16
PG_SKETCH_NO_RESULT = 1

# every symbol declared in include/pyani_gpu.h: (name, restype, argtypes)
_vp, _i32, _u32, _u64, _int = ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int
_P = ctypes.POINTER
SIGNATURES = {
    "pg_version": (ctypes.c_char_p, []),
    "pg_create": (_int, [_P(_vp), _int]),
    "pg_destroy": (None, [_vp]),
    "pg_last_error": (ctypes.c_char_p, [_vp]),
    "pg_sync": (_int, [_vp]),
    "pg_add_genome": (_int, [_vp, _vp, _vp, _u32, _P(_i32)]),
    "pg_add_fasta": (_int, [_vp, ctypes.c_char_p, _P(_i32), _P(_u64), _P(_u32)]),
    "pg_add_fasta_batch": (_int, [_vp, _vp, _u32, _u32, _vp, _vp, _vp]),
    "pg_genome_count": (_int, [_vp]),
    "pg_genome_length": (_int, [_vp, _i32, _P(_u64), _P(_u32)]),
    "pg_clear_genomes": (_int, [_vp]),
    "pg_upload": (_int, [_vp]),
    "pg_tetra_algorithmic_bytes": (_int, [_vp, _vp, _u32, _P(_u64), _P(_u64)]),
    "pg_tetra_counts": (_int, [_vp, _vp, _u32, _vp, _vp, _vp]),
    "pg_tetra_zscores": (_int, [_vp, _vp, _vp, _vp, _u32, _vp, _vp]),
    "pg_tetra_corr": (_int, [_vp, _vp, _vp, _u32, _vp]),
    "pg_tetra_matrix": (_int, [_vp, _vp, _u32, _vp, _vp, _vp]),
    "pg_tetra_matrix_enqueue": (_int, [_vp, _vp, _u32, _int]),
    "pg_tetra_matrix_fetch": (_int, [_vp, _u32, _vp, _vp, _vp]),
    "pg_tetra_zscores_dev": (_int, [_vp, _vp, _u32, _vp, _vp]),
    "pg_tetra_corr_rows_dev": (_int, [_vp, _vp, _vp, _u32, _u32, _u32, _vp]),
    "pg_anim_pairs": (_int, [_vp, _vp, _vp, _u64, _int, _int, _vp]),
    "pg_anim_pairs_enqueue": (_int, [_vp, _vp, _vp, _u64, _int, _int, _vp]),
    "pg_anim_pairs_fetch": (_int, [_vp, _u64, _vp, _u64]),
    "pg_anim_reduce": (_int, [_vp, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _vp]),
    "pg_anim_set_batch_budget": (_int, [_vp, _u32, ctypes.c_uint64]),
    "pg_anim_set_workers": (_int, [_vp, _int]),
    "pg_anim_set_extender": (_int, [_vp, _int]),
    "pg_anim_counters": (_int, [_vp, _vp, _int]),
    "pg_anim_pair_alignments": (_int, [_vp, _i32, _i32, _vp, _u32, _P(_u32)]),
    "pg_anim_alignments_batch": (_int, [_vp, _vp, _vp, _u64, _int, _int, _vp, _P(_u64)]),
    "pg_anim_alignments_read": (_int, [_vp, _vp, _vp, _vp]),
    "pg_anib_reduce": (_int, [_vp, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pg_anib_pairs": (_int, [_vp, _vp, _vp, _u64, _u32, _vp]),
    "pg_anib_pair_rows": (_int, [_vp, _i32, _i32, _u32, _vp, _u32, _P(_u32)]),
    "pg_sketch_pairs": (_int, [_vp, _vp, _vp, _u64, _i32, _i32, ctypes.c_double, _vp]),
    "pg_profile_enable": (_int, [_vp, _int]),
    "pg_profile_config": (_int, [_vp, _u32, _u32]),
    "pg_profile_reset": (_int, [_vp]),
    "pg_profile_get": (_int, [_vp, _int, _P(ctypes.c_double), _P(_u64)]),
    "pg_kernel_name": (ctypes.c_char_p, [_int]),
}

_lib = None


class PyaniGpuError(RuntimeError):
    """Any failure reported by libpyani_gpu.so (status code in .code)."""

    def __init__(self, code, msg):
        super().__init__(f"libpyani_gpu error {code}: {msg}")
        self.code = code


def load():
    """Load the shared library and declare all prototypes; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(f"{LIB_PATH} not built — run `python -m pyani_amd.build` (needs hipcc); "
                              "pyani_amd has no CPU fallback")
        lib = ctypes.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib
