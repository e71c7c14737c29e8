"""ctypes binding of liblrm_accel.so (include/lrm_accel.h, include/lrm_index_host.h).

This is what a Python host would bind; the structs mirror the reference's own
(accaln.h / alnmain.h / fmidx.h / lchash.h / histo.h).  There is no fallback: if the
library is missing the import of this module fails loudly.
"""
import ctypes as C
import os

from . import _build

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
i32p = C.POINTER(C.c_int32)


class Entry(C.Structure):              # histo.h:21-23
    _fields_ = [("key", C.c_uint64), ("val", C.c_uint64), ("bucket", C.c_uint64)]


class Params(C.Structure):             # alnmain.h:10-13
    _fields_ = [("batch_size", C.c_uint64), ("seed_len", C.c_uint32), ("thres", C.c_uint32)]


class DnaFmi(C.Structure):             # fmidx.h:16-21
    _fields_ = [("length", C.c_uint64), ("o_len", C.c_uint64), ("csa_len", C.c_uint64),
                ("c", u64p), ("o", u64p), ("csa", u64p),
                ("o_ratio", C.c_int), ("csa_ratio", C.c_int), ("bwt", C.c_void_p)]


class LcHash(C.Structure):             # lchash.h:16-20
    _fields_ = [("lc", u64p), ("len", C.c_uint64), ("hlen", C.c_int)]


class Ui40(C.Structure):               # sa_use.h:17-20 (8 bytes in RAM)
    _fields_ = [("low", C.c_uint32), ("high", C.c_uint8)]


class SaMem(C.Structure):              # fmidx.h:23-26
    _fields_ = [("start", C.c_uint64), ("len", C.c_uint64), ("mem", C.POINTER(Ui40))]


class MtaEntry(C.Structure):           # accaln.h:67-71 flattened
    _fields_ = [("name_len", C.c_uint64), ("name", C.c_char_p), ("name_own", C.c_int),
                ("offset", C.c_uint64), ("seq_len", C.c_size_t)]


class SeqMeta(C.Structure):            # alnmain.c:143-148
    _fields_ = [("loc", C.c_uint64), ("off", C.c_uint64), ("seq_id", C.c_int32), ("strand", C.c_uint8)]


class Cigar(C.Structure):              # gact cigar (mutils.c:97-103)
    _fields_ = [("cigar", u8p), ("n_cigar_op", C.c_int), ("score", C.c_int)]


class GactParams(C.Structure):
    _fields_ = [("T", C.c_int), ("O", C.c_int), ("W", C.c_int)]


class IndexOptions(C.Structure):       # lrm_index_options
    _fields_ = [("struct_size", C.c_uint32), ("sa_sampled", C.c_int32), ("lc_long", C.c_int32),
                ("lc_long_max", C.c_int32), ("lc_pair", C.c_int32), ("lcx_threshold", C.c_uint32),
                ("lc_entry_bytes", C.c_uint32), ("lc_core", C.c_int32), ("lc_count_bits", C.c_uint32),
                ("seed_table", C.c_int32), ("seed_table_len", C.c_uint32), ("seed_table_share", C.c_uint32),
                ("seed_table_bits", C.c_uint32), ("seed_table_count_bits", C.c_uint32), ("reserved", C.c_uint32 * 2)]


class IndexTables(C.Structure):        # lrm_index_tables
    _fields_ = [("lc_long", C.c_int32), ("lc_pair", C.c_int32), ("lc_entry_bytes", C.c_int32), ("lc_core", C.c_int32),
                ("seed_table_len", C.c_int32), ("seed_table_share", C.c_int32), ("seed_table_bits", C.c_int32),
                ("seed_table_slot_bytes", C.c_int32), ("seed_table_count_bits", C.c_int32), ("reserved0", C.c_int32),
                ("seed_table_side_entries", C.c_uint64), ("derived_bytes", C.c_uint64), ("reserved", C.c_uint64 * 4)]


class MapOptions(C.Structure):         # lrm_map_options
    _fields_ = [("struct_size", C.c_uint32), ("dense_results", C.c_int32), ("gact_impl", C.c_int32),
                ("seed_rounds", C.c_int32), ("vote_exact_only", C.c_int32), ("slice_reads", C.c_uint32),
                ("sub_batches", C.c_uint32), ("group_subs", C.c_uint32), ("bs_waves", C.c_uint32),
                ("cigar_text", C.c_uint32), ("copy_threads", C.c_uint32), ("keep_reads", C.c_uint32), ("reserved", C.c_uint32 * 7)]


class Stats(C.Structure):
    _fields_ = [("vote_tier2_items", C.c_uint64), ("vote_tier3_items", C.c_uint64),
                ("reads_decided_phase0", C.c_uint64), ("gact_tiles", C.c_uint64), ("seeds_evaluated", C.c_uint64),
                ("seed_table_lookups", C.c_uint64), ("seed_rank_requests", C.c_uint64), ("vote_redo_items", C.c_uint64)]


class ReadBatch(C.Structure):          # lrm_io_host.h
    _fields_ = [("n", C.c_uint64), ("stride", C.c_uint64), ("max_len", C.c_uint32), ("seqs", C.c_void_p),
                ("lens", u32p), ("names", C.POINTER(C.c_char_p)), ("quals", C.POINTER(C.c_char_p)),
                ("name_arena", C.c_void_p), ("qual_arena", C.c_void_p), ("seqs_borrowed", C.c_int)]


class HostIndex(C.Structure):          # lrm_index_host.h
    _fields_ = [("fmi", DnaFmi), ("lch", LcHash), ("sa", SaMem), ("content", C.c_void_p),
                ("con_len", C.c_uint64), ("mta", C.POINTER(MtaEntry)), ("mta_len", C.c_int)]


assert C.sizeof(Ui40) == 8 and C.sizeof(Entry) == 24 and C.sizeof(SeqMeta) == 24

# every symbol include/*.h declares: (restype, argtypes)
SYMBOLS = {
    "lrm_last_error": (C.c_char_p, []),
    "lrm_abi_version": (C.c_int, []),
    "lrm_device_count": (C.c_int, []),
    "lrm_index_blob_bytes": (C.c_uint64, [C.c_uint64, C.c_int, C.c_int]),
    "lrm_index_pack_blob": (C.c_int, [C.POINTER(DnaFmi), C.POINTER(LcHash), C.POINTER(SaMem), C.c_void_p,
                                      C.c_uint64, C.POINTER(MtaEntry), C.c_int, C.c_void_p, C.c_uint64]),
    "lrm_index_upload": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(DnaFmi), C.POINTER(LcHash),
                                   C.POINTER(SaMem), C.c_void_p, C.c_uint64, C.POINTER(MtaEntry), C.c_int, C.c_int]),
    "lrm_index_pack_device": (C.c_int, [C.POINTER(DnaFmi), C.POINTER(LcHash), C.POINTER(SaMem), C.c_void_p,
                                        C.c_uint64, C.POINTER(MtaEntry), C.c_int, C.c_void_p, C.c_uint64, C.c_int]),
    "lrm_index_options_init": (None, [C.POINTER(IndexOptions)]),
    "lrm_map_options_init": (None, [C.POINTER(MapOptions)]),
    "lrm_index_blob_bytes_opt": (C.c_uint64, [C.c_uint64, C.c_int, C.c_int, C.POINTER(IndexOptions)]),
    "lrm_index_pack_blob_opt": (C.c_int, [C.POINTER(DnaFmi), C.POINTER(LcHash), C.POINTER(SaMem), C.c_void_p,
                                          C.c_uint64, C.POINTER(MtaEntry), C.c_int, C.c_void_p, C.c_uint64,
                                          C.POINTER(IndexOptions)]),
    "lrm_index_pack_device_opt": (C.c_int, [C.POINTER(DnaFmi), C.POINTER(LcHash), C.POINTER(SaMem), C.c_void_p,
                                            C.c_uint64, C.POINTER(MtaEntry), C.c_int, C.c_void_p, C.c_uint64, C.c_int,
                                            C.POINTER(IndexOptions)]),
    "lrm_index_upload_opt": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(DnaFmi), C.POINTER(LcHash),
                                       C.POINTER(SaMem), C.c_void_p, C.c_uint64, C.POINTER(MtaEntry), C.c_int,
                                       C.POINTER(C.c_int), C.c_int, C.POINTER(IndexOptions)]),
    "lrm_index_adopt_device_opt": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint64, C.c_int,
                                             C.POINTER(IndexOptions)]),
    "lrm_index_upload_blob_opt": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint64, C.c_int,
                                            C.POINTER(IndexOptions)]),
    "lrm_index_set_map_options": (C.c_int, [C.c_void_p, C.POINTER(MapOptions)]),
    "lrm_index_get_tables": (C.c_int, [C.c_void_p, C.POINTER(IndexTables)]),
    "lrm_map_batch_submit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, Params, GactParams,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.POINTER(MapOptions), C.POINTER(C.c_void_p)]),
    "lrm_map_batch_wait": (C.c_int, [C.c_void_p]),
    "lrm_debug_reload_env": (C.c_int, [C.c_void_p]),
    "lrm_debug_set_vote_limits": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "lrm_debug_gact_impl": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, GactParams, C.c_int, C.c_void_p,
                                      C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]),
    "lrm_index_adopt_device": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint64, C.c_int]),
    "lrm_index_upload_blob": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint64, C.c_int]),
    "lrm_index_free": (None, [C.c_void_p]),
    "lrm_seed_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, Params, C.c_void_p]),
    "lrm_extend_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p,
                                   GactParams, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                   C.c_void_p]),
    "lrm_map_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, Params, GactParams,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lrm_host_alloc": (C.c_void_p, [C.c_uint64]),
    "lrm_host_free": (None, [C.c_void_p]),
    "lrm_host_register": (C.c_int, [C.c_void_p, C.c_uint64]),
    "lrm_host_unregister": (C.c_int, [C.c_void_p]),
    "lrm_pair_end": (C.c_int, [C.c_int, C.c_void_p]),
    "lrm_index_upload_multi": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(DnaFmi), C.POINTER(LcHash),
                                         C.POINTER(SaMem), C.c_void_p, C.c_uint64, C.POINTER(MtaEntry), C.c_int,
                                         C.POINTER(C.c_int), C.c_int]),
    "lrm_index_replicas": (C.c_int, [C.c_void_p]),
    "lrm_index_replica": (C.c_void_p, [C.c_void_p, C.c_int]),
    "lrm_result_flags": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lrm_workspace_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32,
                                       C.c_uint32]),
    "lrm_workspace_free": (None, [C.c_void_p]),
    "lrm_workspace_bytes": (C.c_uint64, [C.c_void_p]),
    "lrm_seed_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                     C.c_uint32, Params, C.c_void_p, C.c_void_p]),
    "lrm_extend_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                       C.c_uint32, C.c_void_p, GactParams, C.c_void_p, C.c_uint64, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lrm_workspace_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats), C.c_void_p]),
    "lrm_workspace_set_counting": (C.c_int, [C.c_void_p, C.c_int]),
    "lrm_workspace_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "lrm_workspace_timing": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lrm_kernel_name": (C.c_char_p, [C.c_int]),
    "lrm_debug_seed_search": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, u64p]),
    "lrm_debug_rccl_selftest": (C.c_int, [C.c_int, C.c_uint64]),
    "lrm_debug_gact": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, GactParams, C.c_void_p,
                                 C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]),
    # lrm_index_host.h
    "lrm_cat_from_seqs": (C.c_int, [C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), u64p, C.c_int, C.c_uint64,
                                    C.POINTER(C.c_void_p), u64p, C.POINTER(C.POINTER(MtaEntry))]),
    "lrm_sa_build": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p]),
    "lrm_host_index_build": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(MtaEntry), C.c_int, C.c_int, C.c_int,
                                       C.POINTER(HostIndex)]),
    "lrm_host_index_free": (None, [C.POINTER(HostIndex)]),
    "lrm_host_index_write": (C.c_int, [C.POINTER(HostIndex), C.c_char_p]),
    "lrm_host_index_read": (C.c_int, [C.c_char_p, C.POINTER(HostIndex)]),
    "lrm_fmi_write": (C.c_int, [C.POINTER(DnaFmi), C.c_char_p]),
    "lrm_fmi_read": (C.c_int, [C.POINTER(DnaFmi), C.c_char_p]),
    "lrm_lc_write": (C.c_int, [C.c_char_p, C.POINTER(LcHash)]),
    "lrm_lc_read": (C.c_int, [C.c_char_p, C.POINTER(LcHash)]),
    "lrm_sa5_write": (C.c_int, [C.c_char_p, C.c_void_p, C.c_uint64]),
    "lrm_sa5_read": (C.c_int64, [C.c_char_p, C.c_void_p, C.c_uint64]),
    "lrm_mta_write": (C.c_int, [C.c_char_p, C.POINTER(MtaEntry), C.c_int]),
    "lrm_mta_read": (C.c_int, [C.c_char_p, C.POINTER(C.POINTER(MtaEntry))]),
    "lrm_mta_free": (None, [C.POINTER(MtaEntry), C.c_int]),
    "lrm_accidx": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_uint64]),
    # lrm_io_host.h
    "lrm_reader_open": (C.c_int, [C.POINTER(C.c_void_p), C.c_char_p]),
    "lrm_reader_next": (C.c_int64, [C.c_void_p, C.c_uint64, C.POINTER(ReadBatch)]),
    "lrm_reader_next_into": (C.c_int64, [C.c_void_p, C.c_uint64, C.POINTER(ReadBatch), C.c_void_p, C.c_uint64]),
    "lrm_read_batch_free": (None, [C.POINTER(ReadBatch)]),
    "lrm_reader_close": (None, [C.c_void_p]),
    "lrm_parse_cigar": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "lrm_sam_header": (C.c_void_p, [C.POINTER(MtaEntry), C.c_int, C.c_long, u64p]),
    "lrm_sam_format": (C.c_void_p, [C.POINTER(ReadBatch), C.POINTER(MtaEntry), C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_uint64, u64p]),
    "lrm_free": (None, [C.c_void_p]),
    "lrm_accaln": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, Params, GactParams, C.c_int, C.c_long, u64p, u64p]),
}


def _load():
    # torch bundles its own libamdhip64.so.7 (+ HSA runtime).  Two HIP runtimes in one process do
    # not both get the GPU, and the dynamic linker keys on the SONAME: whichever is loaded first
    # serves both.  torch.distributed / torch tensors are this package's device plumbing, so let
    # torch's runtime win: import it before liblrm_accel.so pulls in /opt/rocm's.
    try:
        import torch  # noqa: F401
    except Exception:      # torch absent: the system runtime is the only one
        pass
    path = _build.ACCEL_LIB
    if not os.path.exists(path):
        path = _build.build_accel()
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def index_options(**kw):
    """lrm_index_options with the automatic choices, then the given fields (None = automatic)."""
    o = IndexOptions()
    lib.lrm_index_options_init(C.byref(o))
    for k, v in kw.items():
        if v is not None:
            setattr(o, k, v)          # AttributeError for a field the struct does not have
    return o


def map_options(**kw):
    o = MapOptions()
    lib.lrm_map_options_init(C.byref(o))
    for k, v in kw.items():
        if v is not None:
            getattr(o, k)             # AttributeError for a field the struct does not have
            setattr(o, k, v)
    return o


class LrmError(RuntimeError):
    pass


def check(rc, what=""):
    if rc < 0:
        raise LrmError("%s: %s" % (what or "liblrm_accel", lib.lrm_last_error().decode(errors="replace")))
    return rc
