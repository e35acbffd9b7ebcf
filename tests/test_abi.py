"""CPU tier: the C-ABI shared library loads and exports every symbol include/*.h declares.
No compute calls (no GPU here); argument validation and the 'no device' refusal are exercised."""
from __future__ import annotations

import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest
import scipy.sparse as sp

from similaripy_amd import _abi, _host

ROOT = Path(__file__).resolve().parent.parent
HEADER = (ROOT / "include" / "sp_knn.h").read_text()
PREP_HEADER = (ROOT / "include" / "sp_prep.h").read_text()


def test_library_builds_and_loads():
    lib = _abi.load()
    assert lib.sp_abi_version() == 5


def test_every_declared_symbol_is_exported():
    # function declarations of the header: `<ret> name(args);` at top level
    declared = set(re.findall(r"^\s*(?:const\s+char\s*\*\s*|int64_t\s+|int\s+)(sp_[a-z0-9_]+)\s*\(", HEADER + PREP_HEADER, flags=re.M))
    assert declared, "no declarations parsed"
    assert declared == set(_abi.EXPORTED_SYMBOLS), (declared, _abi.EXPORTED_SYMBOLS)
    lib = _abi.load()
    for sym in declared:
        assert getattr(lib, sym) is not None


def test_struct_layout_matches_header_field_order():
    # every struct field of the header appears, in order, in the ctypes mirror
    body = HEADER[HEADER.index("typedef struct sp_knn_args {") + len("typedef struct sp_knn_args {"):HEADER.index("} sp_knn_args;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for stmt in body.split(";"):
        stmt = stmt.strip()
        if not stmt:
            continue
        # "const float *Xtversky, *Ytversky" / "float a1, l1" / "int64_t reserved[4]"
        first, *rest = stmt.split(",")
        names.append(re.sub(r"\[.*\]", "", first.split()[-1].lstrip("*")))
        names += [re.sub(r"\[.*\]", "", r.strip().lstrip("*")) for r in rest]
    mirror = [f[0] for f in _abi.SpKnnArgs._fields_]
    assert names == mirror


def test_transpose_struct_layout_and_refusals():
    body = PREP_HEADER[PREP_HEADER.index("typedef struct sp_csr_transpose_args {") + len("typedef struct sp_csr_transpose_args {"):PREP_HEADER.index("} sp_csr_transpose_args;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = [re.sub(r"\[.*\]", "", stmt.strip().split()[-1].lstrip("*")) for stmt in body.split(";") if stmt.strip()]
    assert names == [f[0] for f in _abi.SpCsrTransposeArgs._fields_]
    lib = _abi.load()
    a = _abi.SpCsrTransposeArgs()
    a.struct_size = 4
    assert lib.sp_csr_transpose_f32_i32(C.byref(a)) == -1             # SP_EINVAL
    a.struct_size = C.sizeof(_abi.SpCsrTransposeArgs)
    a.n_rows, a.n_cols, a.nnz = 3, 4, 10
    assert lib.sp_csr_transpose_workspace_bytes(C.byref(a)) >= 10 * 8 + 5 * 8
    indptr = np.zeros(4, dtype=np.int32)
    out = np.zeros(5, dtype=np.int32)
    a.nnz = 0
    a.indptr, a.out_indptr = indptr.ctypes.data, out.ctypes.data
    if lib.sp_device_count() == 0:
        assert lib.sp_csr_transpose_f32_i32(C.byref(a)) == -2         # SP_ENODEVICE: no CPU fallback
        assert b"no HIP device" in lib.sp_last_error()


def test_sqsums_struct_layout_and_refusals():
    body = PREP_HEADER[PREP_HEADER.index("typedef struct sp_csr_sqsums_args {") + len("typedef struct sp_csr_sqsums_args {"):PREP_HEADER.index("} sp_csr_sqsums_args;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = [re.sub(r"\[.*\]", "", stmt.strip().split()[-1].lstrip("*")) for stmt in body.split(";") if stmt.strip()]
    assert names == [f[0] for f in _abi.SpCsrSqsumsArgs._fields_]
    lib = _abi.load()
    a = _abi.SpCsrSqsumsArgs()
    a.struct_size = 4
    assert lib.sp_csr_row_sqsums_f32(C.byref(a)) == -1                # SP_EINVAL
    a.struct_size = C.sizeof(_abi.SpCsrSqsumsArgs)
    indptr = np.zeros(4, dtype=np.int32)
    a.n_rows, a.nnz, a.indptr = 3, 0, indptr.ctypes.data
    if lib.sp_device_count() == 0:
        assert lib.sp_csr_row_sqsums_f32(C.byref(a)) == -2            # SP_ENODEVICE: no CPU fallback
        with pytest.raises(_abi.HipLibraryError):
            _host.squared_norms_m1t_hip(np.ones(3, np.float32), np.array([0, 1, 3], np.int32))


def test_normalize_struct_layout_and_refusals():
    body = PREP_HEADER[PREP_HEADER.index("typedef struct sp_csr_normalize_args {") + len("typedef struct sp_csr_normalize_args {"):PREP_HEADER.index("} sp_csr_normalize_args;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for stmt in body.split(";"):
        stmt = stmt.strip()
        if not stmt:
            continue
        first, *rest = stmt.split(",")
        names.append(first.split()[-1].lstrip("*"))
        names += [r.strip().lstrip("*") for r in rest]
    assert names == [f[0] for f in _abi.SpCsrNormalizeArgs._fields_]
    lib = _abi.load()
    a = _abi.SpCsrNormalizeArgs()
    a.struct_size = 4
    assert lib.sp_csr_normalize(C.byref(a)) == -1                     # SP_EINVAL
    a.struct_size = C.sizeof(_abi.SpCsrNormalizeArgs)
    a.mode = 9
    assert lib.sp_csr_normalize(C.byref(a)) == -1 and b"bad mode" in lib.sp_last_error()
    indptr = np.zeros(4, dtype=np.int32)
    a.mode, a.n_rows, a.nnz, a.indptr = _abi.SP_NORM_L1, 3, 0, indptr.ctypes.data
    if lib.sp_device_count() == 0:
        assert lib.sp_csr_normalize(C.byref(a)) == -2                 # SP_ENODEVICE: no CPU fallback
        from similaripy_amd.normalization import normalize
        with pytest.raises(_abi.HipLibraryError):
            normalize(sp.random_array((5, 4), density=0.5, format="csr", dtype=np.float32, random_state=np.random.default_rng(0)), norm="l1")
    assert lib.sp_device_cache_trim() >= 0


def test_colsums_struct_layout_and_refusals():
    body = PREP_HEADER[PREP_HEADER.index("typedef struct sp_csr_colsums_args {") + len("typedef struct sp_csr_colsums_args {"):PREP_HEADER.index("} sp_csr_colsums_args;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = [stmt.strip().split()[-1].lstrip("*") for stmt in body.split(";") if stmt.strip()]
    assert names == [f[0] for f in _abi.SpCsrColsumsArgs._fields_]
    lib = _abi.load()
    a = _abi.SpCsrColsumsArgs()
    a.struct_size = 4
    assert lib.sp_csr_col_sums_f32(C.byref(a)) == -1                  # SP_EINVAL
    a.struct_size = C.sizeof(_abi.SpCsrColsumsArgs)
    out = np.zeros(3, dtype=np.float32)
    a.n_cols, a.nnz, a.out = 3, 0, out.ctypes.data
    if lib.sp_device_count() == 0:
        assert lib.sp_csr_col_sums_f32(C.byref(a)) == -2              # SP_ENODEVICE: no CPU fallback


def test_struct_size_is_checked():
    lib = _abi.load()
    a = _abi.SpKnnArgs()
    a.struct_size = 8
    assert lib.sp_knn_f32_i32(C.byref(a)) == -1          # SP_EINVAL
    assert b"size mismatch" in lib.sp_last_error()


def _call(**kw):
    m = sp.random_array((30, 20), density=0.2, format="csr", dtype=np.float32, random_state=np.random.default_rng(0))
    return _host.prepare(m, k=5, **kw)


def test_argument_validation_without_device():
    lib = _abi.load()
    a = _abi.SpKnnArgs()
    a.struct_size = C.sizeof(_abi.SpKnnArgs)
    a.k = 0
    assert lib.sp_knn_f32_i32(C.byref(a)) == -1 and b"k must be >= 1" in lib.sp_last_error()
    a.k = 3
    a.n_targets = 4                                      # but no pointers
    assert lib.sp_knn_f32_i32(C.byref(a)) == -1 and b"NULL" in lib.sp_last_error()


@pytest.mark.skipif(_abi.device_count() > 0, reason="only meaningful on a box without a GPU")
def test_no_cpu_fallback():
    """The product path must fail loudly when there is no HIP device."""
    assert _abi.device_count() == 0
    with pytest.raises(_abi.HipLibraryError, match="no HIP device"):
        _host.run_hip(_call())
    import similaripy_amd as sim
    m = sp.random_array((30, 20), density=0.2, format="csr", dtype=np.float32, random_state=np.random.default_rng(0))
    with pytest.raises(_abi.HipLibraryError):
        sim.cosine(m, k=3, verbose=False)


def test_workspace_query():
    lib = _abi.load()
    a = _abi.SpKnnArgs()
    a.k = 100
    a.n_targets = 0
    n = _abi.workspace_bytes(a)
    assert n >= 256
    a.k = 100000                                          # candidate buffers no longer fit LDS -> global scratch
    a.n_targets = 1000
    # pointers are not dereferenced by the query, but validate() wants them non-NULL
    dummy = np.zeros(4, dtype=np.int32)
    for f in ("targets", "m1_indptr", "m2_indptr", "cols", "values", "rows"):
        setattr(a, f, dummy.ctypes.data)
    assert _abi.workspace_bytes(a) > n


def test_abi5_multi_device_fields_are_validated():
    """sp_knn_args.n_devices / device_ids (ABI 5) and SP_FLAG_REUSE_M2_PREP: argument checks that need no device."""
    lib = _abi.load()
    assert [f[0] for f in _abi.SpKnnArgs._fields_][-3:] == ["n_devices", "_pad3", "device_ids"]
    assert _abi.SP_FLAG_REUSE_M2_PREP == int(re.search(r"#define\s+SP_FLAG_REUSE_M2_PREP\s+(\d+)u", HEADER).group(1))
    a = _abi.SpKnnArgs()
    a.struct_size = C.sizeof(_abi.SpKnnArgs)
    a.k = 3
    a.n_devices = -1
    assert lib.sp_knn_f32_i32(C.byref(a)) == -1 and b"n_devices" in lib.sp_last_error()
    a.n_devices = 65
    assert lib.sp_knn_f32_i32(C.byref(a)) == -1 and b"n_devices" in lib.sp_last_error()
    a.n_devices = 2
    if lib.sp_device_count() > 0:
        a.flags = _abi.SP_FLAG_REUSE_M2_PREP              # host mode, no workspace: refused
        assert lib.sp_knn_f32_i32(C.byref(a)) == -1 and b"SP_FLAG_REUSE_M2_PREP" in lib.sp_last_error()
        a.flags = 0
    if lib.sp_device_count() == 0:
        assert lib.sp_knn_f32_i32(C.byref(a)) == -2       # SP_ENODEVICE: several devices asked for, none there, no CPU fallback
        with pytest.raises(_abi.HipLibraryError, match="no HIP device"):
            _host.run_hip(_call(), devices=[0, 1])
    with pytest.raises(ValueError, match="devices is empty"):
        _host.run_hip(_call(), devices=[])
