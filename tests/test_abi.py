"""The C-ABI library loads and exports every symbol include/mslesions3d_hip.h declares, and the ctypes
signatures used by the Python host agree with the header, parameter by parameter.  No compute (CPU only)."""
import ctypes
import os
import re

import pytest

from mslesions3d_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mslesions3d_hip.h")


def _prototypes():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|size_t)\s+(msl_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        ret, name, params = m.group(1), m.group(2), " ".join(m.group(3).split())
        plist = [] if params in ("void", "") else [p.strip() for p in params.split(",")]
        protos[name] = (ret, plist)
    return protos


def _ctype_of(param):
    if "*" in param:
        return ctypes.c_void_p
    base = param.rsplit(" ", 1)[0].replace("const ", "").strip()
    return {"int": ctypes.c_int, "unsigned int": ctypes.c_uint, "float": ctypes.c_float, "double": ctypes.c_double, "size_t": ctypes.c_size_t,
            "long long": ctypes.c_longlong}[base]


def test_header_declares_something():
    assert len(_prototypes()) >= 40


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    for name in _prototypes():
        assert hasattr(lib, name), f"{name} declared in the header but not exported by {_lib.LIB_PATH}"


def test_python_binding_matches_header():
    protos = _prototypes()
    assert sorted(protos) == _lib.exported_names()
    for name, (ret, plist) in protos.items():
        res, args = _lib._SIGNATURES[name]
        assert res is (ctypes.c_int if ret == "int" else ctypes.c_size_t), name
        assert len(args) == len(plist), f"{name}: header has {len(plist)} parameters, binding {len(args)}"
        for i, (p, a) in enumerate(zip(plist, args)):
            assert _ctype_of(p) is a, f"{name} parameter {i} ({p!r}) bound as {a}"


def test_pure_host_entry_points():
    lib = _lib.load()
    assert lib.msl_abi_version() == 1
    # shape planning helpers are host-only arithmetic: config A (128^3, batch 4) of BASELINE.json
    assert lib.msl_stem_conv_fwd_num_partials(4, 64, 64, 64) == 4 * 256
    assert lib.msl_dwconv_fwd_variant(4, 32, 64, 64, 64, 2) == 3   # L1: planes in registers, eight waves per plane
    assert lib.msl_dwconv_fwd_variant(2, 32, 96, 96, 96, 2) == 1   # 192^3 inference, L1: streamed planes
    assert lib.msl_dwconv_fwd_variant(4, 64, 32, 32, 32, 2) == 3   # L2: planes in registers, two waves per plane
    assert lib.msl_dwconv_fwd_variant(2, 128, 24, 24, 24, 1) == 2  # 192^3 inference, L3: LDS-resident slab
    assert lib.msl_dwconv_fwd_variant(4, 128, 16, 16, 16, 2) == 3  # L4: planes in registers
    assert lib.msl_dwconv_fwd_variant(4, 128, 16, 16, 16, 1) == 3  # L3: planes in registers, one wave per slab
    assert lib.msl_dwconv_fwd_variant(2, 512, 2, 2, 2, 1) == 0     # 64^3 config tail: generic kernel
    assert lib.msl_pwconv_fwd_num_partials(4, 32, 64, 32768) == 4 * 128   # two 32-column tiles per wave
    assert lib.msl_pwconv_fwd_num_partials(4, 512, 512, 64) == 4 * 2      # K-split wave form: 32-column tiles
    assert lib.msl_head_packed_weight_elems(128, 2) == 128 // 4 * 27 * 64
    # statistic / weight-gradient partial counts of the depthwise wave kernels (config A): N x depth slabs (x workgroups
    # of a split plane)
    assert lib.msl_dwconv_fwd_num_partials(4, 32, 64, 64, 64, 2) == 4 * 8 * 2    # block 1: 8 slabs, two 4-wave workgroups
    assert lib.msl_dwconv_fwd_num_partials(4, 64, 32, 32, 32, 2) == 4 * 4        # block 2: one 2-wave workgroup per plane
    assert lib.msl_dwconv_fwd_num_partials(4, 128, 16, 16, 16, 1) == 4 * 2
    assert lib.msl_dwconv_fwd_num_partials(4, 512, 4, 4, 4, 1) == 4 * 2
    assert lib.msl_dwconv_bwd_weight_num_partials(4, 64, 32, 32, 32, 2) == 4 * 4
    assert lib.msl_dwconv_bwd_weight_num_partials(4, 256, 8, 8, 8, 1) == 4 * 2
    assert lib.msl_multibox_loss_var_workspace_bytes(4, 9344) == 4 * 9344 * 4 + 4 * 8


def test_native_program_runner_covers_the_abi():
    """Every int-returning entry point without out-parameters has a trampoline in the generated runner."""
    lib = _lib.load()
    skipped = {"msl_run_program", "msl_program_fn_id", "msl_event_create", "msl_event_create_device",
               "msl_bn_finalize_table_set",
               "msl_grad_reduce_table_set", "msl_run_program_mt",
               "msl_event_create_timed", "msl_event_elapsed_ms"}
    for name, (ret, _) in _prototypes().items():
        fid = lib.msl_program_fn_id(name.encode())
        assert (fid >= 0) == (ret == "int" and name not in skipped), name
    # host-only call through the runner: msl_abi_version() == 1 means rc 1 is reported as the failing code
    import ctypes
    ids = (ctypes.c_int * 1)(lib.msl_program_fn_id(b"msl_abi_version"))
    slots = (ctypes.c_ulonglong * 28)()
    failed = ctypes.c_int(-1)
    assert lib.msl_run_program(ids, slots, 28, 1, ctypes.byref(failed)) == 1 and failed.value == 0


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmsl3d_hip.so")
    with pytest.raises(_lib.HipKernelError, match="no CPU fallback"):
        _lib.load()
