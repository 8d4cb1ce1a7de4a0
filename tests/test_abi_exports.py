"""The drop-in boundary: both shared libraries load and export every symbol the headers declare,
with the struct layouts the reference's cgo side expects.  No GPU needed (nothing is called)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from aresdb_amd import abi

ROOT = abi.REPO_ROOT


def _built():
    algo, mem = abi.hip_library_paths()
    if not (os.path.exists(algo) and os.path.exists(mem)):
        import __graft_entry__ as g
        g.build()
    return algo, mem


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"^(?:CGoCallResHandle|DeviceMemoryFlags)\s+(\w+)\s*\(", src, flags=re.M)))


def test_struct_sizes_match_reference_layout():
    for struct, size in abi.ABI_SIZES.items():
        assert C.sizeof(struct) == size, struct.__name__
    assert abi.VectorPartySlice.DataType.offset == 20
    assert abi.VectorPartySlice.DefaultValue.offset == 24
    assert abi.VectorPartySlice.Length.offset == 48
    assert abi.InputVector.Type.offset == 72
    assert abi.OutputVector.Type.offset == 24
    assert abi.DimensionVector.NumDimsPerDimWidth.offset == 28
    assert abi.ForeignColumnVector.DataType.offset == 44
    assert abi.ConstantVector.IsValid.offset == 16


def test_headers_compile_as_c_and_cxx(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "ares_algorithm.h"\n#include "ares_memory.h"\nint main(void){return 0;}\n')
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c11", "-I", inc, "-c", str(src), "-o", str(tmp_path / "c.o")])
    subprocess.check_call(["g++", "-std=c++17", "-x", "c++", "-I", inc, "-c", str(src), "-o", str(tmp_path / "x.o")])


def test_libalgorithm_exports_every_declared_symbol():
    algo, _ = _built()
    names = _declared("ares_algorithm.h")
    assert len(names) == 14 and set(names) == set(abi.ALGORITHM_SYMBOLS)
    lib = C.CDLL(algo, mode=os.RTLD_NOW | os.RTLD_LOCAL)
    for n in names:
        assert getattr(lib, n) is not None


def test_libmem_exports_every_declared_symbol():
    _, mem = _built()
    names = _declared("ares_memory.h")
    assert set(names) == set(abi.MEMORY_SYMBOLS) | {"GetFlags"}
    lib = C.CDLL(mem, mode=os.RTLD_NOW | os.RTLD_LOCAL)
    for n in names:
        assert getattr(lib, n) is not None


def test_extension_symbols_are_exported_by_the_library_that_declares_them():
    """include/ares_extensions.h: profiler, deferral hooks between the two libraries, fused entry point."""
    algo, mem = _built()
    src = open(os.path.join(ROOT, "include", "ares_extensions.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    declared = set(re.findall(r"^(?:void|size_t|CGoCallResHandle)\s+(\w+)\s*\(", src, flags=re.M))
    in_mem = {n for n in declared if n.startswith("AresMem")}
    assert {"AresMemSetFlushHook", "AresMemSetDeferralHooks", "AresMemReleaseHeld", "AresMemSetAuxHooks",
            "AresMemTrimCache", "AresMemStats", "AresMemNoteWrite", "AresMemEnableWriteTracking",
            "AresMemDriverCalls"} <= in_mem
    assert {"AresFlushDeferred", "AresProfilerEnable", "AresProfilerReport", "AresReloadEnv",
            "AresFusedFilterHashReduce"} <= declared
    la, lm = C.CDLL(algo, mode=os.RTLD_NOW | os.RTLD_LOCAL), C.CDLL(mem, mode=os.RTLD_NOW | os.RTLD_LOCAL)
    for n in declared:
        assert getattr(lm if n in in_mem else la, n) is not None


def test_libmem_exports_nothing_but_the_abi_and_its_declared_extensions():
    """The reference's libmem has 23 symbols (cgoutils/memory.h); ours adds the AresMem* extensions of
    include/ares_extensions.h and no stray internals."""
    _, mem = _built()
    out = subprocess.check_output(["nm", "-D", "--defined-only", mem], text=True)
    exported = {ln.split()[-1] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] in "TW"}
    exported = {n for n in exported if not n.startswith("_")}  # (_init/_fini, C++ runtime weak symbols)
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "ares_extensions.h")).read(), flags=re.S)
    ext = {n for n in re.findall(r"^(?:void|size_t|CGoCallResHandle)\s+(\w+)\s*\(", src, flags=re.M) if n.startswith("AresMem")}
    abi_names = set(_declared("ares_memory.h"))
    assert len(abi_names) == 23
    assert exported == abi_names | ext, (sorted(exported - abi_names - ext), sorted((abi_names | ext) - exported))


def test_libalgorithm_does_not_depend_on_libmem_or_oracle():
    """Either library can be swapped on its own, and the product never links test infrastructure."""
    algo, mem = _built()
    for lib in (algo, mem):
        out = subprocess.check_output(["readelf", "-d", lib], text=True)
        needed = re.findall(r"NEEDED.*\[(.*?)\]", out)
        assert not any("oracle" in n or "libmem" in n for n in needed), needed
        assert any("amdhip64" in n for n in needed), needed


def test_product_backend_fails_loudly_without_libraries(tmp_path, monkeypatch):
    monkeypatch.setattr(abi, "LIB_DIR", str(tmp_path))
    with pytest.raises(FileNotFoundError):
        abi.load_hip_backend()
