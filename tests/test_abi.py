"""CPU-side checks of the C-ABI library: it builds, loads, and exports what include/h3d.h declares."""
import ctypes
import importlib
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build = importlib.import_module("3dhumangan_amd._build")
    path = build.build_lib()
    return ctypes.CDLL(path)


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "h3d.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(h3d_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(lib):
    names = declared_symbols()
    assert "h3d_ray_integrate" in names and "h3d_version" in names
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/h3d.h but not exported by libh3d.so: {missing}"


def test_version_and_error_string(lib):
    lib.h3d_version.restype = ctypes.c_int
    assert lib.h3d_version() == 101
    lib.h3d_last_error.restype = ctypes.c_char_p
    assert isinstance(lib.h3d_last_error(), bytes)


def test_argument_validation_needs_no_gpu(lib):
    """Bad arguments are rejected before any HIP call."""
    lib.h3d_ray_integrate.restype = ctypes.c_int
    lib.h3d_ray_integrate.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int64] + [ctypes.c_int] * 5 + [ctypes.c_void_p]
    rc = lib.h3d_ray_integrate(None, None, None, None, None, None, 4, 8, 3, 0, 0, 0, None)
    assert rc == -1
    assert b"null pointer" in lib.h3d_last_error()


def test_product_has_no_cpu_fallback():
    import torch
    vr = importlib.import_module("3dhumangan_amd.lib.generators.volume_rendering")
    h3dlib = importlib.import_module("3dhumangan_amd._lib")
    with pytest.raises(h3dlib.H3DError):
        vr.ray_integration(torch.zeros(1, 2, 4, 5), torch.zeros(1, 2, 4, 1), clamp_mode="relu", noise_std=0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "3dhumangan_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "h3d_oracle" not in src and "import oracle" not in src and "from oracle" not in src, f


def test_every_m0_write_in_the_library_is_a_weight_ring_dma(lib, tmp_path):
    """WeightRing (csrc/x3_common.hpp) writes M0 once per stage and issues the stage's remaining LDS-DMA pieces sections later
    WITHOUT rewriting it.  That is only sound while nothing the compiler generates touches M0: disassemble every gfx950 code
    object of the library and require that each instruction naming m0 is `s_mov_b32 m0, sN` directly followed by a
    global_load_lds (ours).  A compiler-generated M0 user (s_movrel, readlane by M0, ...) would show up here."""
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    build = importlib.import_module("3dhumangan_amd._build")
    so = shutil.copy(build.LIB, str(tmp_path / "libh3d.so"))
    subprocess.run([objdump, "--offloading", so], cwd=str(tmp_path), capture_output=True, check=True)
    objs = [str(tmp_path / f) for f in os.listdir(tmp_path) if "gfx950" in f]
    assert objs, "no gfx950 code objects in the library"
    n_writes = 0
    for o in objs:
        text = subprocess.run([objdump, "-d", o], capture_output=True, text=True, check=True).stdout.split("\n")
        for i, line in enumerate(text):
            if re.search(r"\bm0\b", line.split("//")[0]):
                assert re.search(r"s_mov_b32 m0, (s\d+|vcc_lo|vcc_hi|ttmp\d+)\b", line), f"unexpected M0 user: {line.strip()}"
                assert "global_load_lds_dwordx4" in text[i + 1], f"M0 write not followed by an LDS-DMA: {text[i + 1].strip()}"
                n_writes += 1
    assert n_writes > 100        # the register engines' rings are in there


def test_no_flat_memory_instruction_in_the_ring_kernels(lib, tmp_path):
    """Round 5: a FLAT load / store (a generic pointer, e.g. one laundered through an empty asm) keeps LLVM's wait-count pass in
    its "pending FLAT" state until a vmcnt(0) -- which the weight-ring kernels never execute -- and while it is pending EVERY LDS
    wait is emitted as s_waitcnt lgkmcnt(0): the synthesis engines drained their whole fragment look-ahead at every table read
    because of three FLAT stores of the RGB image.  No kernel that runs an LDS-DMA ring may contain a FLAT memory instruction."""
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    build = importlib.import_module("3dhumangan_amd._build")
    so = shutil.copy(build.LIB, str(tmp_path / "libh3d.so"))
    subprocess.run([objdump, "--offloading", so], cwd=str(tmp_path), capture_output=True, check=True)
    objs = [str(tmp_path / f) for f in os.listdir(tmp_path) if "gfx950" in f]
    ring_kernels = 0
    for o in objs:
        text = subprocess.run([objdump, "-d", o], capture_output=True, text=True, check=True).stdout
        for body in re.split(r"\n(?=[0-9a-f]+ <)", text):                     # one chunk per symbol
            if "global_load_lds_dwordx4" not in body:
                continue
            ring_kernels += 1
            name = body.split("\n", 1)[0]
            bad = [ln.strip() for ln in body.split("\n") if re.search(r"\bflat_(load|store|atomic)", ln)]
            assert not bad, f"FLAT memory instruction in a weight-ring kernel {name}: {bad[:3]}"
    assert ring_kernels >= 20


def test_per_file_build_flags_name_existing_sources():
    """_build.FILE_FLAGS (per-file compiler options: round 5 compiles field_x3.hip without LLVM's post-RA scheduler) must name
    files that exist, or the option silently stops applying when a file is renamed."""
    import importlib
    import os
    b = importlib.import_module("3dhumangan_amd._build")
    names = {os.path.basename(s) for s in b.sources()}
    assert b.FILE_FLAGS and set(b.FILE_FLAGS) <= names
    assert b.FILE_FLAGS["field_x3.hip"] == ["-mllvm", "-enable-post-misched=false"]


def test_every_barrier_in_the_ring_kernels_follows_a_wait_for_the_lds_reads(lib, tmp_path):
    """Round 6: the weight ring's write-after-read safety is by construction -- a wave arrives at a stage barrier only after its LDS
    reads have returned (s_waitcnt .. lgkmcnt(0)), and refills are issued behind the barrier.  (The distance argument of rounds 2-5
    failed in conv_x3.hip with several workgroups per CU.)  Disassemble the library: in every kernel that runs an LDS-DMA ring, each
    s_barrier must be preceded -- scalar bookkeeping aside -- by an s_waitcnt with lgkmcnt(0)."""
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    build = importlib.import_module("3dhumangan_amd._build")
    so = shutil.copy(build.LIB, str(tmp_path / "libh3d.so"))
    subprocess.run([objdump, "--offloading", so], cwd=str(tmp_path), capture_output=True, check=True)
    objs = [str(tmp_path / f) for f in os.listdir(tmp_path) if "gfx950" in f]
    barriers = 0
    for o in objs:
        text = subprocess.run([objdump, "-d", o], capture_output=True, text=True, check=True).stdout
        for body in re.split(r"\n(?=[0-9a-f]+ <)", text):
            if "global_load_lds_dwordx4" not in body:
                continue
            name = body.split("\n", 1)[0]
            lines = [ln.split("//")[0].strip() for ln in body.split("\n")]
            lines = [ln for ln in lines if ln]
            other = []
            for i, ln in enumerate(lines):
                if not ln.startswith("s_barrier"):
                    continue
                barriers += 1
                j = i - 1
                while j >= 0 and re.match(r"s_(cmp|cbranch|and|or|mov|add|sub|lshl|cselect|nop)", lines[j]) and i - j < 8:
                    j -= 1          # a uniform branch / address arithmetic between the wait and the barrier
                if not (lines[j].startswith("s_waitcnt") and "lgkmcnt(0)" in lines[j]):
                    other.append(lines[j])
            # the only barriers of another shape: the two __syncthreads of conv_x3_kernel's moments epilogue (they fence its staging
            # area after the ring has drained, not a ring stage; the compiler hoists their wait above the divergent code in front)
            assert not other or ("conv_x3_kernel" in name and len(other) <= 2), f"{name}: barrier after {other[:3]}"
    assert barriers > 1000
