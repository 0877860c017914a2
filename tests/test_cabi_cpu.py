"""CPU: the C-ABI library loads and exports every symbol include/cradle_hip.h declares."""
import os
import re

from cradle_vae_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "cradle_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cradle_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = hip.load()
    names = _declared_symbols()
    assert len(names) >= 8
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/cradle_hip.h but not exported"
        assert name in hip.PROTOTYPES, f"{name} has no ctypes prototype in cradle_vae_amd/hip.py"
    for name in hip.PROTOTYPES:
        assert name in names, f"{name} bound in hip.py but not declared in the header"
    assert lib.cradle_abi_version() == hip.ABI_VERSION


def test_descriptor_layouts_agree_between_header_library_and_binding():
    """include/cradle_hip.h is the one definition of the descriptor structs (the .hip files include it); the
    ctypes mirrors in hip.py must have the same size and field offsets as the compiled library reports, and as
    many fields as the header declares."""
    lib = hip.load()
    text = open(os.path.join(ROOT, "include", "cradle_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for cname, cls in (("cradle_mlp_ptrs", hip.MlpPtrs), ("cradle_gemm_desc", hip.GemmDesc),
                       ("cradle_step_desc", hip.StepDesc), ("cradle_batch_desc", hip.BatchDesc)):
        mine, theirs = hip.struct_layout(cls), hip.library_struct_layout(lib, cname)
        assert mine == theirs, (cname, mine, theirs)
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), text, flags=re.S).group(1)
        declared = 0
        for stmt in body.split(";"):
            stmt = stmt.strip()
            if stmt:
                declared += stmt.count(",") + 1
        assert declared == len(cls._fields_), (cname, declared, len(cls._fields_))
    for src in ("step.hip", "gemm.hip"):  # no second definition anywhere in the sources
        code = open(os.path.join(ROOT, "cradle_vae_amd", "csrc", src)).read()
        assert "struct cradle_step_desc {" not in code and "struct cradle_gemm_desc {" not in code
    assert lib.cradle_struct_layout(b"nonsense", (hip.ctypes.c_int64 * 4)(), 4) == -1


def test_host_side_helpers_need_no_gpu():
    lib = hip.load()
    assert lib.cradle_nb_particle_split(5, 512) == 1
    assert lib.cradle_nb_particle_split(100, 128) == 4
    assert lib.cradle_nb_workspace_floats(5, 512, 5000) == 512 * 5000


def test_library_is_newer_than_its_sources():
    """The .so travels to the GPU box prebuilt: a stale build would silently test old code."""
    import glob

    so = os.path.getmtime(hip.lib_path())
    for src in glob.glob(os.path.join(ROOT, "cradle_vae_amd", "csrc", "*.h*")):
        assert os.path.getmtime(src) <= so, f"{src} is newer than libcradle_hip.so: run make -C cradle_vae_amd/csrc"
