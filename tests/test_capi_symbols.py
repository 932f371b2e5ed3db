"""The C-ABI library loads on a CPU-only host and exports every symbol include/mbavo.h declares.
No compute call is made here (there is no GPU and the product has no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mbavo.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mbavo_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_lists_agree(mbavo):
    assert _declared_symbols() == sorted(mbavo.capi.SYMBOLS)


def test_library_exports_every_declared_symbol(mbavo):
    lib = C.CDLL(mbavo.LIB_PATH)
    missing = [s for s in _declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    assert b"gfx950" in mbavo.load().mbavo_version()
    assert mbavo.load().mbavo_packed_len(2) == 91 and mbavo.load().mbavo_packed_len(4) == 325


def test_cxx_api_symbols_exported(mbavo):
    """The reference's C++ entry points (namespace SLAM::VO) are exported with C++ linkage."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", "-C", mbavo.LIB_PATH], capture_output=True, text=True).stdout
    for name in ["SLAM::VO::evaluate_cost_hessian_gradient(", "SLAM::VO::compute_virtual_camera_poses(",
                 "SLAM::VO::compute_local_patches_xy(", "SLAM::VO::compute_pixel_jacobian_residual(",
                 "SLAM::VO::compute_patch_cost_gradient_hessian(", "SLAM::VO::compute_frame_cost_gradient_hessian(",
                 "SLAM::VO::merge_hessian_gradient_cost(", "SLAM::VO::solve_normal_equation(",
                 "SLAM::VO::initialize_shared_cuda_storages(", "SLAM::VO::free_shared_cuda_storages("]:
        assert name in out, name


def test_no_cpu_fallback(mbavo):
    """Without a HIP device the context cannot be created: the product path never routes to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    assert mbavo.load().mbavo_create(C.byref(h), 0) == -3  # MBAVO_E_NODEVICE
    with pytest.raises(RuntimeError):
        mbavo.capi.Context(0)


def test_product_does_not_reference_oracle():
    """Nothing under mba-vo_amd/ or include/ may import, link or open anything under oracle/."""
    bad = []
    for base in ("mba-vo_amd", "include"):
        for dp_, _, files in os.walk(os.path.join(ROOT, base)):
            if "build" in dp_:
                continue
            for f in files:
                if f.endswith((".so", ".o", ".pyc")):
                    continue
                txt = open(os.path.join(dp_, f), errors="ignore").read()
                if re.search(r"oracle[/.]|mbavo_oracle|orc_[a-z]", txt) and f not in ("__init__.py",):
                    bad.append(os.path.join(dp_, f))
    assert not bad, bad


def test_binding_structs_mirror_the_library(mbavo):
    """ABI 3: the ctypes mirrors of every struct that crosses the boundary have the size the library was compiled with
    (mbavo_sizeof), and the header's revision is the library's."""
    lib, capi = mbavo.load(), mbavo.capi
    assert lib.mbavo_abi_version() == 3
    mirrors = [capi.Problem, capi.TrackOpts, capi.LmBatchOpts, capi.VoOptions, capi.EngineOpts, capi.VoState, capi.TraceRec, capi.Level,
               capi.LmBatchResult, capi.VoInfo]
    for which, cls in enumerate(mirrors):
        assert lib.mbavo_sizeof(which) == C.sizeof(cls), (which, cls.__name__, lib.mbavo_sizeof(which), C.sizeof(cls))
    assert lib.mbavo_sizeof(len(mirrors)) == -1


def test_environment_is_read_in_one_place():
    """the switches that change results or scheduling are options; the environment is an override layer
    read by ONE function (csrc/host_math.cpp: scan_environment behind options.h's read_env_overrides, once per process); at most four getenv names (diagnostics) elsewhere."""
    names = {}
    base = os.path.join(ROOT, "mba-vo_amd", "csrc")
    for f in os.listdir(base):
        txt = open(os.path.join(base, f), errors="ignore").read()
        if f == "host_math.cpp":  # the one reader: cut its body out
            a = txt.index("static EnvOverrides scan_environment()")
            txt = txt[:a] + txt[txt.index("return e;", a):]
        for m in re.finditer(r'getenv\("(MBAVO_[A-Z0-9_]+)"\)', txt):
            names.setdefault(m.group(1), set()).add(f)
    assert set(names) <= {"MBAVO_TIMING", "MBAVO_LM_STAMPS", "MBAVO_LM_STATS"}, names
