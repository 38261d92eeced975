"""Build the HIP library in-tree (hipcc cross-compiles gfx950 without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libgsr_hip.so")
SOURCES = ["gsr_kernels.hip", "loss_kernels.hip", "optim_kernels.hip", "knn_kernels.hip"]
HEADERS = ["gsr_math.h", "adam_math.h", "radix_sort.h", "blend_common.h", os.path.join("..", "..", "include", "gsr.h")]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -> csrc/libgsr_hip.so.  Returns the library path."""
    if force or _stale():
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
               "-Wno-unused-result",
               # keep scalar f32 arithmetic scalar: hipcc's SLP pass packs adjacent adds/muls into v_pk_*_f32 and pays
               # for it in v_mov operand shuffles (MI355X guide, "packed f32 VALU ... an anti-lever"); the kernels
               # that want packed math ask for it explicitly with float2 vector types
               "-fno-slp-vectorize", "-Wl,-soname,libgsr_hip.so"] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


EXT_DIR = os.path.join(CSRC, "torch_build")
EXT_LIB = os.path.join(EXT_DIR, "gsr_torch.so")
EXT_SRC = os.path.join(CSRC, "torch_ext.cpp")


def build_torch_ext(force: bool = False, verbose: bool = False) -> str:
    """torch.utils.cpp_extension -> csrc/torch_build/gsr_torch.so: the PyTorch-ROCm extension over the C ABI
    (csrc/torch_ext.cpp; plain C++, the kernels stay in libgsr_hip.so which it links by $ORIGIN-relative rpath)."""
    build()
    stale = (not os.path.exists(EXT_LIB)) or any(
        os.path.getmtime(f) > os.path.getmtime(EXT_LIB) for f in (EXT_SRC, os.path.join(HERE, "..", "include", "gsr.h")))
    if force or stale:
        from torch.utils import cpp_extension as ce
        os.makedirs(EXT_DIR, exist_ok=True)
        os.environ.setdefault("PYTORCH_ROCM_ARCH", "gfx950")
        ce.load(name="gsr_torch", sources=[EXT_SRC], build_directory=EXT_DIR, is_python_module=False, verbose=verbose,
                extra_cflags=["-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-Wno-deprecated-declarations"],
                extra_include_paths=["/opt/rocm/include"],
                extra_ldflags=[f"-L{CSRC}", "-l:libgsr_hip.so", "-Wl,-rpath,'$$ORIGIN/..'", "-L/opt/rocm/lib", "-lamdhip64",
                               "-lc10_hip", "-ltorch_hip"])
    return EXT_LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_torch_ext(force=True, verbose=True))
