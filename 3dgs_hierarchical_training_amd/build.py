"""Build the HIP library in-tree (hipcc cross-compiles gfx950 without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libgsr_hip.so")
SOURCES = ["gsr_kernels.hip", "loss_kernels.hip", "optim_kernels.hip", "knn_kernels.hip"]
HEADERS = ["gsr_math.h", "adam_math.h", "radix_sort.h", os.path.join("..", "..", "include", "gsr.h")]


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
               "-fno-slp-vectorize"] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
