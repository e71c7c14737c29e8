"""Build the native libraries in-tree (no JIT cache: the .so files travel with the repo).

  liblrm_accel.so  hipcc --offload-arch=gfx950 : HIP kernels + C-ABI + CPU index builder
  liblrm_synth.so  g++ -fopenmp                : synthetic reference / read generators (tooling)

hipcc cross-compiles gfx950 without a GPU, so this runs in the CPU-only build container.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")

ACCEL_SRCS = ["lrm_api.hip", "lrm_host.hip", "seed_kernels.hip", "gact_kernels.hip", "gact_bs_kernels.hip", "index_host.cpp", "io_host.cpp"]
ACCEL_DEPS = ACCEL_SRCS + ["lrm_internal.h", "../../include/lrm_accel.h", "../../include/lrm_index_host.h",
                           "../../include/lrm_io_host.h"]
ACCEL_LIB = os.environ.get("LRM_ACCEL_LIB") or os.path.join(HERE, "liblrm_accel.so")     # LRM_ACCEL_LIB: a tuning build (tools/)
SYNTH_LIB = os.path.join(HERE, "liblrm_synth.so")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def hipcc_path():
    for p in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if p and os.path.exists(p):
            return p
    return None


def build_accel(force=False, defines=(), out=None):
    deps = [os.path.join(CSRC, d) for d in ACCEL_DEPS]
    target = out or ACCEL_LIB
    if os.environ.get("LRM_ACCEL_LIB") and not out:
        return ACCEL_LIB                 # a tuning build chosen by the caller: use it as it is
    if not force and not _stale(target, deps):
        return target
    hipcc = hipcc_path()
    if hipcc is None:
        if os.path.exists(target):
            return target           # GPU box without a compiler on PATH: use the prebuilt library
        raise RuntimeError("hipcc not found and no prebuilt liblrm_accel.so")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fopenmp",
           "-I" + os.path.join(ROOT, "include")]
    cmd += ["-D" + d for d in defines]
    cmd += [os.path.join(CSRC, s) for s in ACCEL_SRCS]
    cmd += ["-lz", "-ldl", "-o", target]
    _run(cmd)
    return target


def build_synth(force=False):
    src = os.path.join(CSRC, "synth.cpp")
    if not force and not _stale(SYNTH_LIB, [src]):
        return SYNTH_LIB
    _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fopenmp", src, "-o", SYNTH_LIB])
    return SYNTH_LIB


def build_all(force=False):
    return build_accel(force), build_synth(force)


if __name__ == "__main__":
    print(build_all(force=True))
