"""Builds libsnappy_hip.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsnappy_hip.so")
SOURCES = ["snappy_hip.hip", "common.h", "decode_kernel.h", "decode2_kernel.h", "index_kernel.h",
           "encode_kernel.h", "crc_pack_kernels.h", "framed_kernels.h", "split_kernels.h", "sparse_kernel.h"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "snappy_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, debug=False):
    """hipcc --offload-arch=gfx950 -shared: cross-compiles without a GPU.

    debug=True (or SNAPPY_HIP_BUILD_DEBUG=1 at build time) adds -DSNAPPY_HIP_DEBUG: only such a
    build reads the SNAPPY_HIP_* debug knobs and carries the kernels' phase-ablation / cycle-counter
    code.  The default library has neither."""
    debug = debug or os.environ.get("SNAPPY_HIP_BUILD_DEBUG") == "1"
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    tmp = "%s.%d.tmp" % (LIB, os.getpid())  # (several ranks may find the library stale at once)
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
           "-Wno-unused-function", "-o", tmp, os.path.join(CSRC, "snappy_hip.hip"),
           "-Wl,-rpath,/opt/rocm/lib"]
    if debug:
        cmd.insert(1, "-DSNAPPY_HIP_DEBUG")
    if verbose:
        print(" ".join(cmd))
    try:
        subprocess.run(cmd, check=True)
        os.replace(tmp, LIB)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force=True, verbose=True, debug="--debug" in sys.argv))
