"""Builds the test-only variants of the library (fault injection, the checksumming ring instantiation) into
tools/probes/ so that they travel to the GPU box with the tree.  Not part of the product build
(__graft_entry__.build() builds libsnappy_hip.so only); tests/test_gpu_faults.py builds whatever is missing or stale."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = {
    "fault1": ["-DENC_INJECT_ORDER_FAULT=1", "-DD2_INJECT_GIVE_UP=1"],
    "fault7": ["-DENC_INJECT_ORDER_FAULT=7", "-DD2_INJECT_GIVE_UP=7"],
    "fusedcrc": ["-DD2_FUSED_CRC=1"],
    "dbg": ["-DSNAPPY_HIP_DEBUG"],  # reads the debug knobs (SNAPPY_HIP_STATS, SNAPPY_HIP_SPLIT_KNOBS ...)
}


def build(names=None):
    for name in names or VARIANTS:
        r = subprocess.run([os.path.join(ROOT, "tools", "mkvariant.sh"), name] + VARIANTS[name])
        if r.returncode:
            print("variant %s did not build (rc %d)" % (name, r.returncode), file=sys.stderr)


if __name__ == "__main__":
    build(sys.argv[1:])
