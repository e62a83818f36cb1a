"""sha256 over the kernel sources (nim-snappy_amd/csrc/* and include/snappy_hip.h): what a profile in profiles/ was
measured on.  tools/profile_summary.py stores it in *_traffic.json, bench.py compares it with the tree it runs from."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha256(root=ROOT):
    h = hashlib.sha256()
    d = os.path.join(root, "nim-snappy_amd", "csrc")
    files = sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith((".h", ".hip")))
    files.append(os.path.join(root, "include", "snappy_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


if __name__ == "__main__":
    print(csrc_sha256())
