"""sha256 over the device sources (gpujpeg_amd/csrc/*.hip, *.h): PMC figures kept under profiles/ are only quoted by bench.py for the
kernels they were measured on (tools/rocprof_summary.py stamps them with this value)."""
import glob
import hashlib
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def kernel_source_hash():
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")) + glob.glob(os.path.join(HERE, "csrc", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]
