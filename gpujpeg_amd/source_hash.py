"""sha256 over the device sources (gpujpeg_amd/csrc/*.hip, *.h -- without the files that hold no device code: the HIP runtime wrappers and the
host's C header): PMC figures kept under profiles/ are only quoted by bench.py for the kernels they were measured on (tools/rocprof_summary.py
stamps them with this value)."""
import glob
import hashlib
import os

HERE = os.path.dirname(os.path.abspath(__file__))
HOST_ONLY = {"gj_runtime.hip", "gj_internal.h"}  # (no kernel, no device function: a change there does not touch what the counters measured)


def kernel_source_hash():
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")) + glob.glob(os.path.join(HERE, "csrc", "*.h"))):
        if os.path.basename(f) in HOST_ONLY:
            continue
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]
