"""sha256 (16 hex digits) over the library sources: identifies WHICH build a profile / PMC file was measured on.
bench.py quotes roofline.traffic only from a PMC file that carries the hash of the sources it is running
(.git does not travel to the GPU box, so `git rev-parse HEAD` is not available there)."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def src_sha():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "signaltrain_amd", "csrc", "*")) + [os.path.join(ROOT, "include", "signaltrain_hip.h")])
    for f in files:
        if os.path.isfile(f) and f.endswith((".h", ".hip", ".c", "Makefile")):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(src_sha())
