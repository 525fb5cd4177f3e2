"""Build libcvmi355.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def build(verbose=False, jobs=8):
    cmd = ["make", "-C", os.path.join(HERE, "csrc"), f"-j{jobs}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:])
        print(r.stderr[-8000:])
    if r.returncode != 0:
        raise RuntimeError("hipcc build of libcvmi355.so failed")
    return os.path.join(HERE, "libcvmi355.so")


if __name__ == "__main__":
    print(build(verbose=True))
