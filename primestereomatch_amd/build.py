"""Build libprimesm_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(force: bool = False) -> str:
    cmd = ["make", "-C", os.path.join(_HERE, "csrc")]
    if force:
        cmd.append("-B")
    subprocess.run(cmd, check=True)
    out = os.path.join(_HERE, "lib", "libprimesm_hip.so")
    if not os.path.exists(out):
        raise RuntimeError("build did not produce " + out)
    return out


if __name__ == "__main__":
    print(build())
