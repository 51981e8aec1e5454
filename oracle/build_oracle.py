"""ORACLE — TEST INFRASTRUCTURE ONLY.  Compiles the C restatement(s) under oracle/ into oracle/_build/.

Called by ``__graft_entry__.build()`` ("building the checker is not using it") and lazily by
``oracle.farmhash`` the first time a test needs the library.  gcc only; no GPU involved.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_build")


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    src = os.path.join(HERE, "farmhash64.c")
    out = os.path.join(OUT_DIR, "liboracle_farmhash.so")
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-std=c99", "-shared", "-fPIC", "-o", out, src])
    return out


if __name__ == "__main__":
    print(build(force=True))
