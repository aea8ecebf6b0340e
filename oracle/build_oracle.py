"""Build the oracle's C restatements (gcc) into oracle/_build/ — building the checker is not using it."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build")


def build():
    os.makedirs(OUT, exist_ok=True)
    so = os.path.join(OUT, "libmws_oracle.so")
    src = os.path.join(HERE, "mws.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, src])
    return so


if __name__ == "__main__":
    print(build())
