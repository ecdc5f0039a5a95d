"""Build the plain-C corr oracle (test infrastructure): oracle/_build/libcorr_oracle.so."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build", "libcorr_oracle.so")


def build(verbose=False):
    src = os.path.join(HERE, "corr_oracle.c")
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not os.path.exists(OUT) or os.path.getmtime(src) > os.path.getmtime(OUT):
        cmd = ["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC", "-o", OUT, src, "-lm"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(True)
