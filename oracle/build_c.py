"""ORACLE — builds the C restatement (oracle/text_ref.c) into oracle/_build/libtext_ref.so.  Building the checker is
not using it: only tests load the result."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build", "libtext_ref.so")


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "text_ref.c")
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(src):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.run(["gcc", "-O2", "-shared", "-fPIC", src, "-o", OUT], check=True)
    return OUT


if __name__ == "__main__":
    print(build(True))
