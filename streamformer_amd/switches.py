"""`python -m streamformer_amd.switches`: the library's environment switches (name, current value, what it does) — the one
table of csrc/sf_switches.h, as the built library reports it."""
import os

from . import _native as nat


def table():
    out, i = [], 0
    while True:
        n = nat.lib.sf_switch_info(i, 0)
        if n is None:
            return out
        out.append((n.decode(), (nat.lib.sf_switch_info(i, 1) or b"").decode()))
        i += 1


if __name__ == "__main__":
    for name, what in table():
        print(f"{name:32s} {os.environ.get(name, '-'):8s} {what}")
