"""Hash of the sources libceeg.so is built from (csrc/*.hip, csrc/*.h, include/ceeg.h).  The Makefile bakes it into the
library (ceeg_source_hash()); `_lib.check_build_matches_sources()` and __graft_entry__.smoke() compare it with the tree,
so a stale binary pushed next to newer sources cannot pass for them.  Run as a script it prints the C header."""
import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def source_files():
    csrc = os.path.join(_HERE, "csrc")
    names = sorted(f for f in os.listdir(csrc) if f.endswith((".hip", ".h")) and os.path.isfile(os.path.join(csrc, f)))
    return [os.path.join(csrc, f) for f in names] + [os.path.join(_HERE, "..", "include", "ceeg.h")]


def source_hash():
    h = hashlib.sha256()
    for path in source_files():
        h.update(os.path.basename(path).encode() + b"\0")
        h.update(open(path, "rb").read())
        h.update(b"\0")
    return h.hexdigest()[:32]


if __name__ == "__main__":
    print(f'#define CEEG_SOURCE_HASH "{source_hash()}"')
