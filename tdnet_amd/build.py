"""Build libtdnet_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The library is STAMPED with a hash of what it was built from (every *.h / *.hip under csrc/ + include/tdnet.h + the compile flags + the
ROCm release): the hash is
compiled into tdnet_version(), build() rebuilds whenever the stamp of the existing .so differs from the sources on disk (mtimes are
not trusted: the prebuilt .so travels to the GPU box with the tree), and smoke() / tests/test_gpu_harness.py assert that the
library a GPU process loaded carries the hash of the shipped sources -- so a green GPU run proves it ran HEAD's kernels.
"""
import glob
import hashlib
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "lib", "libtdnet_hip.so")                   # the PRODUCT library: the C ABI of include/tdnet.h, nothing else (csrc/td_model.hip)
OUT_TEST = os.path.join(HERE, "lib", "libtdnet_hip_test.so")         # + the tests' single-operator entry points and probes (csrc/td_model_test.hip, include/tdnet_test.h)
HEADER = os.path.join(os.path.dirname(HERE), "include", "tdnet.h")
HEADER_TEST = os.path.join(os.path.dirname(HERE), "include", "tdnet_test.h")
STAMP_MARK = b"tdnet-src-hash:"


FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", "-ffp-contract=off"]


def extra_flags():
    """TDNET_EXTRA_CXXFLAGS: flags of a NON-shipping build (e.g. -DTDNET_TIMING_PROBES for tools/ab_opts.py's skip probe).  Part of the stamp, so
    such a library never passes for the one the default flags build: a process without the variable rebuilds it."""
    return os.environ.get("TDNET_EXTRA_CXXFLAGS", "").split()


def sources():
    """The regular *.h / *.hip files of csrc/ + the C-ABI header: an editor backup, a stray directory or a build product next to them
    neither changes the stamp nor breaks the hash."""
    return sorted(p for p in glob.glob(os.path.join(CSRC, "*")) if os.path.isfile(p) and p.endswith((".h", ".hip"))) + [HEADER, HEADER_TEST]


def toolchain_id():
    """ROCm release of the image (the compiler that turns the sources into the .so); the same file on the GPU box."""
    try:
        with open("/opt/rocm/.info/version") as f:
            return f.read().strip()
    except OSError:
        return "unknown"


def source_hash():
    """sha256 over (file name, content) of every source + the compile flags + the ROCm release, 16 hex digits: a change of
    -ffp-contract or of the target arch rebuilds like a change of a kernel does."""
    h = hashlib.sha256()
    h.update((" ".join(FLAGS + extra_flags()) + "\0" + toolchain_id() + "\0").encode())
    for p in sources():
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


def built_hash(path=OUT):
    """The stamp inside a built library (read from the file, without loading it); None if absent."""
    if not os.path.exists(path):
        return None
    with open(path, "rb") as f:
        m = re.search(re.escape(STAMP_MARK) + rb"([0-9a-f]{16})", f.read())
    return m.group(1).decode() if m else None


def _compile(src, out, want, verbose):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + extra_flags() + ['-DTDNET_SRC_HASH="%s"' % want, os.path.join(CSRC, src), "-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    if built_hash(out) != want:
        raise RuntimeError("%s was built but does not carry the source hash %s" % (os.path.basename(out), want))


def build(force=False, verbose=False, test_lib=True):
    """Both libraries carry the same stamp (one hash over all sources): the product library from td_model.hip, the tests' superset from
    td_model_test.hip.  Stale ones are rebuilt, side by side (two hipcc processes)."""
    want = source_hash()
    todo = [(src, out) for src, out in (("td_model.hip", OUT), ("td_model_test.hip", OUT_TEST))
            if (out == OUT or test_lib) and (force or built_hash(out) != want)]
    if not todo:
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if len(todo) == 1:
        _compile(todo[0][0], todo[0][1], want, verbose)
    else:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(2) as ex:
            for f in [ex.submit(_compile, src, out, want, verbose) for src, out in todo]:
                f.result()
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
