"""Builds csrc/*.hip into libddsp_amd.so for gfx950 (hipcc cross-compiles without a GPU).

Staleness is decided by CONTENT, not by mtime: every object carries the sha256 of its source, of every header under
csrc/ and include/, of the flags and of the compiler's version string (csrc/build/<name>.o.sha256); an object is
recompiled when that digest differs.  A fresh checkout has no objects (they are git-ignored), so `build_lib()` there
compiles everything; `build_lib(force=True)` (or `python build.py --force`) does so anywhere.  The digests of the
last build are also written to hipddsp/build_manifest.json (tracked), so that a prebuilt library that travelled to
a GPU box can be checked against the sources next to it (`verify_lib()`)."""
import hashlib
import json
import os
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libddsp_amd.so")
MANIFEST = os.path.join(HERE, "build_manifest.json")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    hs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    hs.append(os.path.join(os.path.dirname(PKG), "include", "ddsp_amd.h"))
    return hs


def _sha(path):
    with open(path, "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()


_compiler_id = None


def _compiler():
    global _compiler_id
    if _compiler_id is None:
        r = subprocess.run([HIPCC, "--version"], capture_output=True, text=True)
        _compiler_id = hashlib.sha256((r.stdout + r.stderr).encode()).hexdigest()[:16]
    return _compiler_id


def source_digests():
    """{object name: digest of everything that object is compiled from}."""
    head = hashlib.sha256()
    for h in _headers():
        head.update(os.path.basename(h).encode())
        head.update(_sha(h).encode())
    head.update(" ".join(FLAGS).encode())
    head.update(_compiler().encode())
    base = head.hexdigest()
    return {src[:-4] + ".o": hashlib.sha256((base + _sha(os.path.join(CSRC, src))).encode()).hexdigest()
            for src in _sources()}


def _stamp(obj):
    try:
        with open(obj + ".sha256") as fh:
            return fh.read().strip()
    except OSError:
        return None


def build_lib(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    want = source_digests()
    jobs, objs = [], []
    for src in _sources():
        name = src[:-4] + ".o"
        o = os.path.join(OBJ, name)
        objs.append(o)
        if force or not os.path.exists(o) or _stamp(o) != want[name]:
            jobs.append((name, [HIPCC, *FLAGS, "-c", os.path.join(CSRC, src), "-o", o]))

    def run(cmd):
        if verbose:
            print("[build]", " ".join(os.path.relpath(c, PKG) if os.path.isabs(c) and c.startswith(PKG) else c
                                      for c in cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    def compile_one(job):
        name, cmd = job
        stamp = os.path.join(OBJ, name + ".sha256")
        if os.path.exists(stamp):
            os.remove(stamp)
        run(cmd)
        with open(stamp, "w") as fh:
            fh.write(want[name])

    t0 = time.time()
    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(compile_one, jobs))
    manifest = {"objects": want, "flags": FLAGS, "compiler": _compiler()}
    old = None
    if os.path.exists(MANIFEST):
        try:
            old = json.load(open(MANIFEST))
        except ValueError:
            old = None
    relink = bool(jobs) or force or not os.path.exists(LIB) or old != manifest
    if relink:
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
        with open(MANIFEST, "w") as fh:
            json.dump(manifest, fh, indent=1, sort_keys=True)
            fh.write("\n")
    if verbose:
        print(f"[build] compiled {len(jobs)} of {len(objs)} objects ({', '.join(n for n, _ in jobs) or 'none stale'}), "
              f"{'linked' if relink else 'library up to date'}, {time.time() - t0:.1f} s", flush=True)
    return LIB


def verify_lib():
    """True when libddsp_amd.so exists and build_manifest.json matches the sources beside it (a prebuilt library that
    travelled with the tree is the build of THESE sources)."""
    if not (os.path.exists(LIB) and os.path.exists(MANIFEST)):
        return False
    try:
        return json.load(open(MANIFEST)).get("objects") == source_digests()
    except ValueError:
        return False


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv))
