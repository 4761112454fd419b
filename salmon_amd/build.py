"""Build every native artefact of salmon_amd for gfx950 (in-tree, no JIT cache).

  libsalmon_hip.so   product: host index builder + HIP kernels + C ABI   (hipcc --offload-arch=gfx950)
  oracle/_build/liboracle.so   CPU checker (test infrastructure)         (g++)
  oracle/_ref/libedlib_ref.so  the reference's infix aligner, compiled from /root/reference where present (g++)
  tools/_build/libsqsynth.so   synthetic transcriptome / read generator  (g++)
"""
import os, subprocess, sys, hashlib, concurrent.futures as cf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "salmon_amd", "csrc")
OBJ = os.path.join(ROOT, "build", "obj")
LIB = os.path.join(ROOT, "salmon_amd", "libsalmon_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-march=x86-64-v3", "-Wall", "-Wno-unused-function",
          "-Wno-unused-result", "-Wno-unused-value", "-Wno-pass-failed", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include")]


def _sources():
    out = []
    for sub in ("host", "hip"):
        d = os.path.join(CSRC, sub)
        for f in sorted(os.listdir(d)):
            if f.endswith((".cpp", ".hip")):
                out.append(os.path.join(d, f))
    return out


def _deps_stamp():
    h = hashlib.sha1()
    for base, _, files in os.walk(CSRC):
        for f in sorted(files):
            if f.endswith((".h", ".hpp")):
                h.update(open(os.path.join(base, f), "rb").read())
    for f in sorted(os.listdir(os.path.join(ROOT, "include"))):
        h.update(open(os.path.join(ROOT, "include", f), "rb").read())
    h.update((" ".join(COMMON) + repr(sorted(PER_FILE.items()))).encode())
    return h.hexdigest()[:12]


# inflate_dev.hip: its decoding loop is a wave acting as a scalar processor — every branch in it is uniform.  The structurizer's default turns such a region into
# flag registers and mask tests all the same (half of the loop's scalar instructions, and the scalar port is what bounds the kernel: DESIGN.md section 4); with
# uniform regions skipped the branches stay plain scalar jumps.  tests/test_inflate.py holds the result to zlib on the device.
PER_FILE = {"inflate_dev.hip": ["-mllvm", "-structurizecfg-skip-uniform-regions=1"], "gzip_dev.hip": ["-mllvm", "-structurizecfg-skip-uniform-regions=1"]}


def _compile(src, stamp):
    obj = os.path.join(OBJ, os.path.basename(src) + "." + stamp + ".o")
    if os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(src):
        return obj, False
    cmd = [HIPCC] + COMMON + PER_FILE.get(os.path.basename(src), []) + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s\n%s" % (src, r.stdout[-4000:], r.stderr[-8000:]))
    if r.stderr.strip():
        sys.stderr.write(r.stderr[-3000:])
    return obj, True


def build_product(verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    stamp = _deps_stamp()
    srcs = _sources()
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, stamp), srcs))
    objs = [o for o, _ in res]
    for f in os.listdir(OBJ):   # objects of older header stamps only bloat the tree that travels to the GPU box
        if f.endswith(".o") and os.path.join(OBJ, f) not in objs:
            os.remove(os.path.join(OBJ, f))
    if any(c for _, c in res) or not os.path.exists(LIB):
        cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950"] + objs + ["-lz", "-lpthread", "-ldl", "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-8000:])
        if verbose:
            print("built", LIB)
    return LIB


def build_cli():
    """salmon-hip: C++ host driver (index / quant) over the C ABI."""
    out = os.path.join(ROOT, "salmon_amd", "bin")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "salmon-hip")
    src = os.path.join(CSRC, "cli", "salmon_main.cpp")
    if os.path.exists(exe) and os.path.getmtime(exe) >= max(os.path.getmtime(src), os.path.getmtime(LIB)):
        return exe
    cmd = ["g++", "-O2", "-std=c++17", "-march=x86-64-v3", src, "-o", exe, "-L" + os.path.dirname(LIB), "-lsalmon_hip", "-lz",
           "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("cli build failed:\n" + r.stderr[-6000:])
    return exe


def build_oracle():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout[-3000:] + r.stderr[-6000:])
    return os.path.join(ROOT, "oracle", "_build", "liboracle.so")


def build_oracle_ref(reference="/root/reference"):
    """oracle/_ref/libedlib_ref.so: the reference's own infix aligner (src/edlib.cpp, row a5) compiled from where it lies
    under /root/reference plus a C shim; oracle/_ref/libdefaults_ref.so: its option defaults and log-space helpers (two header-only
    files).  Only where the reference tree exists (this container); the GPU box uses the prebuilt files that travel with the
    snapshot.  Returns the path of the first or None."""
    out = os.path.join(ROOT, "oracle", "_ref", "libedlib_ref.so")
    if not os.path.exists(os.path.join(reference, "src", "edlib.cpp")):
        return out if os.path.exists(out) else None
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref", "REF=" + reference], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle/_ref build failed:\n" + r.stdout[-3000:] + r.stderr[-6000:])
    return out


def build_tools():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "tools")], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("tools build failed:\n" + r.stdout[-3000:] + r.stderr[-6000:])
    return os.path.join(ROOT, "tools", "_build", "libsqsynth.so")


def build_microbench():
    """tools/*.hip: stand-alone gfx950 microbenchmarks quoted in DESIGN.md (random-sector gather ceiling, grid-barrier cost)."""
    out = os.path.join(ROOT, "tools", "_build"); os.makedirs(out, exist_ok=True)
    for name in ("gather_bench", "gridbar_bench", "gridbar2_bench", "fetch_calib"):
        src = os.path.join(ROOT, "tools", name + ".hip"); exe = os.path.join(out, name)
        if os.path.exists(exe) and os.path.getmtime(exe) >= os.path.getmtime(src):
            continue
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-Wno-unused-result", src, "-o", exe], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stderr[-4000:]))


def build_all():
    build_product()
    build_cli()
    build_oracle()
    build_oracle_ref()
    build_tools()
    build_microbench()


if __name__ == "__main__":
    build_all()
