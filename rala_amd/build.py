"""In-tree builds of the native pieces (no network, no pip).

* ``build_hip()``   -> rala_amd/csrc/librala_hip.so   (hipcc --offload-arch=gfx950)
* ``build_host()``  -> rala_amd/host/librala.so + rala_amd/host/rala (C++ host API / CLI)
* ``build_synth()`` -> rala_amd/synth/libralasynth.so (synthetic input generator)
* ``build_oracle()``-> oracle/_build/liboracle.so and, where /root/reference exists,
  oracle/_ref/liboracle_ref.so (test infrastructure; building the checker is not using it)

Every target is rebuilt only when a source is newer than the output.
"""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "rala_amd")
REFERENCE = "/root/reference"


def _stale(out, srcs):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(s) > t for s in srcs if os.path.exists(s))


def _run(cmd, cwd=None):
    res = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("build failed: %s\n%s" % (" ".join(cmd), res.stdout))
    return res.stdout


def _link(cmd_before_out, out, cmd_after_out):
    """Compile to a private temporary next to `out` and rename it into place: several processes
    (ranks under torchrun, pytest workers) may find the same stale library at once, and nobody may
    ever dlopen a half-written file."""
    tmp = "%s.%d.tmp" % (out, os.getpid())
    try:
        _run(cmd_before_out + ["-o", tmp] + cmd_after_out)
        os.replace(tmp, out)
    finally:
        if os.path.exists(tmp):
            os.unlink(tmp)


def _glob(d, exts):
    out = []
    for base, _, files in os.walk(d):
        for f in files:
            if f.endswith(exts):
                out.append(os.path.join(base, f))
    return sorted(out)


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def build_hip(force=False, jobs=None):
    """One object per .hip file (rala_amd/_build/*.o, rebuilt when that file or any header is newer),
    compiled side by side, then linked: editing one kernel file costs one compile, not fourteen."""
    from concurrent.futures import ThreadPoolExecutor

    d = os.path.join(PKG, "csrc")
    out = os.path.join(d, "librala_hip.so")
    srcs = _glob(d, (".hip",))
    hdrs = _glob(d, (".h", ".hpp")) + _glob(os.path.join(ROOT, "include"), (".h",))
    obj_dir = os.path.join(PKG, "_build")
    os.makedirs(obj_dir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
             "-I" + os.path.join(ROOT, "include"), "-I" + d]
    # (measurement scripts: extra definitions for a variant build, e.g. RALA_HIPCC_FLAGS='-DRALA_ROW_STORE_MOD="sc1"')
    import shlex
    flags += shlex.split(os.environ.get("RALA_HIPCC_FLAGS", ""))
    cc = hipcc()
    objs, todo = [], []
    for src in srcs:
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            todo.append((src, obj))

    def compile_one(job):
        src, obj = job
        tmp = "%s.%d.tmp" % (obj, os.getpid())
        try:
            _run([cc] + flags + ["-c", src, "-o", tmp])
            os.replace(tmp, obj)
        finally:
            if os.path.exists(tmp):
                os.unlink(tmp)

    if todo:
        with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as pool:
            list(pool.map(compile_one, todo))
    if force or todo or _stale(out, objs):
        # (-z defs: a launcher that is declared but not defined must fail here, not at dlopen on the GPU box)
        _link([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-z,defs"], out, objs + ["-ldl", "-lpthread"])
    return out


def build_synth(force=False):
    d = os.path.join(PKG, "synth")
    out = os.path.join(d, "libralasynth.so")
    srcs = [os.path.join(d, "synth.cpp")]
    if force or _stale(out, srcs):
        _link(["g++", "-O2", "-std=c++14", "-fPIC", "-shared"], out, srcs)
    return out


def build_host(force=False):
    d = os.path.join(PKG, "host")
    srcs = _glob(d, (".cpp",))
    if not srcs:
        return None
    out = os.path.join(d, "librala.so")
    deps = srcs + _glob(d, (".hpp", ".h")) + _glob(os.path.join(ROOT, "include"), (".h",))
    lib_srcs = [s for s in srcs if not s.endswith("cli.cpp") and not s.endswith("_capi.cpp")]
    if force or _stale(out, deps):
        hip = build_hip()
        _link(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I" + os.path.join(ROOT, "include"),
               "-I" + d], out, lib_srcs + [hip, "-Wl,-rpath," + os.path.dirname(hip), "-lz"])
    # the clean-up stages and the readers alone (no HIP dependency), for the CPU test-suite
    ag = os.path.join(d, "libassembly_graph.so")
    ag_srcs = [os.path.join(d, f) for f in ("assembly_graph.cpp", "assembly_graph_capi.cpp", "io.cpp", "io_capi.cpp")]
    if force or _stale(ag, ag_srcs + [os.path.join(d, "assembly_graph.hpp"), os.path.join(d, "io.hpp")]):
        _link(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I" + d], ag, ag_srcs + ["-lz"])
    # C entry points over stand-alone Pile / Overlap objects (tests of the class interface)
    api = os.path.join(d, "librala_api.so")
    api_src = os.path.join(d, "host_api_capi.cpp")
    if force or _stale(api, deps):
        _link(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I" + os.path.join(ROOT, "include"),
               "-I" + d], api, [api_src, out, "-Wl,-rpath," + d, "-Wl,-rpath," + os.path.join(PKG, "csrc")])
    exe = os.path.join(d, "rala")
    cli = os.path.join(d, "cli.cpp")
    if force or _stale(exe, deps):
        _link(["g++", "-O2", "-std=c++17", "-pthread", "-I" + d], exe,
              [cli, out, "-Wl,-rpath," + d, "-Wl,-rpath," + os.path.join(PKG, "csrc")])
    build_reference_cli(out, force)
    return out


def build_reference_cli(librala, force=False):
    """Drop-in check of SURVEY 8(b): the reference's own command line, compiled from where it
    lies (never copied), against this package's headers and librala.so.  Only possible where
    /root/reference exists; the binary (rala_amd/host/_refcli/rala_ref, git-ignored) travels to
    the GPU box like the other built files."""
    src = os.path.join(REFERENCE, "src", "main.cpp")
    if not os.path.exists(src):
        return None
    d = os.path.join(PKG, "host")
    out_dir = os.path.join(d, "_refcli")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "rala_ref")
    if force or _stale(exe, [src, librala]):
        _link(["g++", "-O2", "-std=c++11", "-pthread", "-I" + d], exe,
              [src, librala, "-Wl,-rpath," + d, "-Wl,-rpath," + os.path.join(PKG, "csrc")])
    return exe


def build_oracle(force=False):
    d = os.path.join(ROOT, "oracle")
    if force:
        _run(["make", "clean"], cwd=d)
    _run(["make"], cwd=d)
    out = [os.path.join(d, "_build", "liboracle.so")]
    if os.path.isdir(os.path.join(REFERENCE, "src")):
        _run(["make", "ref"], cwd=d)
        out.append(os.path.join(d, "_ref", "liboracle_ref.so"))
    return out


def build_all(force=False):
    return {"hip": build_hip(force), "synth": build_synth(force), "host": build_host(force),
            "oracle": build_oracle(force)}
