"""Builds the gfx950 HIP library in-tree (smcsmc_amd/csrc/libsmcsmc_pf.so) and the host binary.

hipcc cross-compiles for gfx950 without a GPU present, so this runs in the CPU-only build
container as well as on the MI355X box.  The built .so stays in-tree (git-ignored).
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libsmcsmc_pf.so")
BIN = os.path.join(os.path.dirname(HERE), "bin", "smcsmc")

# -O2, not -O3: the row kernels are one long dependent chain per wavefront and their time is their instruction count;
# measured on the headline shape, alternating libraries on one box: -O2 27.9k, -O3 27.6k, -O1 27.4k, -Os 27.3k segments/s
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O2", "-ffp-contract=off", "-fPIC", "-std=c++17",
               "-Wno-unused-value", "-Wno-unused-result"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built (no CPU fallback exists)")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _compile_units(out, extra_flags, force, tag):
    """Each translation unit to its own object file (two hipcc processes side by side: the units take minutes), then one link.
    An object is rebuilt when its unit or any header is newer."""
    import concurrent.futures
    units = [os.path.join(CSRC, "pf_hip.hip"), os.path.join(CSRC, "pf_mp.hip"), os.path.join(CSRC, "pf_probe.hip")]
    deps = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "smcsmc_pf.h"))
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = [os.path.join(objdir, os.path.basename(u)[:-4] + tag + ".o") for u in units]
    todo = [(u, o) for u, o in zip(units, objs) if force or _stale(o, [u] + deps)]
    if todo:
        def cc(uo):
            subprocess.check_call([_hipcc()] + HIPCC_FLAGS + extra_flags + ["-c", uo[0], "-o", uo[1]])
        with concurrent.futures.ThreadPoolExecutor(len(todo)) as ex:
            list(ex.map(cc, todo))
    if todo or force or _stale(out, objs):
        subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


def build_lib(force=False):
    return _compile_units(LIB, [], force, "")


def build_stamps_lib(force=False):
    """The profiling build of the library (-DPF_STAMPS: wall-clock stamps per wavefront, row and phase of the extend
    workgroups, read back through pf_debug_stamps).  Used by profiles/stamps.py and profiles/stamps_mp.py only."""
    return _compile_units(os.path.join(CSRC, "libsmcsmc_pf_stamps.so"), ["-DPF_STAMPS"], force, "_stamps")


def build_all(force=False):
    lib = build_lib(force)
    host = os.path.join(CSRC, "host")
    if os.path.isdir(host) and os.path.exists(os.path.join(host, "Makefile")):
        subprocess.check_call(["make", "-s", "-C", host])
    return lib
