"""In-tree builds: the gfx950 decoder library, the corpus generator, and (for tests) the CPU oracle."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_gpu(force=False):
    out = os.path.join(HERE, "libzstdgpu.so")
    srcs = [os.path.join(CSRC, f) for f in ("zstdgpu_host.cpp", "kernels.hip", "descriptors.hip", "noalloc.hip")]
    deps = srcs + [os.path.join(CSRC, "zstdgpu_internal.h"), os.path.join(CSRC, "zstdgpu_objects.h"), os.path.join(ROOT, "include", "zstdgpu.h")]
    if force or _stale(out, deps):
        subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Wshadow",
                               "-Wno-unused-value", "-o", out] + srcs)
    return out


def build_corpus(force=False):
    out = os.path.join(HERE, "_corpus.so")
    src = os.path.join(CSRC, "corpus.c")
    if force or _stale(out, [src]):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", out, src])
    return out


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


def build_all(force=False):
    build_gpu(force)
    build_corpus(force)
    build_oracle()
