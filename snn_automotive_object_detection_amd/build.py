"""In-tree build of libsnnhip.so (gfx950 only).  hipcc cross-compiles without a GPU; the built .so
is git-ignored but travels to the GPU box with the working tree."""
import fcntl
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libsnnhip.so")
SOURCES = ["snn_kernels.hip"]
HEADERS = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(ROOT, "include", "snn_hip.h"), os.path.join(ROOT, "include", "snn_hip_debug.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         "-ffp-contract=off",            # neuron arithmetic must round like the reference's un-fused ops
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


STAMP = os.path.join(LIB_DIR, "libsnnhip.stamp")


def source_digest() -> str:
    """content hash of everything the library is built from (file times do not survive a copy to another machine)"""
    import hashlib
    h = hashlib.sha256(" ".join(f for f in FLAGS if not f.startswith("-I")).encode())    # include paths move with the tree
    for d in [os.path.join(CSRC, s) for s in SOURCES] + HEADERS:
        with open(d, "rb") as f:
            h.update(os.path.basename(d).encode() + b"\0" + f.read())
    return h.hexdigest()


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != source_digest()


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile under an exclusive file lock and publish with an atomic rename, so that several ranks of one node
    (torchrun starts them together) can call this concurrently: one compiles, the others wait and reuse."""
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():          # somebody else built it while we waited
                return LIB_PATH
            tmp = LIB_PATH + ".tmp.%d" % os.getpid()
            cmd = [HIPCC] + FLAGS + ["-o", tmp] + [os.path.join(CSRC, s) for s in SOURCES]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if r.returncode != 0:
                if os.path.exists(tmp):
                    os.remove(tmp)
                raise RuntimeError("hipcc failed:\n" + r.stdout)
            os.replace(tmp, LIB_PATH)
            with open(STAMP + ".tmp", "w") as f:
                f.write(source_digest() + "\n")
            os.replace(STAMP + ".tmp", STAMP)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


ASAN_RUNTIME = "/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so"


def asan_runtime() -> str:
    """the shared AddressSanitizer runtime of the ROCm clang (to LD_PRELOAD into python)"""
    if os.path.exists(ASAN_RUNTIME):
        return ASAN_RUNTIME
    import glob
    hits = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    return hits[-1] if hits else ""


def build_asan(out_path: str) -> str:
    """Host-side AddressSanitizer build of the same translation unit (CPU box only: GPU ASan / xnack+ code objects are not
    available on this pool).  The device code is compiled un-instrumented; what is checked is the host half of the C ABI:
    argument validation, the level / image tables copied from caller memory, workspace layouts."""
    # host: -O1 -g instrumented; device: the product's -O3 (the LDS-DMA inline asm takes SGPR operands, which only the optimised device
    # code keeps out of vector registers - and the device half is not what this build is for)
    cmd = [HIPCC] + [f for f in FLAGS if f != "-O3"] + ["-Xarch_host", "-O1", "-Xarch_device", "-O3", "-g", "-fsanitize=address", "-fno-gpu-sanitize",
                                                          "-shared-libsan", "-o", out_path] + [os.path.join(CSRC, s) for s in SOURCES]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc (asan) failed:\n" + r.stdout)
    return out_path


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
