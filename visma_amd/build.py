"""Build the gfx950 shared library (HIP kernels + C ABI) in-tree with hipcc."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libvisma_icp.so")
SOURCES = ["kernels.hip", "grid.hip", "grid_coop.hip", "grid_wave.hip", "grid_ring.hip", "icp_loop.hip", "voxel.hip", "order.hip", "normals.hip", "mesh.hip", "hip_engine.cpp", "hip_engine_clouds.cpp", "hip_engine_passes.cpp", "hip_engine_comm.cpp", "driver.cpp", "aux_api.cpp", "corpus.cpp", "io.cpp"]
HEADERS = ["kernels.h", "device_common.h", "so3.h", "host_math.hpp", "engine.hpp", "hip_engine.hpp", "driver_ctx.hpp", "grid_coop_probe.h", "plane_math.hpp",
           os.path.join("..", "..", "include", "visma_icp.h"), os.path.join("..", "..", "include", "visma_icp_testing.h")]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (need ROCm; this package has no other backend)")
    return exe


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_lib(force=False, verbose=False, defines=(), out=None, extra_flags=()):
    """Compile visma_amd/lib/libvisma_icp.so for gfx950 (cross-compiles without a GPU).
    One hipcc process per source file, run side by side, then one link."""
    if out is None and not force and not is_stale():
        return LIB_PATH
    lib_path = out or LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, "_obj" + ("" if out is None else "_" + os.path.basename(out)))
    os.makedirs(obj_dir, exist_ok=True)
    # -fno-slp-vectorize (round 6): packed fp32 VALU ops issue at half the rate of plain ones on gfx950 (tools/ubench/valu_rate:
    # v_pk_fma_f32 4.9 cycles per wave against 2.6 for v_fma_f32 -- nothing to gain), and the register pairs they need cost the
    # persistent search kernel spills: 8 -> 4 spilled VGPRs, 33.3 -> 32.9 us per iteration (profiles/r06_build_knobs_ab.txt)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden", "-Wall",
             "-fno-slp-vectorize"]
    flags += ["-D" + d for d in defines] + list(extra_flags)
    procs, objs = [], []
    sources = list(SOURCES) + (["tile.hip"] if "VISMA_WITH_TILE" in defines else [])
    for src in sources:
        obj = os.path.join(obj_dir, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        cmd = [hipcc()] + flags + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    link = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib_path, "-ldl", "-lpthread"]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    shutil.rmtree(obj_dir, ignore_errors=True)
    return lib_path


EXPERIMENTS_LIB = os.path.join(LIB_DIR, "libvisma_icp_experiments.so")


def build_experiments(force=False):
    """The side build with the experiments that lost (tile.hip: -DVISMA_WITH_TILE) and the test seam
    visma_icp_create_with_engine (-DVISMA_TEST_SEAMS): what tests load through VISMA_ICP_LIB, never the product."""
    import fcntl
    os.makedirs(LIB_DIR, exist_ok=True)
    with open(os.path.join(LIB_DIR, ".experiments.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)                 # (several test processes may get here together)
        if force or not os.path.exists(EXPERIMENTS_LIB) or is_stale_against(EXPERIMENTS_LIB):
            tmp = EXPERIMENTS_LIB + ".tmp.so"
            build_lib(force=True, defines=("VISMA_WITH_TILE", "VISMA_TEST_SEAMS", "VISMA_SOLVE_IN_FOLD=1"), out=tmp)
            os.replace(tmp, EXPERIMENTS_LIB)
    return EXPERIMENTS_LIB


def is_stale_against(path):
    t = os.path.getmtime(path)
    deps = [os.path.join(CSRC, s) for s in SOURCES + ["tile.hip"] + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
