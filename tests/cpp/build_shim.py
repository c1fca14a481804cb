"""Compile tests/cpp/shim_driver.cpp against the shim headers.

Needs Eigen headers at compile time only.  This image has none of its own, so
the build uses the copy vendored inside the reference checkout when that is
present (this container); the resulting binaries (tests/cpp/_build/, git-ignored)
travel to the GPU box with the snapshot like the other prebuilt artefacts.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "_build")
EIGEN_CANDIDATES = ["/usr/include/eigen3", "/usr/local/include/eigen3",
                    "/root/reference/thirdparty/Open3D/3rdparty/Eigen"]


def eigen_dir():
    for d in EIGEN_CANDIDATES:
        if os.path.exists(os.path.join(d, "Eigen", "Core")):
            return d
    return None


def build():
    e = eigen_dir()
    if e is None:
        return None
    os.makedirs(OUT, exist_ok=True)
    outs = []
    for name, extra, cpp in (("shim_driver", [], "shim_driver.cpp"),
                             ("shim_driver_rowmajor", ["-DEIGEN_DEFAULT_TO_ROW_MAJOR"], "shim_driver.cpp"),
                             ("mesh_refine_driver", [], "mesh_refine_driver.cpp"),
                             ("mesh_refine_driver_rowmajor", ["-DEIGEN_DEFAULT_TO_ROW_MAJOR"], "mesh_refine_driver.cpp"),
                             ("annotate_driver", [], "annotate_driver.cpp"),
                             ("annotate_driver_rowmajor", ["-DEIGEN_DEFAULT_TO_ROW_MAJOR"], "annotate_driver.cpp")):
        src = os.path.join(HERE, cpp)
        out = os.path.join(OUT, name)
        cmd = ["g++", "-std=c++11", "-O2", "-w"] + extra + [
            "-I" + os.path.join(ROOT, "include"), "-I" + e, src, "-o", out,
            "-L" + os.path.join(ROOT, "visma_amd", "lib"), "-lvisma_icp",
            "-Wl,-rpath,$ORIGIN/../../../visma_amd/lib"]
        subprocess.check_call(cmd)
        outs.append(out)
    return outs


O3D_SRC = "/root/reference/thirdparty/Open3D/src"          # the REAL Open3D 0.3.0 headers (this container only)
REF_LIB_DIR = os.path.join(ROOT, "oracle", "_ref")          # the compiled reference: stands in for Open3D's libCore


def build_real():
    """Against the REAL Open3D headers, both Eigen storage orders:
      * interpose/open3d_registration_interpose.cpp -> _build/libvisma_open3d_interpose[_rowmajor].so (the mangled
        open3d::RegistrationICP / EvaluateRegistration, forwarding to the C ABI);
      * tests/cpp/interpose_driver.cpp -- a caller that includes ONLY Open3D's headers -- linked with the interposer
        AHEAD of oracle/_ref/libvisma_ref.so, which holds Open3D's own definitions of the same symbols (column-major:
        the order the compiled reference was built with; the row-major objects are compiled, not linked);
      * shim_driver.cpp / mesh_refine_driver.cpp compiled (objects only) with Open3D's include directory FIRST, the
        way INTEGRATION.md tells an integrator to order them.
    Returns the list of artefacts, or None where the reference checkout is absent (the GPU box: the binaries travel)."""
    e = eigen_dir()
    if e is None or not os.path.exists(os.path.join(O3D_SRC, "Core", "Registration", "Registration.h")):
        return None
    os.makedirs(OUT, exist_ok=True)
    inc = ["-I" + O3D_SRC, "-I" + e, "-I" + os.path.join(ROOT, "include")]          # Open3D's headers first
    outs = []
    for tag, extra in (("", []), ("_rowmajor", ["-DEIGEN_DEFAULT_TO_ROW_MAJOR"])):
        so = os.path.join(OUT, "libvisma_open3d_interpose%s.so" % tag)
        subprocess.check_call(["g++", "-std=c++11", "-O2", "-w", "-shared", "-fPIC"] + extra + inc + [
            os.path.join(ROOT, "interpose", "open3d_registration_interpose.cpp"), "-o", so,
            "-L" + os.path.join(ROOT, "visma_amd", "lib"), "-lvisma_icp", "-Wl,-rpath,$ORIGIN/../../../visma_amd/lib"])
        outs.append(so)
        for cpp in ("interpose_driver.cpp", "shim_driver.cpp", "mesh_refine_driver.cpp", "annotate_driver.cpp"):
            obj = os.path.join(OUT, cpp.replace(".cpp", "_real_headers%s.o" % tag))
            subprocess.check_call(["g++", "-std=c++11", "-O2", "-w", "-c", "-DVISMA_ICP_OPEN3D_NO_UMBRELLA"] + extra + inc +
                                  [os.path.join(HERE, cpp), "-o", obj])
            outs.append(obj)
    if os.path.exists(os.path.join(REF_LIB_DIR, "libvisma_ref.so")):
        exe = os.path.join(OUT, "interpose_driver")
        subprocess.check_call(["g++", os.path.join(OUT, "interpose_driver_real_headers.o"), "-o", exe,
                               "-L" + OUT, "-lvisma_open3d_interpose",                   # AHEAD of "libCore"
                               "-L" + REF_LIB_DIR, "-lvisma_ref", "-fopenmp",
                               "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,$ORIGIN/../../../oracle/_ref",
                               "-Wl,-rpath,$ORIGIN/../../../visma_amd/lib"])
        outs.append(exe)
    return outs


if __name__ == "__main__":
    print(build())
    print(build_real())
