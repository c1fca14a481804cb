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
                             ("mesh_refine_driver_rowmajor", ["-DEIGEN_DEFAULT_TO_ROW_MAJOR"], "mesh_refine_driver.cpp")):
        src = os.path.join(HERE, cpp)
        out = os.path.join(OUT, name)
        cmd = ["g++", "-std=c++11", "-O2", "-w"] + extra + [
            "-I" + os.path.join(ROOT, "include"), "-I" + e, src, "-o", out,
            "-L" + os.path.join(ROOT, "visma_amd", "lib"), "-lvisma_icp",
            "-Wl,-rpath,$ORIGIN/../../../visma_amd/lib"]
        subprocess.check_call(cmd)
        outs.append(out)
    return outs


if __name__ == "__main__":
    print(build())
