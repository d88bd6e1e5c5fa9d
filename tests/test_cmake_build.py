"""north_star: "Host code stays C++/CMake". The top-level CMakeLists.txt (project(Vulcan LANGUAGES CXX HIP),
CMAKE_HIP_ARCHITECTURES gfx950) builds vk_hip, vk_comm, vulcan, host_tests, fuse_sequence and rig_rehearsal and
installs a package that exports Vulcan::vulcan — the counterpart of upstream's `cuda_add_library(vulcan SHARED ...)`
(CMakeLists.txt:148) and of its exported package (cmake/install_package.cmake, cmake/PackageConfig.cmake.in).

Here, without a GPU: configure + build with cmake and ninja (hipcc cross-compiles gfx950), check that the library
built this way exports exactly what include/vk.h declares, install it, and build and run a consumer project that
finds the package and links the class layer (a program that needs no device: the binary-interface check and the PLY
writer)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONSUMER_CMAKE = """cmake_minimum_required(VERSION 3.21)
project(consumer LANGUAGES CXX)
find_package(Vulcan 1.0 REQUIRED)
add_executable(consumer main.cpp)
target_link_libraries(consumer PRIVATE Vulcan::vulcan)
"""

CONSUMER_MAIN = """#include <vulcan/vulcan.h>
#include <vk.h>
#include <cstdio>
int main(int argc, char** argv)
{
  if (vk_abi_check(VK_ABI_VERSION, sizeof(vk_volume), sizeof(vk_frame), VK_CTR_COUNT) != VK_OK) return 3;
  vulcan::Mesh mesh;
  mesh.points.push_back(vulcan::Vector3f(0.0f, 0.0f, 1.0f));
  mesh.points.push_back(vulcan::Vector3f(0.5f, 0.0f, 1.0f));
  mesh.points.push_back(vulcan::Vector3f(0.0f, 0.25f, 2.0f));
  mesh.faces.push_back(vulcan::Vector3i(0, 1, 2));
  vulcan::Exporter exporter(argv[1]);
  exporter.Export(mesh);
  std::printf("abi %d\\n", vk_abi_version());
  return 0;
}
"""


def run(cmd, cwd=None):
    proc = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert proc.returncode == 0, f"{' '.join(cmd)}\n{proc.stdout[-4000:]}"
    return proc.stdout


@pytest.mark.skipif(shutil.which("cmake") is None or shutil.which("ninja") is None, reason="needs cmake and ninja")
def test_cmake_builds_installs_and_exports_the_package(tmp_path):
    build, prefix = tmp_path / "build", tmp_path / "prefix"
    run(["cmake", "-S", ROOT, "-B", str(build), "-G", "Ninja", "-DCMAKE_BUILD_TYPE=Release"])
    run(["cmake", "--build", str(build), "-j", str(min(8, os.cpu_count() or 1))])
    for name in ("libvk_hip.so", "libvk_comm.so", "libvulcan.so", "host_tests", "fuse_sequence", "rig_rehearsal"):
        assert (build / name).exists(), name

    # the same C ABI as the Makefile build: every entry point of vk.h, nothing else
    text = open(os.path.join(ROOT, "include", "vk.h")).read()
    declared = sorted(set(re.findall(r"VK_API\s+[\w\s\*]+?\b(vk_\w+)\s*\(", text)))
    out = run(["nm", "-D", "--defined-only", str(build / "libvk_hip.so")])
    assert sorted(l.split()[-1] for l in out.splitlines() if " T vk_" in l) == declared
    # gfx950 code objects inside
    assert "gfx950" in run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o",
                            f"--input={build / 'libvk_hip.so'}"]) or b"gfx950" in open(build / "libvk_hip.so", "rb").read()
    # the host-thread rehearsal of the rig protocol needs no GPU
    assert "0 failure(s)" in run([str(build / "rig_rehearsal")])

    run(["cmake", "--install", str(build), "--prefix", str(prefix)])
    assert (prefix / "lib" / "cmake" / "Vulcan" / "VulcanConfig.cmake").exists()
    assert (prefix / "include" / "vulcan" / "volume.h").exists() and (prefix / "include" / "vk.h").exists()

    consumer = tmp_path / "consumer"
    consumer.mkdir()
    (consumer / "CMakeLists.txt").write_text(CONSUMER_CMAKE)
    (consumer / "main.cpp").write_text(CONSUMER_MAIN)
    run(["cmake", "-S", str(consumer), "-B", str(consumer / "build"), "-G", "Ninja", f"-DCMAKE_PREFIX_PATH={prefix}"])
    run(["cmake", "--build", str(consumer / "build")])
    ply = tmp_path / "mesh.ply"
    env_out = subprocess.run([str(consumer / "build" / "consumer"), str(ply)], stdout=subprocess.PIPE, text=True,
                             env=dict(os.environ, LD_LIBRARY_PATH=str(prefix / "lib")))
    assert env_out.returncode == 0 and env_out.stdout.startswith("abi ")
    lines = ply.read_text().split("\n")
    assert lines[0] == "ply" and lines[2] == "element vertex 3" and "3 0 1 2" in lines
